// amber_cli.cc -- thin command-line driver: the caller side of the integrator boundary.
//
// Mirrors cli::Application (/root/reference/src/amber/cli/application.cc:49-215) and the flags of
// cli::ParseCommandLineOption (option.cc:40-100) as far as they concern the path tracer:
//   --algorithm pt|lt   --spp N (0 = until --time / SIGINT)   --width --height   --time SECONDS
//   --output BASENAME (writes BASENAME.png tone-mapped and BASENAME.exr raw)   --threads (accepted, ignored)
// plus --seed, --device, --max-depth, --engine of this implementation, and --devices N / --device-list a,b,c: N GPUs inside
// the one Render() call (rendering.h HipPathTracingOptions.devices), where the reference uses --threads.  The scene is etude::CornelBox(0.050,
// 0.050, 6) as in application.cc:68-73, or --scene FILE through cli::ImportScene (import.h: OBJ + MTL subset, the
// reference reads it through assimp) built with the BVH acceleration (application.cc:74-87).
// Render runs on a std::async thread while the main thread prints the reference's progress line every 500 ms;
// SIGINT or the --time limit call Context::Expire(), a second SIGINT aborts (application.cc:132-147).
#include <atomic>
#include <chrono>
#include <csignal>
#include <cstdlib>
#include <cstring>
#include <future>
#include <iomanip>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "postprocess.h"
#include "rendering.h"
#include "import.h"
#include "scene.h"

namespace {

amber::cli::Context* g_context = nullptr;
std::atomic<int> g_sigints{0};

void OnSigint(int) {
  if (g_sigints.fetch_add(1) > 0) std::abort();
  if (g_context) g_context->Expire();
}

struct Option {
  std::string algorithm = "pt", output = "output", scene;
  std::size_t spp = 0, threads = 1, width = 512, height = 512, time = 0;
  std::uint64_t seed = 12345;
  int device = 0;
  std::vector<int> devices;          // --devices N = ordinals 0..N-1, --device-list a,b,c = exactly those (repeats allowed)
  std::uint32_t max_depth = 0, engine = 0, samples_per_launch = 0;   // 0 = batches adapt to time (rendering.h)
  bool help = false;
};

bool Parse(int argc, char** argv, Option& o) {
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto value = [&](const char* name) -> const char* {
      if (i + 1 >= argc) { std::cerr << "missing value for " << name << std::endl; std::exit(-1); }
      return argv[++i];
    };
    if (a == "--help" || a == "-h") o.help = true;
    else if (a == "--algorithm") o.algorithm = value("--algorithm");
    else if (a == "--output") o.output = value("--output");
    else if (a == "--scene") o.scene = value("--scene");
    else if (a == "--spp") o.spp = std::strtoull(value("--spp"), nullptr, 10);
    else if (a == "--threads") o.threads = std::strtoull(value("--threads"), nullptr, 10);
    else if (a == "--width") o.width = std::strtoull(value("--width"), nullptr, 10);
    else if (a == "--height") o.height = std::strtoull(value("--height"), nullptr, 10);
    else if (a == "--time") o.time = std::strtoull(value("--time"), nullptr, 10);
    else if (a == "--seed") o.seed = std::strtoull(value("--seed"), nullptr, 10);
    else if (a == "--device") o.device = std::atoi(value("--device"));
    else if (a == "--devices") { const int n = std::atoi(value("--devices")); o.devices.clear(); for (int k = 0; k < n; k++) o.devices.push_back(k); }
    else if (a == "--device-list") {
      o.devices.clear();
      std::string list = value("--device-list");
      for (size_t pos = 0; pos <= list.size();) {
        const size_t comma = list.find(',', pos);
        const std::string item = list.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        if (!item.empty()) o.devices.push_back(std::atoi(item.c_str()));
        if (comma == std::string::npos) break;
        pos = comma + 1;
      }
    }
    else if (a == "--max-depth") o.max_depth = static_cast<std::uint32_t>(std::strtoul(value("--max-depth"), nullptr, 10));
    else if (a == "--engine") o.engine = static_cast<std::uint32_t>(std::strtoul(value("--engine"), nullptr, 10));
    else if (a == "--samples-per-launch") o.samples_per_launch = static_cast<std::uint32_t>(std::strtoul(value("--samples-per-launch"), nullptr, 10));
    else { std::cerr << "unknown option " << a << std::endl; return false; }
  }
  return true;
}

}  // namespace

int main(int argc, char** argv) {
  using namespace amber;
  Option option;
  if (!Parse(argc, argv, option) || option.help) {
    std::cerr << "amber: a global illumination renderer (MI355X path tracer)\n"
                 "  --algorithm pt  --spp N  --width W  --height H  --time S  --output NAME  [--threads N]  [--scene FILE.obj]\n"
                 "  [--seed N] [--device N | --devices N | --device-list a,b,..] [--max-depth N] [--engine 0..4|6] [--samples-per-launch N]" << std::endl;
    return option.help ? 0 : -1;
  }
  if (option.spp == 0 && option.time == 0) std::cerr << "note: --spp 0 without --time renders until SIGINT" << std::endl;

  rendering::HipPathTracingOptions hip;
  hip.seed = option.seed; hip.device = option.device; hip.max_depth = option.max_depth; hip.engine = option.engine;
  hip.samples_per_launch = option.samples_per_launch;
  hip.devices = option.devices;                                        // one engine handle per device inside Render()
  std::unique_ptr<rendering::Algorithm<rendering::RGB>> algorithm;
  try {
    algorithm = cli::MakeAlgorithm(option.algorithm, hip);            // algorithm_factory.cc:35-79
  } catch (const cli::UnknownAlgorithmError& e) {
    std::cerr << e.what() << std::endl;
    return -1;                                                         // application.cc:60-65
  }
  std::unique_ptr<scene::RGBScene> scene_holder;
  if (option.scene.empty()) {
    scene_holder = std::make_unique<scene::RGBScene>(etude::CornelBox(0.050f, 0.050f, 6));   // application.cc:68-73
  } else {
    try {                                                              // application.cc:74-87
      std::cerr << "Loading scene ... ";
      scene_holder = std::make_unique<scene::RGBScene>(cli::ImportSceneBVH(option.scene));
      std::cerr << "done." << std::endl;
    } catch (const std::exception& e) {
      std::cerr << std::endl << e.what() << std::endl;
      return -1;
    }
  }
  const scene::RGBScene& scene = *scene_holder;
  const rendering::Sensor sensor(option.width, option.height, static_cast<float>(0.036),
                                 static_cast<float>(0.036 / option.width * option.height));   // application.cc:89-94

  cli::Context context(option.threads, option.spp);                   // application.cc:129
  g_context = &context;
  std::signal(SIGINT, OnSigint);
  std::thread timer;
  std::atomic<bool> done{false};
  if (option.time > 0)
    timer = std::thread([&] {
      const auto until = std::chrono::steady_clock::now() + std::chrono::seconds(option.time);
      while (!done && std::chrono::steady_clock::now() < until) std::this_thread::sleep_for(std::chrono::milliseconds(20));
      context.Expire();
    });

  const auto begin = std::chrono::steady_clock::now();
  auto future = std::async(std::launch::async, [&] { return algorithm->Render(scene, sensor, context); });   // application.cc:149-152
  int status = 0;
  try {
    for (;;) {                                                        // application.cc:156-210
      const bool ready = future.wait_for(std::chrono::milliseconds(500)) == std::future_status::ready;
      const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count();
      const std::size_t i = context.IterationCount();
      std::cerr << "\r" << std::fixed << std::setprecision(1) << s << "s - " << i << "/" << option.spp << " - ";
      if (option.spp) std::cerr << 100.0 * i / option.spp << "% - ";
      std::cerr << (s > 0 ? i / s : 0.0) << " iterations/second" << std::flush;
      if (ready) break;
    }
    std::cerr << std::endl;
    const auto image = future.get();
    done = true;
    if (timer.joinable()) timer.join();
    const rendering::HipPathTracingStats* stp = nullptr;
    if (const auto* hp = dynamic_cast<rendering::HipPathTracing*>(algorithm.get())) stp = &hp->Stats();
    if (const auto* hl = dynamic_cast<rendering::HipLightTracing*>(algorithm.get())) stp = &hl->Stats();
    if (stp) {
      const auto& st = *stp;
      std::cerr << st.passes << " passes, " << st.rays << " rays, kernel " << st.kernel_ms << " ms ("
                << (st.kernel_ms > 0 ? st.rays / st.kernel_ms / 1e3 : 0.0) << " Mrays/s)" << std::endl;
    }
    std::cerr << "Exporting " << option.output << ".png (tonemapped) ... ";
    cli::ExportPNG(postprocess::Gamma()(postprocess::Filmic()(image)), option.output + ".png");   // application.cc:98-108
    std::cerr << "done." << std::endl << "Exporting " << option.output << ".exr (raw) ... ";
    cli::ExportEXR(image, option.output + ".exr");                                                  // application.cc:110-115
    std::cerr << "done." << std::endl;
  } catch (const std::exception& e) {
    done = true;
    if (timer.joinable()) timer.join();
    std::cerr << std::endl << "error: " << e.what() << std::endl;
    status = -1;
  }
  return status;
}
