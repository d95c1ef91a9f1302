// amber/rendering.h -- the integrator boundary (amber::rendering).
//
//   Algorithm<Radiant>::Render(scene, sensor, context)   include/amber/rendering/algorithm.h:30-46
//   Context { ThreadCount, IterationCount, Iterate }     include/amber/rendering/context.h:26-32
//   Sensor(width, height, scene_width, scene_height)     include/amber/rendering/sensor.h:34-92
//   MakeRGBPathTracing()                                 include/amber/rendering/algorithm_pt.h:30-31
//
// MakeRGBHipPathTracing() is the drop-in for MakeRGBPathTracing(): same interface, same Context
// contract (one Iterate() == one sample per pixel over the whole image; the result is the sum of
// the started passes divided by their number, accumulator.h:88-95), rendered by the gfx950 engine.
#pragma once

#include <cstdint>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "prelude.h"
#include "scene.h"

#pragma GCC visibility push(default)   // the C++ interface of the host object model is exported (bin/amber links against it)
namespace amber {
namespace rendering {

using prelude::Image;
using prelude::Pixel;
using prelude::pixel_size_type;
using prelude::real_type;
using RGB = prelude::Vector3;
template <typename Radiant> using Scene = scene::Scene;   // only Radiant = RGB is instantiated (postprocess/forward.h:36)

class Sensor {
 public:
  Sensor(pixel_size_type pixel_width, pixel_size_type pixel_height, real_type scene_width, real_type scene_height) noexcept
      : pixel_width_(pixel_width), pixel_height_(pixel_height), scene_width_(scene_width), scene_height_(scene_height) {}
  template <typename Radiant> Image<Radiant> CreateImage() const { return Image<Radiant>(pixel_width_, pixel_height_); }
  pixel_size_type Size() const noexcept { return pixel_width_ * pixel_height_; }
  real_type SceneArea() const noexcept { return scene_width_ * scene_height_; }
  pixel_size_type Width() const noexcept { return pixel_width_; }
  pixel_size_type Height() const noexcept { return pixel_height_; }
  real_type SceneWidth() const noexcept { return scene_width_; }
  real_type SceneHeight() const noexcept { return scene_height_; }

 private:
  pixel_size_type pixel_width_, pixel_height_;
  real_type scene_width_, scene_height_;
};

class Context {
 public:
  virtual ~Context() {}
  virtual std::size_t ThreadCount() const noexcept = 0;
  virtual std::size_t IterationCount() const noexcept = 0;
  virtual bool Iterate() noexcept = 0;
};

template <typename Radiant>
class Algorithm {
 public:
  virtual ~Algorithm() {}
  virtual const Image<Radiant> Render(const Scene<Radiant>& scene, const Sensor& sensor, Context& context) = 0;
};

/** Options of the HIP integrator (none of them exist in the reference; defaults reproduce it). */
struct HipPathTracingOptions {
  std::uint64_t seed = 12345;       // global seed of the per-(pixel,sample) XorShift sampler
  std::uint32_t max_depth = 0;      // 0 = Russian roulette only
  int device = 0;                   // HIP device ordinal (used when `devices` is empty)
  // Multi-GPU inside ONE Render() call -- the counterpart of the reference's thread fan-out (prelude/parallel.cc:29-40,
  // rendering/parallel.h:57-68), which parallelises inside Render too: one engine handle per listed device (an ordinal
  // may repeat), the framebuffer rows dealt to the handles in interleaved 8-row stripes, every handle renders ALL samples
  // of ITS rows, the row sums are copied straight to the host image.  Pixels are independent under the per-(pixel,sample)
  // sampler, so the image is bit-identical to a single-device render.
  std::vector<int> devices;
  std::uint32_t samples_per_launch = 0;    // Context::Iterate() calls claimed per kernel launch; 0 = adapt to time: start at one
                                           // accumulation chunk, double while a batch takes < 100 ms (expiry latency stays bounded)
  std::uint32_t flags = 0;          // AMBER_PT_FLAG_* bits passed to every handle
  std::uint32_t row_begin = 0, row_end = 0;   // framebuffer band; 0,0 = whole image
  std::uint32_t engine = 0;         // AMBER_ENGINE_*
};

/** Statistics of the last Render() call. */
struct HipPathTracingStats {
  std::uint64_t rays = 0;           // Scene::Cast equivalents
  std::uint64_t passes = 0;         // Context::Iterate() calls that returned true
  double kernel_ms = 0;             // summed kernel time (hipEvents)
  std::uint32_t launches = 0;
};

class HipPathTracing : public Algorithm<RGB> {
 public:
  explicit HipPathTracing(const HipPathTracingOptions& options) : options_(options) {}
  /** Throws std::runtime_error carrying amber_hip_last_error() if the engine fails (no CPU fallback). */
  const Image<RGB> Render(const Scene<RGB>& scene, const Sensor& sensor, Context& context) override;
  const HipPathTracingStats& Stats() const noexcept { return stats_; }

 private:
  HipPathTracingOptions options_;
  HipPathTracingStats stats_;
};

std::unique_ptr<Algorithm<RGB>> MakeRGBHipPathTracing(const HipPathTracingOptions& options = HipPathTracingOptions());

/** Drop-in for MakeRGBLightTracing() (include/amber/rendering/algorithm_lt.h:30-31, algorithm_lt.cc:82-163). */
class HipLightTracing : public Algorithm<RGB> {
 public:
  explicit HipLightTracing(const HipPathTracingOptions& options) : options_(options) {}
  const Image<RGB> Render(const Scene<RGB>& scene, const Sensor& sensor, Context& context) override;
  const HipPathTracingStats& Stats() const noexcept { return stats_; }

 private:
  HipPathTracingOptions options_;
  HipPathTracingStats stats_;
};
std::unique_ptr<Algorithm<RGB>> MakeRGBHipLightTracing(const HipPathTracingOptions& options = HipPathTracingOptions());

}  // namespace rendering

namespace cli {
/** cli::Context, src/amber/cli/context.cc:26-67: mutex-guarded pass counter with expiry. */
class Context : public rendering::Context {
 public:
  Context(std::size_t n_threads, std::size_t n_iterations) noexcept
      : n_threads_(n_threads), n_iterations_(n_iterations), n_iterations_started_(0), is_expired_(false) {}
  std::size_t ThreadCount() const noexcept override { std::lock_guard<std::mutex> l(mutex_); return n_threads_; }
  std::size_t IterationCount() const noexcept override { std::lock_guard<std::mutex> l(mutex_); return n_iterations_started_; }
  bool Iterate() noexcept override {
    std::lock_guard<std::mutex> l(mutex_);
    if (n_iterations_ > 0 && n_iterations_started_ >= n_iterations_) return false;
    if (is_expired_) return false;
    n_iterations_started_++;
    return true;
  }
  void Expire() noexcept { std::lock_guard<std::mutex> l(mutex_); is_expired_ = true; }

 private:
  mutable std::mutex mutex_;
  std::size_t n_threads_, n_iterations_, n_iterations_started_;
  bool is_expired_;
};

/** cli::AlgorithmFactory, src/amber/cli/algorithm_factory.cc:35-79: unknown names throw. */
class UnknownAlgorithmError : public std::runtime_error {
 public:
  explicit UnknownAlgorithmError(const std::string& name) : std::runtime_error("Unknown algorithm: " + name) {}
};
std::unique_ptr<rendering::Algorithm<rendering::RGB>> MakeAlgorithm(const std::string& name,
                                                                   const rendering::HipPathTracingOptions& options);
}  // namespace cli
}  // namespace amber
#pragma GCC visibility pop
