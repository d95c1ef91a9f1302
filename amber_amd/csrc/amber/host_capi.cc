// host_capi.cc -- C shim (include/amber_host.h) over the C++ host object model.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/amber_host.h"
#include "import.h"
#include "postprocess.h"
#include "rendering.h"
#include "scene.h"

using namespace amber;

struct amber_host_scene {
  scene::RGBScene scene;
  explicit amber_host_scene(scene::RGBScene&& s) : scene(std::move(s)) {}
};

namespace {
thread_local std::string g_err;
int Fail(int code, const std::string& m) { g_err = m; return code; }
}  // namespace

extern "C" {

const char* amber_host_last_error(void) { return g_err.c_str(); }

amber_host_scene* amber_host_cornell_box(float focal_length, float aperture_radius, uint32_t n_blades) {
  try {
    return new amber_host_scene(etude::CornelBox(focal_length, aperture_radius, n_blades));
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

amber_host_scene* amber_host_scene_import(const char* filename) {
  try {
    if (!filename) { g_err = "null filename"; return nullptr; }
    return new amber_host_scene(cli::ImportSceneBVH(filename));
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

amber_host_scene* amber_host_scene_create(const AmberFlatObject* objects, uint32_t n_objects,
                                          const AmberFlatMaterial* materials, uint32_t n_materials,
                                          const float t[16], float focal_length, float focus_distance,
                                          float radius, uint32_t n_blades, int accel) {
  try {
    using namespace scene;
    if (!objects || !materials || !t) throw std::invalid_argument("null argument");
    std::vector<std::unique_ptr<Primitive>> primitives;
    std::vector<std::unique_ptr<RGBMaterial>> mats;
    std::vector<RGBObject> objs;
    const Matrix4 tm(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9], t[10], t[11], t[12], t[13], t[14], t[15]);
    // n_blades == 0 selects MakePinholeLens(transform, sensor_distance = focal_length)
    auto lens = n_blades ? MakeThinLens(tm, focal_length, focus_distance, radius, n_blades) : MakePinholeLens(tm, focal_length);
    for (const auto& o : lens->ApertureObjects()) objs.emplace_back(*o);
    for (uint32_t i = 0; i < n_materials; i++) {
      const AmberFlatMaterial& m = materials[i];
      const RGB rho(m.rho[0], m.rho[1], m.rho[2]);
      switch (m.kind) {
        case AMBER_MAT_LAMBERTIAN: mats.emplace_back(MakeLambertian(rho)); break;
        case AMBER_MAT_PHONG: mats.emplace_back(MakePhong(rho, m.param)); break;
        case AMBER_MAT_SPECULAR: mats.emplace_back(MakeSpecular(rho)); break;
        case AMBER_MAT_REFRACTION: mats.emplace_back(MakeRefraction(m.param)); break;
        case AMBER_MAT_DIFFUSE_LIGHT: mats.emplace_back(MakeDiffuseLight(rho)); break;
        case AMBER_MAT_EYE: mats.emplace_back(MakeEye()); break;
        default: throw std::invalid_argument("unknown material kind");
      }
    }
    for (uint32_t i = 0; i < n_objects; i++) {
      const AmberFlatObject& o = objects[i];
      if (o.material >= n_materials) throw std::invalid_argument("material index out of range");
      const float* p = o.p;
      switch (o.kind) {
        case AMBER_PRIM_TRIANGLE:
          primitives.emplace_back(MakeTriangle(Vector3(p[0], p[1], p[2]), Vector3(p[3], p[4], p[5]), Vector3(p[6], p[7], p[8]))); break;
        case AMBER_PRIM_SPHERE: primitives.emplace_back(MakeSphere(Vector3(p[0], p[1], p[2]), p[3])); break;
        case AMBER_PRIM_DISK: primitives.emplace_back(MakeDisk(Vector3(p[0], p[1], p[2]), Vector3(p[3], p[4], p[5]), p[6])); break;
        case AMBER_PRIM_CYLINDER: primitives.emplace_back(MakeCylinder(Vector3(p[0], p[1], p[2]), Vector3(p[3], p[4], p[5]), p[6], p[7])); break;
        default: throw std::invalid_argument("unknown primitive kind");
      }
      objs.emplace_back(primitives.back().get(), mats[o.material].get());
    }
    if (accel == 1)
      return new amber_host_scene(RGBScene::Create<raytracer::List<real_type, RGBObject>>(std::move(primitives), std::move(mats), std::move(objs), std::move(lens)));
    return new amber_host_scene(RGBScene::Create<raytracer::BVH<real_type, RGBObject>>(std::move(primitives), std::move(mats), std::move(objs), std::move(lens)));
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

void amber_host_scene_destroy(amber_host_scene* s) { delete s; }

int amber_host_scene_flatten(const amber_host_scene* s, AmberFlatObject* objects, uint32_t* n_objects,
                             AmberFlatMaterial* materials, uint32_t* n_materials, AmberFlatThinLens* lens) {
  if (!s || !n_objects || !n_materials) return Fail(AMBER_EINVAL, "null argument");
  try {
    const scene::FlatScene fs = s->scene.Flatten();
    if (objects) {
      if (*n_objects < fs.objects.size()) return Fail(AMBER_EINVAL, "object buffer too small");
      std::memcpy(objects, fs.objects.data(), fs.objects.size() * sizeof(AmberFlatObject));
    }
    if (materials) {
      if (*n_materials < fs.materials.size()) return Fail(AMBER_EINVAL, "material buffer too small");
      std::memcpy(materials, fs.materials.data(), fs.materials.size() * sizeof(AmberFlatMaterial));
    }
    *n_objects = static_cast<uint32_t>(fs.objects.size());
    *n_materials = static_cast<uint32_t>(fs.materials.size());
    if (lens) *lens = fs.flat.lens;
    return AMBER_OK;
  } catch (const std::exception& e) { return Fail(AMBER_EINVAL, e.what()); }
}

int amber_host_pt_create(const amber_host_scene* s, const AmberSensor* sensor, const AmberPtParams* params, amber_hip_pt** out) {
  if (!s) return Fail(AMBER_EINVAL, "null scene");
  try {
    const scene::FlatScene fs = s->scene.Flatten();
    const int rc = amber_hip_pt_create(&fs.flat, sensor, params, out);
    if (rc != AMBER_OK) g_err = amber_hip_last_error();
    return rc;
  } catch (const std::exception& e) { return Fail(AMBER_EINVAL, e.what()); }
}

int amber_host_render_devices(const amber_host_scene* s, const char* algorithm, const AmberSensor* sensor, uint32_t spp,
                              uint64_t seed, uint32_t max_depth, const int* devices, uint32_t n_devices, uint32_t samples_per_launch,
                              float* out_rgb, AmberHostStats* stats) {
  if (!s || !algorithm || !sensor || !out_rgb || (n_devices && !devices)) return Fail(AMBER_EINVAL, "null argument");
  try {
    rendering::HipPathTracingOptions opt;
    opt.seed = seed; opt.max_depth = max_depth;
    if (devices && n_devices) { opt.device = devices[0]; opt.devices.assign(devices, devices + n_devices); }
    if (samples_per_launch) opt.samples_per_launch = samples_per_launch;
    auto algo = cli::MakeAlgorithm(algorithm, opt);
    const rendering::Sensor sn(sensor->width, sensor->height, sensor->scene_width, sensor->scene_height);
    cli::Context ctx(1, spp);
    const auto image = algo->Render(s->scene, sn, ctx);
    std::memcpy(out_rgb, image.Data(), static_cast<size_t>(sensor->width) * sensor->height * 3 * sizeof(float));
    if (stats) {
      const rendering::HipPathTracingStats* st = nullptr;
      if (const auto* hp = dynamic_cast<rendering::HipPathTracing*>(algo.get())) st = &hp->Stats();
      if (const auto* hl = dynamic_cast<rendering::HipLightTracing*>(algo.get())) st = &hl->Stats();
      if (st) { stats->rays = st->rays; stats->passes = st->passes; stats->launches = st->launches; stats->kernel_ms = st->kernel_ms; }
    }
    return AMBER_OK;
  } catch (const cli::UnknownAlgorithmError& e) { return Fail(AMBER_EINVAL, e.what()); }
  catch (const std::exception& e) { return Fail(AMBER_EHIP, e.what()); }
}

int amber_host_render(const amber_host_scene* s, const char* algorithm, const AmberSensor* sensor, uint32_t spp,
                      uint64_t seed, uint32_t max_depth, int device, uint32_t samples_per_launch,
                      float* out_rgb, AmberHostStats* stats) {
  return amber_host_render_devices(s, algorithm, sensor, spp, seed, max_depth, &device, 1, samples_per_launch, out_rgb, stats);
}

extern "C++" {
namespace {
postprocess::HDRImage ToImage(const float* rgb, uint32_t w, uint32_t h) {
  postprocess::HDRImage img(w, h);
  std::memcpy(img.Data(), rgb, static_cast<size_t>(w) * h * 3 * sizeof(float));
  return img;
}
}  // namespace
}  // extern "C++"

int amber_host_tonemap(const float* rgb, uint32_t width, uint32_t height, uint8_t* out_rgb8) {
  if (!rgb || !out_rgb8 || !width || !height) return Fail(AMBER_EINVAL, "bad argument");
  try {
    const auto ldr = postprocess::Gamma()(postprocess::Filmic()(ToImage(rgb, width, height)));
    std::memcpy(out_rgb8, ldr.rgb.data(), ldr.rgb.size());
    return AMBER_OK;
  } catch (const std::exception& e) { return Fail(AMBER_EINVAL, e.what()); }
}

int amber_host_export(const float* rgb, uint32_t width, uint32_t height, const char* png_path, const char* exr_path) {
  if (!rgb || !width || !height) return Fail(AMBER_EINVAL, "bad argument");
  try {
    const auto img = ToImage(rgb, width, height);
    if (png_path) cli::ExportPNG(postprocess::Gamma()(postprocess::Filmic()(img)), png_path);
    if (exr_path) cli::ExportEXR(img, exr_path);
    return AMBER_OK;
  } catch (const std::exception& e) { return Fail(AMBER_EINVAL, e.what()); }
}

}  // extern "C"
