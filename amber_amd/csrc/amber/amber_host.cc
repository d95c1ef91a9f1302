// amber_host.cc -- implementation of the host object model (amber/scene.h, amber/rendering.h).
//
// Concrete primitives / materials / lens are private to this file, as in the reference
// (primitive_*.cc, material_*.cc, lens_thin.cc); only the Make* factories are exported.
// Everything computed here is scene DATA for the device; file:line citations are relative to
// /root/reference.
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <chrono>
#include <map>
#include <thread>
#include <vector>
#include <type_traits>

#include "rendering.h"
#include "scene.h"

namespace amber {
namespace prelude {

Matrix3 Matrix3::Inverse() const {   // matrix3.h:111-143
  const real_type e11 = e[0], e12 = e[1], e13 = e[2], e21 = e[3], e22 = e[4], e23 = e[5], e31 = e[6], e32 = e[7], e33 = e[8];
  const real_type d = +e11 * (e22 * e33 - e23 * e32) - e21 * (e12 * e33 - e13 * e32) + e31 * (e12 * e23 - e13 * e22);
  Matrix3 r;
  if (d == 0) {
    for (real_type& x : r.e) x = std::numeric_limits<real_type>::quiet_NaN();
    return r;
  }
  const real_type dinv = 1 / d;
  r.e[0] = +dinv * (e22 * e33 - e23 * e32); r.e[1] = -dinv * (e12 * e33 - e13 * e32); r.e[2] = +dinv * (e12 * e23 - e13 * e22);
  r.e[3] = -dinv * (e21 * e33 - e23 * e31); r.e[4] = +dinv * (e11 * e33 - e13 * e31); r.e[5] = -dinv * (e11 * e23 - e13 * e21);
  r.e[6] = +dinv * (e21 * e32 - e22 * e31); r.e[7] = -dinv * (e11 * e32 - e12 * e31); r.e[8] = +dinv * (e11 * e22 - e12 * e21);
  return r;
}

}  // namespace prelude

namespace scene {
namespace {

void Put3(float* dst, const Vector3& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

// ---- primitives -------------------------------------------------------------------------------
class Triangle : public Primitive {   // primitive_triangle.cc:33-150
 public:
  Triangle(const Vector3& v0, const Vector3& v1, const Vector3& v2) noexcept
      : v0_(v0), v1_(v1), v2_(v2), normal_(Normalize(Cross(v1 - v0, v2 - v0))) {}
  const Vector3 Center() const noexcept override {
    return Vector3((v0_.x + v1_.x + v2_.x) / 3, (v0_.y + v1_.y + v2_.y) / 3, (v0_.z + v1_.z + v2_.z) / 3);
  }
  const AABB BoundingBox() const noexcept override {
    return AABB{Vector3(std::min({v0_.x, v1_.x, v2_.x}), std::min({v0_.y, v1_.y, v2_.y}), std::min({v0_.z, v1_.z, v2_.z})),
                Vector3(std::max({v0_.x, v1_.x, v2_.x}), std::max({v0_.y, v1_.y, v2_.y}), std::max({v0_.z, v1_.z, v2_.z}))};
  }
  real_type SurfaceArea() const noexcept override { return Length(Cross(v1_ - v0_, v2_ - v0_)) / 2; }
  void Flatten(AmberFlatObject& o) const noexcept override {
    o.kind = AMBER_PRIM_TRIANGLE;
    Put3(o.p, v0_); Put3(o.p + 3, v1_); Put3(o.p + 6, v2_); Put3(o.p + 9, normal_);
  }

 private:
  Vector3 v0_, v1_, v2_;
  UnitVector3 normal_;
};

class Sphere : public Primitive {   // primitive_sphere.cc:32-125
 public:
  Sphere(const Vector3& center, real_type radius) noexcept : center_(center), radius_(radius) {}
  const Vector3 Center() const noexcept override { return center_; }
  const AABB BoundingBox() const noexcept override { return AABB{center_ - radius_, center_ + radius_}; }
  real_type SurfaceArea() const noexcept override { return 4 * static_cast<real_type>(kPI) * radius_ * radius_; }
  void Flatten(AmberFlatObject& o) const noexcept override {
    o.kind = AMBER_PRIM_SPHERE;
    Put3(o.p, center_); o.p[3] = radius_;
  }

 private:
  Vector3 center_;
  real_type radius_;
};

AABB DiskBox(const Vector3& c, const Vector3& n, real_type r) {   // primitive_disk.cc:81-92
  const Vector3 f(std::sqrt(1 - n.x * n.x), std::sqrt(1 - n.y * n.y), std::sqrt(1 - n.z * n.z));
  return AABB{c - r * f, c + r * f};
}

class Disk : public Primitive {   // primitive_disk.cc:33-139 (normal is stored as given, not normalised)
 public:
  Disk(const Vector3& center, const Vector3& normal, real_type radius) noexcept : center_(center), normal_(normal), radius_(radius) {}
  const Vector3 Center() const noexcept override { return center_; }
  const AABB BoundingBox() const noexcept override { return DiskBox(center_, normal_, radius_); }
  real_type SurfaceArea() const noexcept override { return static_cast<real_type>(kPI) * radius_ * radius_; }
  void Flatten(AmberFlatObject& o) const noexcept override {
    o.kind = AMBER_PRIM_DISK;
    Put3(o.p, center_); Put3(o.p + 3, normal_); o.p[6] = radius_;
  }

 private:
  Vector3 center_;
  UnitVector3 normal_;
  real_type radius_;
};

class Cylinder : public Primitive {   // primitive_cylinder.cc:33-167
 public:
  Cylinder(const Vector3& center, const Vector3& normal, real_type radius, real_type height) noexcept
      : center_(center), normal_(normal), radius_(radius), height_(height), bb_(AABB::Empty()) {
    bb_ += DiskBox(center_, normal_, radius_);
    bb_ += DiskBox(center_ + height_ * normal_, normal_, radius_);
  }
  const Vector3 Center() const noexcept override { return center_ + height_ / 2 * normal_; }
  const AABB BoundingBox() const noexcept override { return bb_; }
  real_type SurfaceArea() const noexcept override { return 2 * static_cast<real_type>(kPI) * radius_ * height_; }
  void Flatten(AmberFlatObject& o) const noexcept override {
    o.kind = AMBER_PRIM_CYLINDER;
    Put3(o.p, center_); Put3(o.p + 3, normal_); o.p[6] = radius_; o.p[7] = height_;
  }

 private:
  Vector3 center_;
  UnitVector3 normal_;
  real_type radius_, height_;
  AABB bb_;
};

// ---- materials ----------------------------------------------------------------------------------
class FlatMaterial : public Material {
 public:
  FlatMaterial(uint32_t kind, rendering::SurfaceType surface, const RGB& rho, real_type param, real_type r0)
      : kind_(kind), surface_(surface), rho_(rho), param_(param), r0_(r0) {}
  rendering::SurfaceType Surface() const noexcept override { return surface_; }
  const RGB Irradiance() const noexcept override {
    return kind_ == AMBER_MAT_DIFFUSE_LIGHT ? rho_ * RGB(static_cast<real_type>(kPI)) : RGB();
  }
  void Flatten(AmberFlatMaterial& m) const noexcept override {
    m.kind = kind_; m.rho[0] = rho_.x; m.rho[1] = rho_.y; m.rho[2] = rho_.z; m.param = param_; m.r0 = r0_;
  }

 private:
  uint32_t kind_;
  rendering::SurfaceType surface_;
  RGB rho_;
  real_type param_, r0_;
};

// ---- thin lens (lens_thin.cc:32-57, lens_basic.h:100-125) ---------------------------------------------
class ThinLens : public Lens {
 public:
  ThinLens(const Matrix4& transform, real_type focal_length, real_type focus_distance, real_type radius, std::size_t n_blades)
      : origin_(transform(Vector3())), global_(static_cast<Matrix3>(transform)), local_(global_.Inverse()),
        focus_distance_(focus_distance), sensor_distance_(1 / (1 / focal_length - 1 / focus_distance)), p_area_(1),
        eye_(MakeEye()) {
    for (std::size_t i = 0; i < n_blades; i++) {
      const real_type alpha = 2 * static_cast<real_type>(kPI) / n_blades * i;
      const real_type beta = 2 * static_cast<real_type>(kPI) / n_blades * (i + 1);
      primitives_.emplace_back(MakeTriangle(origin_ + global_(radius * Vector3(std::cos(alpha), std::sin(alpha), 0)),
                                            origin_ + global_(radius * Vector3(std::cos(beta), std::sin(beta), 0)), origin_));
    }
    p_area_ /= primitives_.front()->SurfaceArea() * n_blades;
    objects_.reserve(primitives_.size());
    for (const auto& p : primitives_) objects_.emplace_back(p.get(), eye_.get());
  }
  // The reference iterates an unordered_map keyed by primitive address (lens_basic.h:113-124), i.e. an
  // arbitrary order; blade order is used here (object order only matters for exact distance ties).
  std::vector<const Object*> ApertureObjects() const noexcept override {
    std::vector<const Object*> v;
    for (const auto& o : objects_) v.push_back(&o);
    return v;
  }
  void Flatten(AmberFlatThinLens& L) const noexcept override {
    Put3(L.origin, origin_);
    std::memcpy(L.global_, global_.e, sizeof L.global_);
    std::memcpy(L.local_, local_.e, sizeof L.local_);
    L.focus_distance = focus_distance_; L.sensor_distance = sensor_distance_; L.p_area = p_area_;
    L.n_blades = static_cast<uint32_t>(primitives_.size());
    L.kind = AMBER_LENS_THIN;
  }

 private:
  Vector3 origin_;
  Matrix3 global_, local_;
  real_type focus_distance_, sensor_distance_, p_area_;
  std::unique_ptr<Material> eye_;
  std::vector<std::unique_ptr<Primitive>> primitives_;
  std::vector<Object> objects_;
};

// ---- pinhole lens (lens_pinhole.cc:31-106): a degenerate aperture triangle origin/origin/origin with the Eye
// material is part of the scene exactly as in the reference (it can never be hit: its determinant is 0).
class PinholeLens : public Lens {
 public:
  PinholeLens(const Matrix4& transform, real_type sensor_distance)
      : origin_(transform(Vector3())), global_(static_cast<Matrix3>(transform)), local_(global_.Inverse()),
        sensor_distance_(sensor_distance), eye_(MakeEye()), pinhole_(MakeTriangle(origin_, origin_, origin_)),
        object_(pinhole_.get(), eye_.get()) {}
  std::vector<const Object*> ApertureObjects() const noexcept override { return {&object_}; }
  void Flatten(AmberFlatThinLens& L) const noexcept override {
    Put3(L.origin, origin_);
    std::memcpy(L.global_, global_.e, sizeof L.global_);
    std::memcpy(L.local_, local_.e, sizeof L.local_);
    L.focus_distance = 0; L.sensor_distance = sensor_distance_; L.p_area = 1;
    L.n_blades = 1; L.kind = AMBER_LENS_PINHOLE;
  }

 private:
  Vector3 origin_;
  Matrix3 global_, local_;
  real_type sensor_distance_;
  std::unique_ptr<Material> eye_;
  std::unique_ptr<Primitive> pinhole_;
  Object object_;
};

}  // namespace

std::unique_ptr<Lens> MakePinholeLens(const Matrix4& transform, real_type sensor_distance) {
  return std::make_unique<PinholeLens>(transform, sensor_distance);
}

std::unique_ptr<Primitive> MakeSphere(const Vector3& c, real_type r) noexcept { return std::make_unique<Sphere>(c, r); }
std::unique_ptr<Primitive> MakeTriangle(const Vector3& a, const Vector3& b, const Vector3& c) noexcept { return std::make_unique<Triangle>(a, b, c); }
std::unique_ptr<Primitive> MakeDisk(const Vector3& c, const Vector3& n, real_type r) noexcept { return std::make_unique<Disk>(c, n, r); }
std::unique_ptr<Primitive> MakeCylinder(const Vector3& c, const Vector3& n, real_type r, real_type h) noexcept { return std::make_unique<Cylinder>(c, n, r, h); }

std::unique_ptr<Material> MakeLambertian(const RGB& kd) {
  return std::make_unique<FlatMaterial>(AMBER_MAT_LAMBERTIAN, rendering::SurfaceType::Diffuse, kd, 0, 0);
}
std::unique_ptr<Material> MakePhong(const RGB& ks, real_type exponent) {
  return std::make_unique<FlatMaterial>(AMBER_MAT_PHONG, rendering::SurfaceType::Diffuse, ks, exponent, 0);
}
std::unique_ptr<Material> MakeSpecular(const RGB& ks) {
  return std::make_unique<FlatMaterial>(AMBER_MAT_SPECULAR, rendering::SurfaceType::Specular, ks, 0, 0);
}
std::unique_ptr<Material> MakeRefraction(real_type ior) {
  // BasicRefraction::Fresnel material_refraction.cc:265-269: std::pow(float, int) is evaluated in double
  const real_type r0 = static_cast<real_type>(std::pow(static_cast<double>((ior - 1) / (ior + 1)), 2.0));
  return std::make_unique<FlatMaterial>(AMBER_MAT_REFRACTION, rendering::SurfaceType::Specular, RGB(1), ior, r0);
}
std::unique_ptr<Material> MakeDiffuseLight(const RGB& radiance) {
  return std::make_unique<FlatMaterial>(AMBER_MAT_DIFFUSE_LIGHT, rendering::SurfaceType::Light, radiance, 0, 0);
}
std::unique_ptr<Material> MakeEye() {
  return std::make_unique<FlatMaterial>(AMBER_MAT_EYE, rendering::SurfaceType::Eye, RGB(1), 0, 0);
}

std::unique_ptr<Lens> MakeThinLens(const Matrix4& transform, real_type focal_length, real_type focus_distance,
                                   real_type radius, std::size_t n_blades) {
  if (n_blades == 0) throw std::invalid_argument("MakeThinLens: n_blades must be positive");
  return std::make_unique<ThinLens>(transform, focal_length, focus_distance, radius, n_blades);
}

FlatScene Scene::Flatten() const {
  FlatScene fs;
  std::vector<const Material*> mat_ids;
  auto material_index = [&](const Material* m) {
    for (std::size_t i = 0; i < mat_ids.size(); i++)
      if (mat_ids[i] == m) return static_cast<uint32_t>(i);
    mat_ids.push_back(m);
    AmberFlatMaterial fm{};
    m->Flatten(fm);
    fs.materials.push_back(fm);
    return static_cast<uint32_t>(mat_ids.size() - 1);
  };
  fs.objects.reserve(objects_.size());
  for (const Object& o : objects_) {
    AmberFlatObject fo{};
    o.GetPrimitive()->Flatten(fo);
    fo.material = material_index(o.GetMaterial());
    fs.objects.push_back(fo);
  }
  AmberFlatThinLens L{};
  lens_->Flatten(L);
  const auto blades = lens_->ApertureObjects();
  // locate blade 0 and require blade order
  std::size_t first = objects_.size();
  for (std::size_t i = 0; i < objects_.size(); i++)
    if (objects_[i].GetPrimitive() == blades.front()->GetPrimitive()) { first = i; break; }
  if (first + blades.size() > objects_.size()) throw std::runtime_error("Scene::Flatten: aperture objects are not part of the scene");
  for (std::size_t b = 0; b < blades.size(); b++)
    if (objects_[first + b].GetPrimitive() != blades[b]->GetPrimitive())
      throw std::runtime_error("Scene::Flatten: aperture objects must be contiguous and in blade order");
  L.first_blade_object = static_cast<uint32_t>(first);
  // scene::LightSet (light_set.h:61-82, built by Scene::Create scene/scene.h:177-182): Light-surface objects sorted by
  // Power = Sum(SurfaceArea * Irradiance) (scene/object.h:99-103) with running sums; pdf_area per light_set.h:107-111.
  {
    struct Item { uint32_t index; real_type power; RGB irr; };
    std::vector<Item> items;
    for (std::size_t i = 0; i < objects_.size(); i++) {
      if (objects_[i].Surface() != rendering::SurfaceType::Light) continue;
      const RGB irr = objects_[i].Irradiance();
      const RGB p = RGB(objects_[i].SurfaceArea()) * irr;
      items.push_back(Item{static_cast<uint32_t>(i), p.x + p.y + p.z, irr});
    }
    std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.power < b.power; });
    real_type power = 0;
    for (const Item& it : items) {
      power += it.power;
      AmberFlatLight fl{};
      fl.object = it.index; fl.cum_power = power;
      fl.irradiance[0] = it.irr.x; fl.irradiance[1] = it.irr.y; fl.irradiance[2] = it.irr.z;
      fs.lights.push_back(fl);
    }
    for (AmberFlatLight& fl : fs.lights) fl.pdf_area = (fl.irradiance[0] + fl.irradiance[1] + fl.irradiance[2]) / power;
  }
  fs.flat.objects = fs.objects.data(); fs.flat.n_objects = static_cast<uint32_t>(fs.objects.size());
  fs.flat.materials = fs.materials.data(); fs.flat.n_materials = static_cast<uint32_t>(fs.materials.size());
  fs.flat.lens = L;
  fs.flat.lights = fs.lights.empty() ? nullptr : fs.lights.data(); fs.flat.n_lights = static_cast<uint32_t>(fs.lights.size());
  return fs;
}

}  // namespace scene

// ---- etude::CornelBox (src/amber/etude/cornel_box.cc:38-204) -------------------------------------------
namespace etude {

scene::RGBScene CornelBox(scene::real_type focal_length, scene::real_type aperture_radius, std::size_t aperture_n_blades) {
  using namespace scene;
  std::vector<std::unique_ptr<Primitive>> primitives;
  std::vector<std::unique_ptr<RGBMaterial>> materials;
  std::vector<RGBObject> objects;

  auto lens = MakeThinLens(Matrix4(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 4, 0, 0, 0, 1), focal_length, 4, aperture_radius, aperture_n_blades);
  for (const auto& object : lens->ApertureObjects()) objects.emplace_back(*object);

  auto quad = [&](const Material* m, const Vector3& a, const Vector3& b, const Vector3& c, const Vector3& d, const Vector3& e,
                  const Vector3& f) {
    primitives.emplace_back(MakeTriangle(a, b, c));
    objects.emplace_back(primitives.back().get(), m);
    primitives.emplace_back(MakeTriangle(d, e, f));
    objects.emplace_back(primitives.back().get(), m);
  };
  // literals are double in the reference and narrow at the Vector3 / RGB / real_type parameters
  auto F = [](double v) { return static_cast<float>(v); };
  auto V = [&](double x, double y, double z) { return Vector3(F(x), F(y), F(z)); };

  materials.emplace_back(MakeDiffuseLight(RGB(F(1e11), F(1e11), F(1e11))));   // light source
  quad(materials.back().get(), V(0.01, 0.99, 0.01), V(-0.01, 0.99, 0.01), V(-0.01, 0.99, -0.01),
       V(-0.01, 0.99, -0.01), V(0.01, 0.99, -0.01), V(0.01, 0.99, 0.01));
  materials.emplace_back(MakeLambertian(RGB(F(.5), 0, 0)));              // left wall
  quad(materials.back().get(), V(-1, 1, 1), V(-1, -1, 1), V(-1, -1, -1), V(-1, -1, -1), V(-1, 1, -1), V(-1, 1, 1));
  materials.emplace_back(MakeLambertian(RGB(0, F(.5), 0)));              // right wall
  quad(materials.back().get(), V(1, 1, 1), V(1, 1, -1), V(1, -1, -1), V(1, -1, -1), V(1, -1, 1), V(1, 1, 1));
  materials.emplace_back(MakePhong(RGB(F(.95)), 256));                   // back wall
  quad(materials.back().get(), V(1, 1, -1), V(-1, 1, -1), V(-1, -1, -1), V(-1, -1, -1), V(1, -1, -1), V(1, 1, -1));
  materials.emplace_back(MakeLambertian(RGB(F(.5))));                    // floor
  quad(materials.back().get(), V(1, -1, 1), V(1, -1, -1), V(-1, -1, -1), V(-1, -1, -1), V(-1, -1, 1), V(1, -1, 1));
  materials.emplace_back(MakeLambertian(RGB(F(.5))));                    // ceiling
  quad(materials.back().get(), V(1, 1, 1), V(-1, 1, 1), V(-1, 1, -1), V(-1, 1, -1), V(1, 1, -1), V(1, 1, 1));
  materials.emplace_back(MakeRefraction(F(1.333)));                      // water surface + front strip
  quad(materials.back().get(), V(1, -0.5, 1), V(1, -0.5, -1), V(-1, -0.5, -1), V(-1, -0.5, -1), V(-1, -0.5, 1), V(1, -0.5, 1));
  quad(materials.back().get(), V(1, -0.5, 1), V(-1, -0.5, 1), V(-1, -1, 1), V(-1, -1, 1), V(1, -1, 1), V(1, -0.5, 1));

  primitives.emplace_back(MakeSphere(V(0.4, -0.6, -0.5), F(0.4)));       // diffuse sphere
  materials.emplace_back(MakeLambertian(RGB(F(.5))));
  objects.emplace_back(primitives.back().get(), materials.back().get());
  primitives.emplace_back(MakeSphere(V(-0.4, -0.7, 0.1), F(0.3)));       // specular sphere
  materials.emplace_back(MakeSpecular(RGB(F(.95))));
  objects.emplace_back(primitives.back().get(), materials.back().get());
  primitives.emplace_back(MakeSphere(V(0.1, -0.8, 0.6), F(0.2)));        // refraction sphere
  materials.emplace_back(MakeRefraction(F(1.125)));
  objects.emplace_back(primitives.back().get(), materials.back().get());

  return RGBScene::Create<raytracer::BVH<real_type, RGBObject>>(std::move(primitives), std::move(materials),
                                                                std::move(objects), std::move(lens));
}

}  // namespace etude

// ---- rendering::HipPathTracing --------------------------------------------------------------------------
namespace rendering {

namespace {
struct Handle {
  amber_hip_pt* h = nullptr;
  ~Handle() { if (h) amber_hip_pt_destroy(h); }
};
void Check(int rc, const char* what) {
  if (rc != AMBER_OK) throw std::runtime_error(std::string(what) + ": " + amber_hip_last_error());
}
}  // namespace

const Image<RGB> HipPathTracing::Render(const Scene<RGB>& scene, const Sensor& sensor, Context& context) {
  stats_ = HipPathTracingStats();
  const scene::FlatScene fs = scene.Flatten();
  AmberSensor s{static_cast<uint32_t>(sensor.Width()), static_cast<uint32_t>(sensor.Height()), sensor.SceneWidth(), sensor.SceneHeight()};
  const uint32_t rb = options_.row_begin, re = (options_.row_begin == 0 && options_.row_end == 0) ? s.height : options_.row_end;
  if (rb > re || re > s.height) throw std::runtime_error("HipPathTracing: bad row band");

  // One engine handle per device.  With N > 1 handles the band's rows are dealt in interleaved stripes of kStripe rows
  // (handle r renders the rows y with ((y - rb) / kStripe) % N == r): contiguous bands of the Cornell image differ by
  // 1.5x in work, stripes give equal shares.  The handles are created concurrently -- the scene
  // upload and, for large scenes, the BVH build happen per handle.
  const std::vector<int> devices = options_.devices.empty() ? std::vector<int>{options_.device} : options_.devices;
  const uint32_t n_dev = static_cast<uint32_t>(devices.size());
  constexpr uint32_t kStripe = 8;
  std::vector<Handle> handles(n_dev);
  std::vector<std::string> errors(n_dev);
  {
    std::vector<std::thread> workers;
    for (uint32_t r = 0; r < n_dev; r++)
      workers.emplace_back([&, r] {
        try {                                                            // an exception must not leave a std::thread (std::terminate)
          AmberPtParams p{};
          p.seed = options_.seed; p.max_depth = options_.max_depth; p.device = devices[r]; p.engine = options_.engine; p.reserved = options_.flags;
          p.row_begin = rb; p.row_end = re;
          if (n_dev > 1) {
            p.row_begin = std::min(rb + r * kStripe, re); p.row_end = re;
            p.stripe_rows = kStripe; p.stripe_period = kStripe * n_dev;
            if (p.row_begin == p.row_end && p.row_begin == 0) return;     // empty image: nothing to create (0,0 would mean "all rows")
          }
          if (amber_hip_pt_create(&fs.flat, &s, &p, &handles[r].h) != AMBER_OK) errors[r] = amber_hip_last_error();   // thread-local message
        } catch (const std::exception& e) {
          errors[r] = std::string("exception: ") + e.what();
        } catch (...) {
          errors[r] = "unknown exception";
        }
      });
    for (auto& w : workers) w.join();
  }
  for (uint32_t r = 0; r < n_dev; r++)
    if (!errors[r].empty()) throw std::runtime_error("amber_hip_pt_create (device " + std::to_string(devices[r]) + "): " + errors[r]);

  // Context contract (context.cc:48-60): each successful Iterate() is one whole-image sample.
  // Claim a batch of passes, render them on every handle (launches are asynchronous: the devices run concurrently), wait
  // for all, repeat until Iterate() fails.  samples_per_launch = 0 (default): the batch adapts to TIME -- it starts at one
  // accumulation chunk and grows (at least doubling, at most to what the measured rate says fills the target) while a batch takes less than kBatchTargetMs, so that expiry (--time, SIGINT) and the
  // progress line stay responsive whatever the scene costs, and a long render does not synchronise with the host every 64
  // samples (round 2: `--spp 1024` was 16 launches + 16 host synchronisations per device).  Batches are multiples of the
  // chunk, so the summation order -- and every bit of the image -- is that of one launch.
  constexpr double kBatchTargetMs = 100.0;
  constexpr uint32_t kBatchMax = 4096;
  const bool adaptive = options_.samples_per_launch == 0;
  uint32_t batch = adaptive ? AMBER_ACCUM_CHUNK : options_.samples_per_launch;
  uint32_t first = 0;
  for (;;) {
    uint32_t n = 0;
    while (n < batch && context.Iterate()) n++;
    if (n == 0) break;
    const auto t0 = std::chrono::steady_clock::now();
    for (auto& hd : handles) if (hd.h) Check(amber_hip_pt_render_pass(hd.h, first, n), "amber_hip_pt_render_pass");
    for (auto& hd : handles) if (hd.h) Check(amber_hip_pt_sync(hd.h), "amber_hip_pt_sync");   // bounded run-ahead: progress polling stays truthful
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    first += n;
    stats_.passes += n;
    if (n < batch) break;
    if (adaptive && ms < kBatchTargetMs && batch < kBatchMax) {
      // at least double; and straight to the batch the measured rate says fills the target (a launch has a fixed cost -- 0.6 ms on the Cornell box,
      // several ms of long traversals on a large mesh -- so four doublings where one step would do cost the CLI more than half of a short render:
      // the 1M-triangle terrain at 960 x 540 @ 64 spp took 38.6 ms of kernel time in launches of 8 + 16 + 32 + 8 samples).  The first batch's time
      // includes the handle's one-time costs, which only makes the estimate cautious.
      const double per_sample = ms / static_cast<double>(n);
      uint64_t want = per_sample > 0 ? static_cast<uint64_t>(kBatchTargetMs / per_sample) : kBatchMax;
      want = want / AMBER_ACCUM_CHUNK * AMBER_ACCUM_CHUNK;
      batch = static_cast<uint32_t>(std::min<uint64_t>(kBatchMax, std::max<uint64_t>(2ull * batch, want)));
    }
  }

  auto image = sensor.CreateImage<RGB>();
  std::vector<float> band;
  std::map<int, double> device_ms;
  for (uint32_t r = 0; r < n_dev; r++) {
    if (!handles[r].h) continue;
    uint32_t local_rows = 0;
    Check(amber_hip_pt_local_rows(handles[r].h, &local_rows), "amber_hip_pt_local_rows");
    band.resize(static_cast<std::size_t>(local_rows) * s.width * 3);
    uint64_t rays = 0;
    Check(amber_hip_pt_download(handles[r].h, band.data(), &rays), "amber_hip_pt_download");
    stats_.rays += rays;
    uint32_t launches = 0; double ms = 0;
    Check(amber_hip_pt_kernel_time(handles[r].h, &launches, &ms), "amber_hip_pt_kernel_time");
    stats_.launches += launches;
    device_ms[devices[r]] += ms;                                 // handles on one device run one after another ...
    // local row l of handle r is global row y: contiguous band, or the l-th row of its stripes
    const uint32_t y0 = n_dev > 1 ? std::min(rb + r * kStripe, re) : rb;
    for (uint32_t l = 0; l < local_rows; l++) {
      const uint32_t y = n_dev > 1 ? y0 + (l / kStripe) * kStripe * n_dev + l % kStripe : y0 + l;
      for (uint32_t x = 0; x < s.width; x++) {
        const float* v = &band[(static_cast<std::size_t>(l) * s.width + x) * 3];
        image[Pixel(x, y)] = RGB(v[0], v[1], v[2]);
      }
    }
  }
  for (const auto& kv : device_ms) stats_.kernel_ms = std::max(stats_.kernel_ms, kv.second);   // ... distinct devices side by side
  // Accumulator::Mean (accumulator.h:88-95): Sum() / size_ -- component-wise binary32 division
  if (stats_.passes) image /= RGB(static_cast<real_type>(stats_.passes));
  return image;
}

// rendering::LightTracing (algorithm_lt.cc:82-163) on the HIP engine: per claimed pass W*H light paths; their splats come
// back sorted in the reference's accumulation order and are added to a pass image, pass images to the running sum.
const Image<RGB> HipLightTracing::Render(const Scene<RGB>& scene, const Sensor& sensor, Context& context) {
  stats_ = HipPathTracingStats();
  const scene::FlatScene fs = scene.Flatten();
  AmberSensor s{static_cast<uint32_t>(sensor.Width()), static_cast<uint32_t>(sensor.Height()), sensor.SceneWidth(), sensor.SceneHeight()};
  // One handle per listed device, each with the whole scene and sensor.  Light paths are independent (the sampler is
  // seeded per (light path, pass)), so the path INDEX shards across the devices the way rows do for path tracing
  // (the reference parallelises lt exactly like pt: algorithm_lt.cc:82-95 -> ParallelMean): device r traces the
  // paths [r P / N, (r + 1) P / N) of every claimed pass, on a host thread of its own (amber_hip_lt_trace_range is
  // synchronous), and the splat lists -- each sorted (pass, path, bounce), over disjoint path ranges -- are merged into
  // the reference's accumulation order before they are added.  The image is bit-identical to one device's.
  const std::vector<int> devices = options_.devices.empty() ? std::vector<int>{options_.device} : options_.devices;
  const uint32_t n_dev = static_cast<uint32_t>(devices.size());
  const uint32_t n_paths = s.width * s.height;
  std::vector<Handle> handles(n_dev);
  for (uint32_t r = 0; r < n_dev; r++) {
    AmberPtParams p{};
    p.seed = options_.seed; p.max_depth = options_.max_depth; p.device = devices[r]; p.engine = options_.engine; p.reserved = options_.flags;
    Check(amber_hip_pt_create(&fs.flat, &s, &p, &handles[r].h), "amber_hip_pt_create");
  }
  auto sum = sensor.CreateImage<RGB>();
  std::vector<std::vector<AmberSplat>> lists(n_dev, std::vector<AmberSplat>(1u << 16));
  std::vector<uint32_t> counts(n_dev, 0);
  std::vector<uint64_t> rays(n_dev, 0);
  std::vector<std::string> errors(n_dev);
  std::vector<AmberSplat> merged;
  const uint32_t batch = options_.samples_per_launch ? options_.samples_per_launch : 64;
  uint32_t first = 0;
  for (;;) {
    uint32_t n = 0;
    while (n < batch && context.Iterate()) n++;
    if (n == 0) break;
    auto trace = [&](uint32_t r) {
      try {
        const uint32_t p0 = static_cast<uint32_t>(static_cast<uint64_t>(n_paths) * r / n_dev), p1 = static_cast<uint32_t>(static_cast<uint64_t>(n_paths) * (r + 1) / n_dev);
        int rc = amber_hip_lt_trace_range(handles[r].h, first, n, p0, p1, lists[r].data(), static_cast<uint32_t>(lists[r].size()), &counts[r], &rays[r]);
        if (rc == AMBER_ENOMEM && counts[r] > lists[r].size()) {    // grow once and repeat the (deterministic) batch
          lists[r].resize(counts[r]);
          rc = amber_hip_lt_trace_range(handles[r].h, first, n, p0, p1, lists[r].data(), static_cast<uint32_t>(lists[r].size()), &counts[r], &rays[r]);
        }
        if (rc != AMBER_OK) errors[r] = amber_hip_last_error();
      } catch (const std::exception& e) { errors[r] = std::string("exception: ") + e.what(); }
        catch (...) { errors[r] = "unknown exception"; }
    };
    if (n_dev == 1) trace(0);
    else {
      std::vector<std::thread> workers;
      for (uint32_t r = 0; r < n_dev; r++) workers.emplace_back(trace, r);
      for (auto& w : workers) w.join();
    }
    for (uint32_t r = 0; r < n_dev; r++)
      if (!errors[r].empty()) throw std::runtime_error("amber_hip_lt_trace_range (device " + std::to_string(devices[r]) + "): " + errors[r]);
    const AmberSplat* all = lists[0].data();
    uint32_t n_out = counts[0];
    if (n_dev > 1) {
      merged.clear();
      for (uint32_t r = 0; r < n_dev; r++) merged.insert(merged.end(), lists[r].begin(), lists[r].begin() + counts[r]);
      // (pass, path, bounce): within a pass the devices' path ranges ascend with r, so a stable sort by pass would do;
      // the full key keeps the order independent of how the ranges were cut
      std::sort(merged.begin(), merged.end(), [](const AmberSplat& x, const AmberSplat& y) {
        if (x.sample != y.sample) return x.sample < y.sample;
        if (x.path != y.path) return x.path < y.path;
        return x.bounce < y.bounce;
      });
      all = merged.data(); n_out = static_cast<uint32_t>(merged.size());
    }
    for (uint32_t r = 0; r < n_dev; r++) stats_.rays += rays[r];
    stats_.launches++;
    uint32_t k = 0;
    for (uint32_t pass = first; pass < first + n; pass++) {
      if (k >= n_out || all[k].sample != pass) continue;
      auto image = sensor.CreateImage<RGB>();                // one pass image, algorithm_lt.cc:115-122
      for (; k < n_out && all[k].sample == pass; k++)
        image[Pixel(all[k].pixel % s.width, all[k].pixel / s.width)] += RGB(all[k].rgb[0], all[k].rgb[1], all[k].rgb[2]);
      sum += image;
    }
    first += n; stats_.passes += n;
    if (n < batch) break;
  }
  if (stats_.passes) sum /= RGB(static_cast<real_type>(stats_.passes));
  return sum;
}

std::unique_ptr<Algorithm<RGB>> MakeRGBHipLightTracing(const HipPathTracingOptions& options) {
  return std::make_unique<HipLightTracing>(options);
}

std::unique_ptr<Algorithm<RGB>> MakeRGBHipPathTracing(const HipPathTracingOptions& options) {
  return std::make_unique<HipPathTracing>(options);
}

}  // namespace rendering

namespace cli {
std::unique_ptr<rendering::Algorithm<rendering::RGB>> MakeAlgorithm(const std::string& name,
                                                                   const rendering::HipPathTracingOptions& options) {
  // algorithm_factory.cc:35-79 dispatches on --algorithm; this package provides the path tracer only.
  if (name == "pt" || name == "pt-hip") return rendering::MakeRGBHipPathTracing(options);
  if (name == "lt" || name == "lt-hip") return rendering::MakeRGBHipLightTracing(options);
  throw UnknownAlgorithmError(name);
}
}  // namespace cli

}  // namespace amber
