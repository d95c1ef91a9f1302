// amber/scene.h -- host object model: primitives, materials, lens, objects, Scene (amber::scene),
// and the flattening that feeds the HIP engine.
//
// Same factory names and argument meaning as the reference:
//   MakeSphere/MakeTriangle/MakeDisk/MakeCylinder   include/amber/scene/primitive_*.h
//   MakeLambertian/MakePhong/MakeSpecular/MakeRefraction/MakeDiffuseLight/MakeEye
//                                                   include/amber/scene/material_*.h
//   MakeThinLens                                    include/amber/scene/lens_thin.h:32-40
//   Scene::Create<Acceleration>                     include/amber/scene/scene.h:169-193
// What differs by design: the reference's abstract rendering::Scene exposes only per-ray virtual
// queries and keeps its concrete classes private to .cc files, so nothing can enumerate it
// (SURVEY.md section 8(b)).  Here every node of the model can describe itself as plain data
// (Primitive::Flatten, Material::Flatten, Lens::Flatten) and Scene::Flatten() produces the
// AmberFlatScene consumed by include/amber_hip.h.  No host-side intersection / sampling exists.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "../../../include/amber_hip.h"
#include "prelude.h"

#pragma GCC visibility push(default)   // the C++ interface of the host object model is exported (bin/amber links against it)
namespace amber {

namespace rendering {
enum class SurfaceType { Light, Eye, Specular, Diffuse };   // rendering/surface.h:28-33
}

namespace scene {

using prelude::AABB;
using prelude::Matrix3;
using prelude::Matrix4;
using prelude::real_type;
using prelude::UnitVector3;
using prelude::Vector3;
using RGB = prelude::Vector3;   // postprocess/forward.h:36

// ---- primitives (scene/primitive.h:28-38) ---------------------------------------------------
class Primitive {
 public:
  virtual ~Primitive() {}
  virtual const Vector3 Center() const noexcept = 0;
  virtual const AABB BoundingBox() const noexcept = 0;
  virtual real_type SurfaceArea() const noexcept = 0;
  /** Plain-data description (kind + parameters, see AmberFlatObject); material is filled by Object. */
  virtual void Flatten(AmberFlatObject& out) const noexcept = 0;
};
std::unique_ptr<Primitive> MakeSphere(const Vector3& center, real_type radius) noexcept;
std::unique_ptr<Primitive> MakeTriangle(const Vector3& v0, const Vector3& v1, const Vector3& v2) noexcept;
std::unique_ptr<Primitive> MakeDisk(const Vector3& center, const Vector3& normal, real_type radius) noexcept;
std::unique_ptr<Primitive> MakeCylinder(const Vector3& center, const Vector3& normal, real_type radius, real_type height) noexcept;

// ---- materials (scene/material.h:31-88) -----------------------------------------------------
class Material {
 public:
  virtual ~Material() {}
  virtual rendering::SurfaceType Surface() const noexcept = 0;
  /** material.h:93-99: zero except for DiffuseLight (radiance * pi, material_diffuse_light.h:118-125). */
  virtual const RGB Irradiance() const noexcept { return RGB(); }
  virtual void Flatten(AmberFlatMaterial& out) const noexcept = 0;
};
using RGBMaterial = Material;
std::unique_ptr<Material> MakeLambertian(const RGB& kd);
std::unique_ptr<Material> MakePhong(const RGB& ks, real_type exponent);
std::unique_ptr<Material> MakeSpecular(const RGB& ks);
std::unique_ptr<Material> MakeRefraction(real_type ior);
std::unique_ptr<Material> MakeDiffuseLight(const RGB& radiance);
std::unique_ptr<Material> MakeEye();

// ---- object = primitive x material (scene/object.h:33-96) -------------------------------------
class Object {
 public:
  Object() noexcept : primitive_(nullptr), material_(nullptr) {}
  Object(const Primitive* primitive, const Material* material) noexcept : primitive_(primitive), material_(material) {}
  const Vector3 Center() const noexcept { return primitive_->Center(); }
  const AABB BoundingBox() const noexcept { return primitive_->BoundingBox(); }
  real_type SurfaceArea() const noexcept { return primitive_->SurfaceArea(); }
  rendering::SurfaceType Surface() const noexcept { return material_->Surface(); }
  const RGB Irradiance() const noexcept { return material_->Irradiance(); }
  const Primitive* GetPrimitive() const noexcept { return primitive_; }
  const Material* GetMaterial() const noexcept { return material_; }

 private:
  const Primitive* primitive_;
  const Material* material_;
};
using RGBObject = Object;

// ---- lens (scene/lens.h:32-55) ------------------------------------------------------------------
class Lens {
 public:
  virtual ~Lens() {}
  /** Aperture objects (Eye material) in blade order; the caller inserts them into the scene
   *  (cornel_box.cc:62-64). */
  virtual std::vector<const Object*> ApertureObjects() const noexcept = 0;
  /** Fills everything of AmberFlatThinLens except first_blade_object. */
  virtual void Flatten(AmberFlatThinLens& out) const noexcept = 0;
};
std::unique_ptr<Lens> MakeThinLens(const Matrix4& transform, real_type focal_length, real_type focus_distance,
                                   real_type radius, std::size_t n_blades);
/** scene/lens_pinhole.h:32-38 */
std::unique_ptr<Lens> MakePinholeLens(const Matrix4& transform, real_type sensor_distance);

// ---- acceleration tags (raytracer/forward.h) ------------------------------------------------------
// The reference selects its acceleration structure with a template argument of Scene::Create.
// The same spelling is kept; on this engine it selects how the device finds the closest hit.
namespace accel_tag {
struct BVH {};    // host-built, flattened BVH traversed on the device (large scenes)
struct List {};   // brute force over all objects in insertion order (acceleration_list.h:51-68)
}  // namespace accel_tag

/** Owner of the flattened arrays; `flat` points into the vectors. */
struct FlatScene {
  std::vector<AmberFlatObject> objects;
  std::vector<AmberFlatMaterial> materials;
  std::vector<AmberFlatLight> lights;      // scene::LightSet order (light_set.h:61-82)
  AmberFlatScene flat{};
};

class Scene {
 public:
  template <typename Acceleration>
  static Scene Create(std::vector<std::unique_ptr<Primitive>>&& primitives,
                      std::vector<std::unique_ptr<Material>>&& materials, std::vector<Object>&& objects,
                      std::unique_ptr<Lens>&& lens) noexcept {
    return Scene(std::move(primitives), std::move(materials), std::move(objects), std::move(lens),
                 std::is_same<Acceleration, accel_tag::BVH>::value);
  }
  Scene(Scene&&) = default;

  std::size_t ObjectCount() const noexcept { return objects_.size(); }
  bool PrefersBVH() const noexcept { return prefers_bvh_; }

  /** Plain-data description of the whole scene in insertion order.  Throws std::runtime_error if
   *  the aperture objects of the lens are not contiguous, in blade order, in the object list. */
  FlatScene Flatten() const;

 private:
  Scene(std::vector<std::unique_ptr<Primitive>>&& primitives, std::vector<std::unique_ptr<Material>>&& materials,
        std::vector<Object>&& objects, std::unique_ptr<Lens>&& lens, bool prefers_bvh) noexcept
      : primitives_(std::move(primitives)), materials_(std::move(materials)), objects_(std::move(objects)),
        lens_(std::move(lens)), prefers_bvh_(prefers_bvh) {}

  std::vector<std::unique_ptr<Primitive>> primitives_;
  std::vector<std::unique_ptr<Material>> materials_;
  std::vector<Object> objects_;
  std::unique_ptr<Lens> lens_;
  bool prefers_bvh_;
};
using RGBScene = Scene;

}  // namespace scene

namespace raytracer {
// Spelling compatibility with `RGBScene::Create<raytracer::BVH<real_type, RGBObject>>` (cornel_box.cc:197)
template <typename T, typename Object> using BVH = scene::accel_tag::BVH;
template <typename T, typename Object> using List = scene::accel_tag::List;
}  // namespace raytracer

namespace etude {
/** The reference's built-in scene, src/amber/etude/cornel_box.cc:38-204. */
scene::RGBScene CornelBox(scene::real_type focal_length, scene::real_type aperture_radius, std::size_t aperture_n_blades);
}  // namespace etude

}  // namespace amber
#pragma GCC visibility pop
