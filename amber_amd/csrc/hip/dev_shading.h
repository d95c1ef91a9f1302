// dev_shading.h -- part of pt_device.h (included from there, in order; not a stand-alone header): materials, the thin-lens eye ray and light-path start, one bounce of the path (PathStep / PathShade).
#pragma once

namespace amber_dev {

// ---------------------------------------------------------------------------------------------
// materials (src/amber/scene/material_*.cc)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ V3 PerfectReflection(V3 incident, V3 normal, float signed_cos) {   // geometry.h:38-47
  return (2.0f * signed_cos) * normal - incident;
}
__device__ __forceinline__ V3 HemispherePSA(V3 w, uint64_t& rng) {          // sampling.h:234-265
  V3 u, v; OrthonormalBasis(w, u, v);
  const float r0 = Uniform(rng);
  const float r1 = Uniform(rng);
  const float cos_theta = Sqrt(r0);
  const float sin_theta = Sqrt(1.0f - r0);
  const float phi = 2.0f * 3.14159274f * r1;
  float sp, cp; SinCos(phi, sp, cp);
  return u * sin_theta * cp + v * sin_theta * sp + w * cos_theta;
}
__device__ __forceinline__ V3 CosinePower(V3 w, float exponent, uint64_t& rng) {   // sampling.h:267-300
  V3 u, v; OrthonormalBasis(w, u, v);
  const float r0 = Uniform(rng);
  const float r1 = Uniform(rng);
  const float cos_theta = Pow(r0, 1.0f / (exponent + 1.0f));
  const float sin_theta = Sqrt(1.0f - cos_theta * cos_theta);
  const float phi = 2.0f * 3.14159274f * r1;
  float sp, cp; SinCos(phi, sp, cp);
  return u * sin_theta * cp + v * sin_theta * sp + w * cos_theta;
}

// Scene::Radiance -> DiffuseLight::Radiance (material_diffuse_light.h:127-139)
__device__ __forceinline__ V3 Radiance(const DevMaterial& m, V3 normal, V3 dir_out) {
  if (m.kind != MAT_DIFFUSE_LIGHT) return v3(0.f, 0.f, 0.f);
  if (Dot(dir_out, normal) <= 0.0f) return v3(0.f, 0.f, 0.f);
  return ld3(m.rho);
}

// Scene::SampleLight -> Material::SampleLight (+ rho forwarders, material_basic.h:233-245, 327-338)
__device__ __forceinline__ void SampleLight(const DevMaterial& m, V3 normal, V3 dir_out, uint64_t& rng, V3& dir_in, V3& weight) {
  const V3 rho = ld3(m.rho);
  const uint32_t kind = m.kind;
#define AMBER_COS_O() Dot(dir_out, normal)
#define AMBER_MIRROR(c_) PerfectReflection(dir_out, normal, c_)
  if (kind == MAT_LAMBERTIAN || kind == MAT_PHONG) {
    // Lambertian (material_lambertian.cc:61-70, HemispherePSA sampling.h:234-265) and Phong (material_phong.cc:81-106,
    // CosinePower sampling.h:267-300) share the lobe construction -- orthonormal basis, two uniforms, sin/cos of phi,
    // the three-term combination -- and differ only in the lobe axis and in cos(theta).  One code path serves both,
    // so a wave with lanes on both materials pays for the shared part once; each lane still executes exactly the
    // operations of its own material (Phong re-samples until the direction is on the side of dir_out).
    const bool phong = kind == MAT_PHONG;
    const float signed_cos_o = AMBER_COS_O();
    const V3 w = phong ? AMBER_MIRROR(signed_cos_o) : (signed_cos_o > 0.0f ? normal : -normal);
    V3 u, v; OrthonormalBasis(w, u, v);                  // CosinePower rebuilds the same basis on every attempt
    // The reference's Phong loop re-samples forever when no direction of the lobe lies on dir_out's side (possible with a
    // normal that is not of unit length); a kernel must terminate, so attempt AMBER_PHONG_MAX_TRIES is accepted as it
    // is (the oracle does the same; with a proper normal at least half of the lobe is acceptable: probability 2^-1024).
    for (int attempt = 1;; ++attempt) {
      const float r0 = Uniform(rng);
      const float r1 = Uniform(rng);
      float cos_theta, sin_theta;
      if (phong) {
        cos_theta = Pow(r0, m.aux0);                     // r0 ^ (1 / (e + 1))
        sin_theta = Sqrt(1.0f - cos_theta * cos_theta);
      } else {
        cos_theta = Sqrt(r0);
        sin_theta = Sqrt(1.0f - r0);
      }
      const float phi = 2.0f * 3.14159274f * r1;
      float sp, cp; SinCos(phi, sp, cp);
      const V3 di = u * sin_theta * cp + v * sin_theta * sp + w * cos_theta;
      if (!phong) { dir_in = di; weight = 1.0f * rho; break; }
      const float signed_cos_i = Dot(di, normal);
      if (signed_cos_o * signed_cos_i <= 0.0f && attempt < AMBER_PHONG_MAX_TRIES) continue;
      dir_in = di;
      weight = (m.aux1 * Abs(signed_cos_i)) * rho;      // (e + 2) / (e + 1) * |cos|
      break;
    }
  } else if (kind == MAT_SPECULAR) {                     // material_specular.cc:62-70
    dir_in = AMBER_MIRROR(AMBER_COS_O());
    weight = 1.0f * rho;
  } else if (kind == MAT_REFRACTION) {                   // material_refraction.cc:177-220
    const float signed_cos_alpha = AMBER_COS_O();
    const float ior = signed_cos_alpha > 0.0f ? m.aux0 : m.param;   // 1 / ior when entering
    const float squared_cos_beta = 1.0f - (1.0f - signed_cos_alpha * signed_cos_alpha) * (ior * ior);
    const V3 dir_r = AMBER_MIRROR(signed_cos_alpha);
    if (squared_cos_beta < 0.0f) {
      dir_in = dir_r; weight = 1.0f * rho;
    } else {
      const float cos_alpha = Abs(signed_cos_alpha);
      const float cos_beta = Sqrt(squared_cos_beta);
      const V3 dir_t = (-ior) * dir_out + ((signed_cos_alpha < 0.0f ? 1.0f : -1.0f) * cos_beta + ior * signed_cos_alpha) * normal;
      // Schlick (material_refraction.cc:271-275): r0 + (1 - r0) * pow(1 - cos, 5) evaluated in double
      const float rho_r = static_cast<float>(static_cast<double>(m.r0) + static_cast<double>(1.0f - m.r0) * Pow5(1.0f - cos_alpha));
      const float rho_t = (1.0f - rho_r) * (ior * ior);
      const float rho_s = rho_r + rho_t;
      const float p_r = (rho_r / rho_s + 0.5f) / 2.0f;
      const float p_t = (rho_t / rho_s + 0.5f) / 2.0f;
      // one division for both outcomes: the lane's own operands are selected first (the same operation on the same values)
      const bool reflect = Uniform(rng) < p_r;
      dir_in = reflect ? dir_r : dir_t;
      weight = ((reflect ? rho_r : rho_t) / (reflect ? p_r : p_t)) * rho;
    }
  } else if (kind == MAT_EYE) {                          // material_eye.h:146-155
    dir_in = -dir_out; weight = v3(1.f, 1.f, 1.f);
  } else {                                               // DiffuseLight: Scatter() (material_diffuse_light.h:185-194)
    dir_in = v3(0.f, 0.f, 0.f); weight = v3(0.f, 0.f, 0.f);
  }
#undef AMBER_COS_O
#undef AMBER_MIRROR
}

// Scene::SampleImportance: identical to SampleLight for the symmetric forwarders, Eye and DiffuseLight
// (material_basic.h:340-351); BasicRefraction::SampleImportance (material_refraction.cc:222-263) drops the ior^2
// radiance scaling and uses p = (rho + 0.5) / 2.
__device__ __forceinline__ void SampleImportance(const DevMaterial& m, V3 normal, V3 dir_out, uint64_t& rng, V3& dir_in, V3& weight) {
  if (m.kind != MAT_REFRACTION) { SampleLight(m, normal, dir_out, rng, dir_in, weight); return; }
  const V3 rho = ld3(m.rho);
  const float signed_cos_alpha = Dot(dir_out, normal);
  const float ior = signed_cos_alpha > 0.0f ? m.aux0 : m.param;   // 1 / ior when entering
  const float squared_cos_beta = 1.0f - (1.0f - signed_cos_alpha * signed_cos_alpha) * (ior * ior);
  const V3 dir_r = PerfectReflection(dir_out, normal, signed_cos_alpha);
  if (squared_cos_beta < 0.0f) { dir_in = dir_r; weight = 1.0f * rho; return; }
  const float cos_alpha = Abs(signed_cos_alpha);
  const float cos_beta = Sqrt(squared_cos_beta);
  const V3 dir_t = (-ior) * dir_out + ((signed_cos_alpha < 0.0f ? 1.0f : -1.0f) * cos_beta + ior * signed_cos_alpha) * normal;
  const float rho_r = static_cast<float>(static_cast<double>(m.r0) + static_cast<double>(1.0f - m.r0) * Pow5(1.0f - cos_alpha));
  const float rho_t = 1.0f - rho_r;
  const float p_r = (rho_r + 0.5f) / 2.0f;
  const float p_t = (rho_t + 0.5f) / 2.0f;
  const bool reflect = Uniform(rng) < p_r;                 // one division for both outcomes (as in SampleLight)
  dir_in = reflect ? dir_r : dir_t;
  weight = ((reflect ? rho_r : rho_t) / (reflect ? p_r : p_t)) * rho;
}

// LightSet::GenerateRay (light_set.h:84-104) + Primitive::SampleSurfacePoint (primitive_*.cc) + HemispherePSA.
__device__ __forceinline__ void GenerateLightRay(const DevScene& sc, uint64_t& rng, V3& origin, V3& dir, V3& weight, int& origin_slot) {
  const float x = Uniform(rng) * sc.total_power;                       // prelude::Uniform(powers_.back(), sampler)
  uint32_t pos = 0;
  while (pos + 1u < sc.n_lights && sc.lights[pos].cum_power < x) ++pos; // std::lower_bound (clamped to the last light)
  const DevLight* L = sc.lights + pos;
  const uint32_t kind = L->kind;
  V3 normal;
  if (kind == PRIM_TRIANGLE) {                                          // primitive_triangle.cc:136-150
    float u = Uniform(rng), v = Uniform(rng);
    if (u + v >= 1.0f) { u = 1.0f - u; v = 1.0f - v; }
    origin = (1.0f - u - v) * ld3(L->p) + u * ld3(L->p + 3) + v * ld3(L->p + 6);
    normal = ld3(L->p + 9);
  } else if (kind == PRIM_SPHERE) {                                     // primitive_sphere.cc:115-122, SphereSA sampling.h:185-199
    const float r0 = Uniform(rng) * (1.0f - (-1.0f)) + (-1.0f);
    const float r1 = Uniform(rng);
    const float sin_theta = Sqrt(1.0f - r0 * r0);
    float sp, cp; SinCos(2.0f * 3.14159274f * r1, sp, cp);
    normal = v3(r0 * cp, r0 * sp, sin_theta);
    origin = ld3(L->p) + L->p[3] * normal;
  } else if (kind == PRIM_DISK) {                                       // primitive_disk.cc:122-136
    const float radius = Sqrt(Uniform(rng) * (L->p[6] * L->p[6]));
    const V3 N = ld3(L->p + 3);
    V3 u, v; OrthonormalBasis(N, u, v);
    float ay, ax; SinCos(Uniform(rng) * 6.28318548f, ay, ax);           // Circle: theta = Uniform<T>(2 * kPI, sampler)
    origin = ld3(L->p) + (u * ax + v * ay) * radius;
    normal = N;
  } else {                                                              // primitive_cylinder.cc:150-164
    const float height = Uniform(rng) * L->p[7];
    const V3 N = ld3(L->p + 3);
    V3 u, v; OrthonormalBasis(N, u, v);
    float ay, ax; SinCos(Uniform(rng) * 6.28318548f, ay, ax);
    const V3 n = u * ax + v * ay;
    origin = ld3(L->p) + N * height + n * L->p[6];
    normal = Normalize(n);
  }
  dir = HemispherePSA(normal, rng);
  weight = ld3(L->irr) / L->pdf_area;                                   // object.Irradiance() / PDFArea(object)
  origin_slot = L->slot;
}

// The lens record (40 dwords) is needed once per path, not per bounce: it is read from constant memory next to its
// use.  The empty asm makes the pointer opaque per use, otherwise the compiler hoists the 40 scalar loads out of the
// persistent loop, keeps them live across the whole bounce loop and spills SGPRs to VGPR lanes in the hot code.
__device__ __forceinline__ DevLens LoadLens(const DevScene& sc) {
  ConstWords w = (ConstWords)(sc.lens);
  asm volatile("" : "+s"(w));
  union { DevLens lens; uint32_t words[sizeof(DevLens) / 4]; } u;
#pragma unroll
  for (unsigned k = 0; k < sizeof(DevLens) / 4; ++k) u.words[k] = w[k];
  return u.lens;
}

// Lens::Response for Ray(position, direction_out) (scene/scene.h:299-307, lens_thin.cc:109-130, lens_pinhole.cc:70-85,
// Sensor::ResponsePixel sensor.cc:46-59).  Returns false when the ray does not reach the sensor.
__device__ __forceinline__ bool LensResponse(const DevScene& sc, V3 position, V3 direction_out, uint32_t& pixel, float& value) {
  const DevLens L = LoadLens(sc);
  const V3 direction = MatMul(L.local_, direction_out);
  float sx, sy;
  if (L.kind == 1u) {
    const V3 point = (L.sensor_distance / direction.z) * direction;
    sx = point.x; sy = point.y; value = 1.0f;
  } else {
    if (direction.z >= 0.0f) return false;
    const V3 aperture_point = MatMul(L.local_, position - ld3(L.origin));
    const V3 sensor_point = L.neg_sd_over_fd * aperture_point + (L.sensor_distance / direction.z) * direction;
    sx = sensor_point.x; sy = sensor_point.y;
    value = static_cast<float>(Pow4(Normalize(sensor_point - aperture_point).z / direction.z));
  }
  const float uvx = sx / sc.sensor.sw + 0.5f, uvy = sy / sc.sensor.sh + 0.5f;
  const float mn = uvy < uvx ? uvy : uvx, mx = uvx < uvy ? uvy : uvx;  // std::min / std::max of (x, y)
  if (mn < 0.0f || mx >= 1.0f) return false;
  uint32_t ix = static_cast<uint32_t>(uvx * sc.sensor.wf), iy = static_cast<uint32_t>(uvy * sc.sensor.hf);
  if (ix > sc.sensor.w - 1u) ix = sc.sensor.w - 1u;
  if (iy > sc.sensor.h - 1u) iy = sc.sensor.h - 1u;
  pixel = ix + iy * sc.sensor.w;
  return true;
}

// ---------------------------------------------------------------------------------------------
// eye ray: BasicThin::GenerateRay (lens_thin.cc:70-107) + Sensor::PixelBound::Uniform
// (sensor.cc:111-120, jitter draw order Y then X -- the g++ order the reference outputs were made with)
// ---------------------------------------------------------------------------------------------
// near_edge (optional): the aperture sample lies within DevLens.edge_tol (barycentric) of its blade's boundary -- only then can the exact
// test of ANOTHER blade accept the ray's own origin (pt_megakernel's primary rounds: which blades are candidates).
__device__ __forceinline__ void GenerateEyeRay(const DevScene& sc, uint32_t px, uint32_t py, uint64_t& rng,
                                               V3& origin, V3& dir, float& weight, int& origin_slot, bool* near_edge = nullptr) {
  const DevLens L = LoadLens(sc);
  if (L.kind == 1u) {                                      // BasicPinhole::GenerateRay lens_pinhole.cc:48-68
    const float jy = Uniform(rng);
    const float jx = Uniform(rng);
    const float uvx = (static_cast<float>(px) + jx) / sc.sensor.wf;
    const float uvy = (static_cast<float>(py) + jy) / sc.sensor.hf;
    const V3 sensor_point = v3((uvx - 0.5f) * sc.sensor.sw, (uvy - 0.5f) * sc.sensor.sh, L.sensor_distance);
    const V3 ray_dir = Normalize(MatMul(L.global_, -sensor_point));
    // PDFDirection lens_pinhole.cc:93-106 (binary32 throughout; no sensor.Size() factor, unlike the thin lens)
    const V3 dl = MatMul(L.local_, ray_dir);
    const V3 point = (L.sensor_distance / dl.z) * dl;
    const float geometry_factor = dl.z * dl.z / SquaredLength(point);
    const float pdf_dir = L.inv_scene_area / geometry_factor;
    origin = ld3(L.origin); dir = ray_dir;
    weight = 1.0f / 1.0f / pdf_dir;                        // 1 / PDFArea (= kDiracDelta) / PDFDirection
    origin_slot = -1;
    if (near_edge) *near_edge = true;                      // (the pinhole's degenerate blade: keep every blade bit)
    return;
  }
  const float fpos = __builtin_floorf(Uniform(rng) * L.n_blades_f);
  uint32_t pos = static_cast<uint32_t>(fpos);
  if (pos > L.n_blades - 1) pos = L.n_blades - 1;
  const DevBlade* bl = sc.blades + pos;
  float u = Uniform(rng);
  float v = Uniform(rng);
  if (u + v >= 1.0f) { u = 1.0f - u; v = 1.0f - v; }
  if (near_edge) *near_edge = !(u >= L.edge_tol && v >= L.edge_tol && u + v <= 1.0f - L.edge_tol);
  const V3 ap_origin = (1.0f - u - v) * ld3(bl->v0) + u * ld3(bl->v1) + v * ld3(bl->v2);   // primitive_triangle.cc:136-150
  const V3 aperture_point = MatMul(L.local_, ap_origin - ld3(L.origin));
  const float jy = Uniform(rng);
  const float jx = Uniform(rng);
  const float uvx = (static_cast<float>(px) + jx) / sc.sensor.wf;
  const float uvy = (static_cast<float>(py) + jy) / sc.sensor.hf;
  const V3 sensor_point = v3((uvx - 0.5f) * sc.sensor.sw, (uvy - 0.5f) * sc.sensor.sh, L.sensor_distance);
  const V3 direction = Normalize(L.neg_fd_over_sd * sensor_point - aperture_point);
  const double factor = Pow4(Normalize(sensor_point - aperture_point).z / direction.z);
  const V3 ray_dir = Normalize(MatMul(L.global_, direction));
  const V3 dloc = MatMul(L.local_, ray_dir);
  const float pdf_dir = static_cast<float>(static_cast<double>(L.size_over_area) * L.sd2 / Pow4(dloc.z));
  origin = ap_origin; dir = ray_dir;
  origin_slot = bl->slot;                                  // the eye ray starts ON this aperture triangle
  weight = static_cast<float>(factor / static_cast<double>(L.p_area) / static_cast<double>(pdf_dir));
}

// ---------------------------------------------------------------------------------------------
// one bounce of PathTracing::Thread::Render (algorithm_pt.cc:137-157).
// Returns true if the path continues (o, d, weight updated), false if it ended.
// ---------------------------------------------------------------------------------------------
struct Bounce { int object; float t; V3 pos; V3 weight_before; };

// kLight = false: PathTracing::Thread::Render (algorithm_pt.cc:137-157).  kLight = true: LightTracing::Thread::Render
// (algorithm_lt.cc:134-162): a hit on an Eye surface splats weight * response / image.Size() instead of collecting
// emitted radiance, and the material is sampled with SampleImportance.
struct SplatSink { DevSplat* records; unsigned int* count; uint32_t capacity; uint32_t path, sample; float size_f; };

template <bool kTrace, int kEngine, bool kLight>
__device__ __forceinline__ bool PathShade(const DevScene& sc, const DevObject* lds_objects, const HitRec& h, V3& o, V3& d, V3& weight, V3& measurement,
                                          uint64_t& rng, uint32_t& casts, int& origin_slot, Bounce* trace AMBER_STAMP_PARAM, const SplatSink* sink);

template <bool kTrace, int kEngine, bool kLight = false>
__device__ __forceinline__ bool PathStep(const DevScene& sc, const DevObject* lds_objects, int32_t* lds_stack, V3& o, V3& d, V3& weight, V3& measurement,
                                         uint64_t& rng, uint32_t& casts, int& origin_slot, Bounce* trace AMBER_STAMP_PARAM, const SplatSink* sink = nullptr,
                                         const bool use_premask = false, const uint32_t premask = 0u, const int bvh_stack_cap = AMBER_BVH_STACK) {
  HitRec h;
  ClosestHit<kEngine>(sc, lds_objects, lds_stack, o, d, origin_slot, h AMBER_STAMP_ARG, use_premask, premask, bvh_stack_cap);
  return PathShade<kTrace, kEngine, kLight>(sc, lds_objects, h, o, d, weight, measurement, rng, casts, origin_slot, trace AMBER_STAMP_ARG, sink);
}

// Everything of a bounce after the closest-hit query (algorithm_pt.cc:140-157): h is the result of Scene::Cast.
template <bool kTrace, int kEngine, bool kLight>
__device__ __forceinline__ bool PathShade(const DevScene& sc, const DevObject* lds_objects, const HitRec& h, V3& o, V3& d, V3& weight, V3& measurement,
                                          uint64_t& rng, uint32_t& casts, int& origin_slot, Bounce* trace AMBER_STAMP_PARAM, const SplatSink* sink) {
  casts++;
  if (h.idx < 0) {
    if (kTrace) { trace->object = -1; trace->t = __builtin_nanf(""); trace->pos = v3(0, 0, 0); trace->weight_before = v3(0, 0, 0); }
    return false;
  }
  V3 pos, normal; uint32_t mat;
  ResolveHit<kEngine == ENGINE_TWO_PHASE_N ? 0x7fu : 0xffu>((kEngine == ENGINE_TWO_PHASE || kEngine == ENGINE_TWO_PHASE_N) ? lds_objects : ((kEngine == ENGINE_BVH || kEngine == ENGINE_REF_BVH) ? sc.bvh_objects : sc.objects), h, o, d, pos, normal, mat);
  const DevMaterial m = sc.materials[mat];
  const V3 dir_out = -d;
  if (kTrace) { trace->object = h.idx; trace->t = h.t; trace->pos = pos; trace->weight_before = weight; }
  V3 dir_in, sw;
  if (kLight) {
    if (m.kind == MAT_EYE) {                                                 // algorithm_lt.cc:141-147
      uint32_t pixel; float value;
      if (LensResponse(sc, pos, dir_out, pixel, value)) {
        const V3 add = (weight * value) / sink->size_f;
        const unsigned int k = atomicAdd(sink->count, 1u);
        if (k < sink->capacity) {
          DevSplat& r = sink->records[k];
          r.path = sink->path; r.sample = sink->sample; r.bounce = casts; r.pixel = pixel;
          r.rgb[0] = add.x; r.rgb[1] = add.y; r.rgb[2] = add.z; r.pad = 0u;
        }
      }
    }
    AMBER_STAMP(4);
    SampleImportance(m, normal, dir_out, rng, dir_in, sw);
  } else {
    measurement = measurement + weight * Radiance(m, normal, dir_out);      // algorithm_pt.cc:144
    AMBER_STAMP(4);
    SampleLight(m, normal, dir_out, rng, dir_in, sw);                        // :145-146
  }
  AMBER_STAMP(5);
  float p_rr = 0.9375f;                                                      // std::min<real_type>(kRussianRoulette, Max(w)) :148-149
  const float mw = Max3(sw);
  if (mw < p_rr) p_rr = mw;
  if (Uniform(rng) >= p_rr) return false;                                    // :151-153
  if (sc.max_depth && casts >= sc.max_depth) return false;                   // build-side extension (BASELINE config 5)
  o = pos; d = dir_in;                                                       // :155 Ray(pos, UnitVector3) -- no renormalisation
  if (kEngine == ENGINE_TWO_PHASE_N) origin_slot = (lds_objects[h.slot].kind & 0x80u) ? h.slot : -1;      // a filtered triangle of its group (bit 7 of the LDS record)
  else origin_slot = (kEngine == ENGINE_TWO_PHASE && static_cast<uint32_t>(h.slot) < sc.n_prog_tris) ? h.slot : -1;
#if AMBER_SHARED_WEIGHT_QUOTIENT
  {
    // :156  weight *= scatter.Weight() / p -- three binary32 divisions by the same p.  A component of the scatter weight that has the
    // bits of the first one has the first one's quotient, and +0 / p is +0 (p > 0 here: a path with p = 0 has ended above): grey, white,
    // mirror, glass and single-channel materials need ONE division.  The other two run only if some lane of the wave holds a weight
    // that is neither (wave-uniform branch; the same IEEE quotients either way).
    const uint32_t bx = __float_as_uint(sw.x), by = __float_as_uint(sw.y), bz = __float_as_uint(sw.z);
    const float qx = sw.x / p_rr;
    float qy = by == bx ? qx : 0.0f, qz = bz == bx ? qx : 0.0f;
    const bool hard_y = !(by == bx || (by == 0u && p_rr > 0.0f)), hard_z = !(bz == bx || (bz == 0u && p_rr > 0.0f));
    if (__ballot(hard_y || hard_z) != 0ull) {
      if (hard_y) qy = sw.y / p_rr;
      if (hard_z) qz = sw.z / p_rr;
    }
    weight = weight * v3(qx, qy, qz);
  }
#else
  weight = weight * (sw / p_rr);                                             // :156
#endif
  return true;
}

}  // namespace amber_dev
