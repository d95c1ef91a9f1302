// amber/prelude.h -- value types of the host object model (amber::prelude).
//
// Mirrors the names and semantics of the reference's prelude layer
// (/root/reference/include/amber/prelude/{vector3,matrix3,matrix4,ray,aabb,pixel,image}.h) as far as
// the path-tracing hot path needs them on the HOST: scene construction, bounding boxes for the BVH
// build and the returned image.  There is deliberately no host-side ray casting or sampling here:
// the only renderer in this package is the HIP engine (no CPU fallback).
//
// All arithmetic is binary32 in the reference's operation order (the build uses
// -ffp-contract=off), because values computed here (triangle normals, lens matrices, aperture
// blades) are inputs of the device path and must be bit-identical to the reference's.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <valarray>
#include <vector>

#pragma GCC visibility push(default)   // the C++ interface of the host object model is exported (bin/amber links against it)
namespace amber {

// constants.h:25-28
constexpr long double kPI = 3.141592653589793238462643383279503L;
constexpr long double kEPS = 1e-6L;
constexpr long double kDiracDelta = 1;
constexpr long double kRussianRoulette = 0.9375L;

namespace prelude {

using real_type = float;                      // rendering/forward.h:51 (std::float_t == float here)
using pixel_size_type = std::uint_fast32_t;   // prelude/forward.h:30

// vector3.h:36-65.  Binary operators are component-wise; a scalar operand is splatted
// (Vector3(const T&) is implicit in the reference and boost::field_operators builds the rest).
struct Vector3 {
  real_type x = 0, y = 0, z = 0;
  constexpr Vector3() = default;
  constexpr Vector3(real_type xyz) : x(xyz), y(xyz), z(xyz) {}
  constexpr Vector3(real_type x_, real_type y_, real_type z_) : x(x_), y(y_), z(z_) {}
  real_type X() const { return x; }
  real_type Y() const { return y; }
  real_type Z() const { return z; }
  Vector3 operator-() const { return {-x, -y, -z}; }
  Vector3& operator+=(const Vector3& v) { x += v.x; y += v.y; z += v.z; return *this; }
  Vector3& operator-=(const Vector3& v) { x -= v.x; y -= v.y; z -= v.z; return *this; }
  Vector3& operator*=(const Vector3& v) { x *= v.x; y *= v.y; z *= v.z; return *this; }
  Vector3& operator/=(const Vector3& v) { x /= v.x; y /= v.y; z /= v.z; return *this; }
};
inline Vector3 operator+(Vector3 a, const Vector3& b) { return a += b; }
inline Vector3 operator-(Vector3 a, const Vector3& b) { return a -= b; }
inline Vector3 operator*(Vector3 a, const Vector3& b) { return a *= b; }
inline Vector3 operator/(Vector3 a, const Vector3& b) { return a /= b; }
using UnitVector3 = Vector3;   // vector3.h:236-239: UnitVector3(Vector3) is a cast, never a normalisation

inline real_type Dot(const Vector3& u, const Vector3& v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
inline real_type SquaredLength(const Vector3& v) { return Dot(v, v); }
inline real_type Length(const Vector3& v) { return std::sqrt(SquaredLength(v)); }
inline Vector3 Normalize(const Vector3& v) { const real_type l = Length(v); return {v.x / l, v.y / l, v.z / l}; }
inline Vector3 Cross(const Vector3& u, const Vector3& v) {
  return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}
inline real_type Max(const Vector3& v) { return std::max({v.x, v.y, v.z}); }

// matrix3.h:36-145
struct Matrix3 {
  real_type e[9];
  Vector3 operator()(const Vector3& v) const {
    return {e[0] * v.x + e[1] * v.y + e[2] * v.z, e[3] * v.x + e[4] * v.y + e[5] * v.z, e[6] * v.x + e[7] * v.y + e[8] * v.z};
  }
  Matrix3 Inverse() const;
};
// matrix4.h:33-161 (row-major e11..e44)
struct Matrix4 {
  real_type e[16];
  Matrix4(real_type e11, real_type e12, real_type e13, real_type e14, real_type e21, real_type e22, real_type e23,
          real_type e24, real_type e31, real_type e32, real_type e33, real_type e34, real_type e41, real_type e42,
          real_type e43, real_type e44)
      : e{e11, e12, e13, e14, e21, e22, e23, e24, e31, e32, e33, e34, e41, e42, e43, e44} {}
  explicit operator Matrix3() const { return Matrix3{{e[0], e[1], e[2], e[4], e[5], e[6], e[8], e[9], e[10]}}; }
  Vector3 operator()(const Vector3& v) const {
    return {e[0] * v.x + e[1] * v.y + e[2] * v.z + e[3], e[4] * v.x + e[5] * v.y + e[6] * v.z + e[7],
            e[8] * v.x + e[9] * v.y + e[10] * v.z + e[11]};
  }
};

// aabb.h:35-171
struct AABB {
  Vector3 min, max;
  static AABB Empty() { return {Vector3(std::numeric_limits<real_type>::max()), Vector3(std::numeric_limits<real_type>::lowest())}; }
  AABB& operator+=(const AABB& b) {
    min = {std::min(min.x, b.min.x), std::min(min.y, b.min.y), std::min(min.z, b.min.z)};
    max = {std::max(max.x, b.max.x), std::max(max.y, b.max.y), std::max(max.z, b.max.z)};
    return *this;
  }
};
inline AABB operator+(AABB a, const AABB& b) { return a += b; }
inline real_type SurfaceArea(const AABB& bb) {
  const Vector3 s{bb.max.x - bb.min.x, bb.max.y - bb.min.y, bb.max.z - bb.min.z};
  return 2 * (s.x * s.y + s.y * s.z + s.z * s.x);
}

// pixel.h:32-46
struct Pixel {
  pixel_size_type x = std::numeric_limits<pixel_size_type>::max(), y = std::numeric_limits<pixel_size_type>::max();
  Pixel() = default;
  Pixel(pixel_size_type x_, pixel_size_type y_) : x(x_), y(y_) {}
  pixel_size_type X() const { return x; }
  pixel_size_type Y() const { return y; }
  explicit operator bool() const {
    return x < std::numeric_limits<pixel_size_type>::max() && y < std::numeric_limits<pixel_size_type>::max();
  }
};

// image.h:33-131: row-major x + y*W, out_of_range on bad pixels / mismatching sizes
template <typename T>
class Image {
 public:
  Image(pixel_size_type width, pixel_size_type height) : width_(width), height_(height), values_(T(0), width * height) {}
  Image& operator+=(const Image& o) {
    if (o.width_ != width_ || o.height_ != height_) throw std::out_of_range("Image::operator+=: mismatch dimensions");
    values_ += o.values_;
    return *this;
  }
  Image& operator/=(const T& d) { values_ /= d; return *this; }
  Image& operator*=(const T& m) { values_ *= m; return *this; }
  T& operator[](const Pixel& p) { return const_cast<T&>(static_cast<const Image&>(*this)[p]); }
  const T& operator[](const Pixel& p) const {
    if (p.X() >= width_ || p.Y() >= height_) throw std::out_of_range("Image::operator[]: invalid pixel given");
    return values_[p.X() + p.Y() * width_];
  }
  pixel_size_type Width() const { return width_; }
  pixel_size_type Height() const { return height_; }
  pixel_size_type Size() const { return width_ * height_; }
  T* Data() { return &values_[0]; }
  const T* Data() const { return &values_[0]; }

 private:
  pixel_size_type width_, height_;
  std::valarray<T> values_;
};

}  // namespace prelude
}  // namespace amber
#pragma GCC visibility pop
