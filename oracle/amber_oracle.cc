/*
 * amber_oracle.cc -- CPU ORACLE (test infrastructure, NOT product code; see amber_oracle.h).
 *
 * A from-scratch restatement of the reference's unidirectional path tracer and everything it
 * calls.  All arithmetic is IEEE binary32 with the reference's operation order, no FMA
 * contraction (build: -ffp-contract=off, no -mfma), double / long double only where the
 * reference's own expressions promote (noted per function).  File:line citations are relative
 * to /root/reference.
 *
 * Two knobs that the reference does not have, both needed to compare against a GPU:
 *   math  = LIBM      glibc sinf/cosf/powf/pow exactly as the reference calls them
 *         = PORTABLE  sin/cos/pow built from + - * / sqrt and integer ops only, so that the
 *                     same sequence is bit-identical on x86 and gfx950
 *   accel = BVH       the reference's SAH BVH (build + recursive ordered traversal + SSE slab)
 *         = LIST      the reference's brute-force List acceleration (acceleration_list.h:51-68)
 */
#include "amber_oracle.h"

#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <functional>
#include <future>
#include <limits>
#include <memory>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------
// constants  (include/amber/constants.h:25-28 -- all long double in the reference)
// ------------------------------------------------------------------------------------------
const long double kPI = 3.141592653589793238462643383279503L;
const long double kEPS = 1e-6L;
const long double kRussianRoulette = 0.9375L;
const float kPIf = static_cast<float>(kPI);

// ------------------------------------------------------------------------------------------
// Vector3 (include/amber/prelude/vector3.h:36-342).  boost::field_operators generates
// component-wise + - * / from the compound forms; scalars convert through Vector3(const T&).
// ------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 splat(float s) { return V3{s, s, s}; }                        // vector3.h:163-168
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }   // :198-203
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }   // :205-210
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }   // :212-217
inline V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }   // :219-224
inline V3 operator*(float s, V3 v) { return splat(s) * v; }
inline V3 operator*(V3 v, float s) { return v * splat(s); }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }              // :191-196
inline float Dot(V3 u, V3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }        // :290-295
inline float SquaredLength(V3 v) { return Dot(v, v); }
inline float Length(V3 v) { return std::sqrt(SquaredLength(v)); }
inline V3 Normalize(V3 v) { const float l = Length(v); return V3{v.x / l, v.y / l, v.z / l}; }  // :311-317
inline V3 Cross(V3 u, V3 v) {                                           // :319-328
  return V3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}
inline float Max3(V3 v) { return std::max({v.x, v.y, v.z}); }           // :276-281
inline void OrthonormalBasis(V3 w, V3& u, V3& v) {                      // :330-342
  u = Normalize(Cross(w, std::abs(w.x) < std::abs(w.y) ? v3(1, 0, 0) : v3(0, 1, 0)));
  v = Normalize(Cross(w, u));
}

struct Ray { V3 o, d; };
struct Hit {                                                            // hit.h:61-94
  V3 pos{0, 0, 0}, n{0, 0, 0};
  float t = std::numeric_limits<float>::quiet_NaN();
  explicit operator bool() const { return std::isfinite(t); }
};
inline Hit MakeHitN(V3 p, V3 n_raw, float t) { Hit h; h.pos = p; h.n = Normalize(n_raw); h.t = t; return h; }  // hit.h:68-77
inline Hit MakeHitU(V3 p, V3 n_unit, float t) { Hit h; h.pos = p; h.n = n_unit; h.t = t; return h; }          // hit.h:79-88

// ------------------------------------------------------------------------------------------
// math modes
// ------------------------------------------------------------------------------------------
// PORTABLE sin/cos: Cody-Waite 3-term reduction by multiples of pi/4 and the Cephes
// single-precision minimax polynomials (published algorithm: S. Moshier, Cephes sinf.c/cosf.c),
// written with separate mul/add only.  Valid for 0 <= phi <= ~8 (the path only needs [0, 2pi]).
void PortableSinCos(float x, float* s_out, float* c_out) {
  const float FOPI = 1.27323954473516f;         // 4/pi
  const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
  int j = static_cast<int>(FOPI * x);           // x >= 0
  j += (j & 1);                                 // map to even octant boundary
  const float y = static_cast<float>(j);
  const float r = ((x - y * DP1) - y * DP2) - y * DP3;
  const float z = r * r;
  // sin polynomial on [-pi/4, pi/4]
  float ps = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * r + r;
  // cos polynomial on [-pi/4, pi/4]
  float pc = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z
             - 0.5f * z + 1.0f;
  const int q = (j >> 1) & 3;                   // quadrant: angle = r + q*pi/2
  float s, c;
  switch (q) {
    case 0: s = ps; c = pc; break;
    case 1: s = pc; c = -ps; break;
    case 2: s = -ps; c = -pc; break;
    default: s = -pc; c = ps; break;
  }
  *s_out = s; *c_out = c;
}

inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// PORTABLE pow(x, y) for x >= 0: exp2(y * log2(x)) from + - * / and integer ops only.
// log: atanh series on m in [sqrt(1/2), sqrt(2)); exp: Taylor in t = g*ln2, |g| <= 1/2.
float PortablePow(float x, float y) {
  if (y == 0.0f) return 1.0f;
  if (x == 0.0f) return y > 0.0f ? 0.0f : std::numeric_limits<float>::infinity();
  if (x == 1.0f) return 1.0f;
  uint32_t bits = f2u(x);
  int e = static_cast<int>((bits >> 23) & 0xff);
  if (e == 0) {  // subnormal: scale up by 2^24 exactly
    x = x * 16777216.0f; bits = f2u(x); e = static_cast<int>((bits >> 23) & 0xff) - 24;
  }
  e -= 127;
  float m = u2f((bits & 0x007fffffu) | 0x3f800000u);    // [1, 2)
  if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
  const float f = m - 1.0f;
  const float s = f / (2.0f + f);
  const float z = s * s;
  // ln(m) = 2s (1 + z/3 + z^2/5 + z^3/7 + z^4/9 + z^5/11)
  float p = 0.0909090909f;
  p = p * z + 0.111111111f;
  p = p * z + 0.142857143f;
  p = p * z + 0.2f;
  p = p * z + 0.333333333f;
  p = p * z + 1.0f;
  const float ln_m = 2.0f * s * p;
  const float log2x = static_cast<float>(e) + ln_m * 1.44269504f;
  const float w = y * log2x;
  if (w >= 128.0f) return std::numeric_limits<float>::infinity();
  if (w < -149.0f) return 0.0f;
  const float nf = std::floor(w + 0.5f);
  const float g = w - nf;
  const float t = g * 0.693147181f;
  float q = 1.98412698e-4f;               // 1/5040
  q = q * t + 1.38888889e-3f;             // 1/720
  q = q * t + 8.33333333e-3f;             // 1/120
  q = q * t + 4.16666667e-2f;             // 1/24
  q = q * t + 1.66666667e-1f;             // 1/6
  q = q * t + 0.5f;
  q = q * t + 1.0f;
  q = q * t + 1.0f;
  int n = static_cast<int>(nf);
  // scale by 2^n in two exact steps (keeps subnormal results correctly rounded once)
  if (n < -126) { q = q * u2f(static_cast<uint32_t>(n + 126 + 127) << 23); n = -126; }
  return q * u2f(static_cast<uint32_t>(n + 127) << 23);
}


// GLIBC mode: restatement of glibc 2.35's binary32 sincosf and powf (sysdeps/ieee754/flt-32/s_sincosf.c,
// sysdeps/x86_64/fpu/sincosf_poly.h, s_sincosf_data.c; e_powf.c, e_powf_log2_data.c, e_exp2f_data.c -- the
// algorithms and tables of ARM's optimized-routines, Szabolcs Nagy / Wilco Dijkstra), which is what the reference's
// std::sin / std::cos (merged into ONE sincosf call by g++ -O2, sampling.h:249-250) and std::pow(float, float)
// (sampling.h:279) execute.  libm is a third-party dependency of the reference that cannot travel as source; this is
// its published algorithm in the form glibc's x86-64 FMA ifunc variant executes it (the variant every FMA-capable CPU
// selects: this container's Xeon and the GPU box's EPYC 9575F): double-precision kernels in which EVERY a*b+c of the
// source is one fused multiply-add, one rounding to binary32 at the end.  The tables are the values of libm.so.6's
// .rodata.  tests/test_math_modes.py proves the restatement equal to the live libm for every argument the path can
// produce (all 2^24 phi = 2 pi u, all 2^24 r0 for the scene's Phong exponents) and on 1e8 random arguments.
// Domain: sincos |x| < 120 (the path needs [0, 2 pi]; beyond that glibc switches to a 192-bit reduction, not restated:
// NaN is returned); pow: every binary32 pair.
struct GlibcSinCosTab { double sign[4], hpi_inv, hpi, c0, c1, s1, c2, s2, c3, s3, c4; };
static const GlibcSinCosTab kGlibcSinCos[2] = {
  {{1.0, -1.0, -1.0, 1.0}, 0x1.45f306dc9c883p+23, 0x1.921fb54442d18p+0, 0x1p0, -0x1.ffffffd0c621cp-2, -0x1.555545995a603p-3,
   0x1.55553e1068f19p-5, 0x1.1107605230bc4p-7, -0x1.6c087e89a359dp-10, -0x1.994eb3774cf24p-13, 0x1.99343027bf8c3p-16},
  {{1.0, -1.0, -1.0, 1.0}, 0x1.45f306dc9c883p+23, 0x1.921fb54442d18p+0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.555545995a603p-3,
   -0x1.55553e1068f19p-5, 0x1.1107605230bc4p-7, 0x1.6c087e89a359dp-10, -0x1.994eb3774cf24p-13, -0x1.99343027bf8c3p-16},
};
#define ORACLE_FMA_TARGET __attribute__((target("fma")))
// sincosf_poly (sysdeps/x86_64/fpu/sincosf_poly.h): both polynomials at once, n odd swaps the outputs
ORACLE_FMA_TARGET static inline void GlibcSinCosPoly(double x, double x2, const GlibcSinCosTab* p, int n, float* sinp, float* cosp) {
  const double s1 = __builtin_fma(x2, p->s3, p->s2);
  const double c2 = __builtin_fma(x2, p->c4, p->c3);
  const double c1 = __builtin_fma(x2, p->c1, p->c0);
  const double x3 = x2 * x, x4 = x2 * x2;
  const double x5 = x2 * x3, x6 = x2 * x4;
  const double s = __builtin_fma(x3, p->s1, x);
  const double c = __builtin_fma(x4, p->c2, c1);
  const float sv = static_cast<float>(__builtin_fma(x5, s1, s));
  const float cv = static_cast<float>(__builtin_fma(x6, c2, c));
  if (n & 1) { *cosp = sv; *sinp = cv; } else { *sinp = sv; *cosp = cv; }
}
ORACLE_FMA_TARGET void GlibcSinCos(float y, float* sinp, float* cosp) {     // s_sincosf.c:__sincosf
  double x = y;
  const GlibcSinCosTab* p = &kGlibcSinCos[0];
  const uint32_t top = (f2u(y) >> 20) & 0x7ffu;                 // abstop12
  if (top < 0x3f4u) {                                           // |y| < pi/4
    const double x2 = x * x;
    if (top < 0x398u) { *sinp = y; *cosp = 1.0f; return; }      // |y| < 2^-12
    GlibcSinCosPoly(x, x2, p, 0, sinp, cosp);
  } else if (top < 0x42fu) {                                    // |y| < 120: reduce_fast
    const double r = x * p->hpi_inv;
    const int n = (static_cast<int32_t>(r) + 0x800000) >> 24;   // round to nearest multiple of pi/2
    x = __builtin_fma(-static_cast<double>(n), p->hpi, x);
    const double s = p->sign[n & 3];
    if (n & 2) p = &kGlibcSinCos[1];
    GlibcSinCosPoly(x * s, x * x, p, n, sinp, cosp);
  } else {
    *sinp = *cosp = std::numeric_limits<float>::quiet_NaN();    // outside the restated domain
  }
}
static const double kGlibcLog2Tab[16][2] = {   // __powf_log2_data.tab: {invc, logc}
  {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
  {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
  {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
  {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
  {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
  {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
  {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
  {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
static const double kGlibcLog2Poly[5] = {0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2, 0x1.ec70a6ca7baddp-2, -0x1.7154748bef6c8p-1, 0x1.71547652ab82bp0};
static const uint64_t kGlibcExp2Tab[32] = {    // __exp2f_data.tab
  0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa,
  0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
  0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74,
  0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
  0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
  0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};
static const double kGlibcExp2Shift = 0x1.8p+52 / 32;                         // shift_scaled
static const double kGlibcExp2Poly[3] = {0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1};
inline uint64_t d2u(double d) { uint64_t u; std::memcpy(&u, &d, 8); return u; }
inline double u2d(uint64_t u) { double d; std::memcpy(&d, &u, 8); return d; }
static inline int GlibcCheckInt(uint32_t iy) {                                // e_powf.c:checkint: 0 not int, 1 odd, 2 even
  const int e = iy >> 23 & 0xff;
  if (e < 0x7f) return 0;
  if (e > 0x7f + 23) return 2;
  if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
  if (iy & (1u << (0x7f + 23 - e))) return 1;
  return 2;
}
static inline bool GlibcZeroInfNan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000 - 1; }
ORACLE_FMA_TARGET float GlibcPow(float x, float y) {                          // e_powf.c:__powf
  uint32_t sign_bias = 0;
  uint32_t ix = f2u(x);
  const uint32_t iy = f2u(y);
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || GlibcZeroInfNan(iy)) {
    if (GlibcZeroInfNan(iy)) {
      if (2 * iy == 0) return 1.0f;
      if (ix == 0x3f800000u) return 1.0f;
      if (2 * ix > 2u * 0x7f800000 || 2 * iy > 2u * 0x7f800000) return x + y;
      if (2 * ix == 2 * 0x3f800000u) return 1.0f;
      if ((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;      // |x| < 1 && y == inf or |x| > 1 && y == -inf
      return y * y;
    }
    if (GlibcZeroInfNan(ix)) {
      float x2 = x * x;
      if ((ix & 0x80000000u) && GlibcCheckInt(iy) == 1) x2 = -x2;
      return (iy & 0x80000000u) ? 1 / x2 : x2;
    }
    if (ix & 0x80000000u) {                                                   // finite x < 0
      const int yint = GlibcCheckInt(iy);
      if (yint == 0) return std::numeric_limits<float>::quiet_NaN();
      if (yint == 1) sign_bias = 1u << 16;
      ix &= 0x7fffffffu;
    }
    if (ix < 0x00800000u) { ix = f2u(u2f(ix) * 0x1p23f); ix &= 0x7fffffffu; ix -= 23u << 23; }   // subnormal x
  }
  // log2_inline
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (tmp >> 19) % 16;
  const uint32_t top = tmp & 0xff800000u;
  const uint32_t iz = ix - top;
  const int k = static_cast<int32_t>(top) >> 23;
  const double z = u2f(iz);
  const double r = __builtin_fma(z, kGlibcLog2Tab[i][0], -1.0);
  const double y0 = kGlibcLog2Tab[i][1] + static_cast<double>(k);
  const double* A = kGlibcLog2Poly;
  const double r2 = r * r;
  double yy = __builtin_fma(A[0], r, A[1]);
  const double pp = __builtin_fma(A[2], r, A[3]);
  const double r4 = r2 * r2;
  double q = __builtin_fma(A[4], r, y0);
  q = __builtin_fma(pp, r2, q);
  yy = __builtin_fma(yy, r4, q);
  const double ylogx = static_cast<double>(y) * yy;
  if ((d2u(ylogx) >> 47 & 0xffff) >= (d2u(126.0) >> 47)) {                    // |y * log2(x)| >= 126
    const float sgn = sign_bias ? -1.0f : 1.0f;
    if (ylogx > 0x1.fffffffd1d571p+6) return sgn * std::numeric_limits<float>::infinity();
    if (ylogx <= -150.0) return sgn * 0.0f;
    if (ylogx < -149.0) return sgn * 0x1p-149f;                               // __math_may_uflowf: 0x1.4p-75f squared
  }
  // exp2_inline
  double kd = ylogx + kGlibcExp2Shift;
  const uint64_t ki = d2u(kd);
  kd -= kGlibcExp2Shift;
  const double rr = ylogx - kd;
  uint64_t t = kGlibcExp2Tab[ki % 32];
  t += (ki + sign_bias) << (52 - 5);
  const double s = u2d(t);
  const double* C = kGlibcExp2Poly;
  const double zz = __builtin_fma(C[0], rr, C[1]);
  const double rr2 = rr * rr;
  double e = __builtin_fma(C[2], rr, 1.0);
  e = __builtin_fma(zz, rr2, e);
  return static_cast<float>(e * s);
}
// x^4, x^5 of a binary32 x as glibc's double pow returns them -- up to that function's own rounding slips: d*d is exact
// (24 + 24 bits), (d*d)^2 is therefore the correctly rounded x^4, and x^5 is formed from the exact (hi, lo) pair of
// x^4 so that it too rounds once.  Live pow() differs from these in the last bit of the double for 1e-3 of the
// arguments (it is not correctly rounded), which never survived the conversion to binary32 in 2e8 trials.
ORACLE_FMA_TARGET double GlibcPow4(float x) { const double d = x, d2 = d * d; return d2 * d2; }
ORACLE_FMA_TARGET double GlibcPow5(float x) {
  const double d = x, d2 = d * d;
  const double h = d2 * d2, l = __builtin_fma(d2, d2, -h);     // x^4 = h + l exactly
  const double p = h * d, pl = __builtin_fma(h, d, -p);        // h * d = p + pl exactly
  return p + __builtin_fma(l, d, pl);
}

struct Math {
  int mode;
  void sincos(float phi, float& s, float& c) const {
    if (mode == ORACLE_MATH_LIBM) { c = std::cos(phi); s = std::sin(phi); }   // sampling.h:248-249 (cosf/sinf)
    else if (mode == ORACLE_MATH_GLIBC) GlibcSinCos(phi, &s, &c);
    else PortableSinCos(phi, &s, &c);
  }
  float powf_(float x, float y) const {                                       // sampling.h:279 (powf)
    return mode == ORACLE_MATH_LIBM ? std::pow(x, y) : mode == ORACLE_MATH_GLIBC ? GlibcPow(x, y) : PortablePow(x, y);
  }
  // std::pow(float, int) promotes to double pow (C++11 [c.math]); used with exponents 2, 4, 5.
  double pow_i(float x, int n) const {
    if (mode == ORACLE_MATH_LIBM) return std::pow(x, n);
    if (mode == ORACLE_MATH_GLIBC && n == 5) return GlibcPow5(x);
    const double d = x, d2 = d * d;            // exact (24+24 bits)
    if (n == 2) return d2;
    const double d4 = d2 * d2;                 // one rounding
    if (n == 4) return d4;
    return d4 * d;                             // n == 5
  }
};

// ------------------------------------------------------------------------------------------
// Samplers (include/amber/prelude/sampling.h:35-57, 148-175)
// ------------------------------------------------------------------------------------------
struct Sampler { virtual double operator()() = 0; virtual ~Sampler() {} };

// GenericSampler<std::mt19937_64>: uniform_real_distribution<double>(0,1)(engine) in libstdc++
// is generate_canonical<double,53> = double(x) / 2^64, clamped below 1 (verified in
// tests/test_oracle_pin.py against std::uniform_real_distribution on this toolchain).
struct MTSampler : Sampler {
  std::mt19937_64 engine;
  explicit MTSampler(uint64_t seed) : engine(seed) {}
  double operator()() override {
    double r = static_cast<double>(engine()) * 0x1p-64;
    if (r >= 1.0) r = std::nextafter(1.0, 0.0);
    return r;
  }
};

// per-(pixel,sample) XorShift sampler (the north-star's "XorShift sampler"; the reference has
// none -- SURVEY.md section 0).  Spec shared with the product: splitmix64-hashed seed, Marsaglia
// xorshift64 (13,7,17), uniform = top 24 bits * 2^-24 (exact in float, never 1.0).
inline uint64_t SplitMix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline uint64_t XorShiftSeed(uint64_t global_seed, uint32_t pixel, uint32_t sample) {
  const uint64_t key = (static_cast<uint64_t>(pixel) << 32) | sample;
  uint64_t s = SplitMix64(SplitMix64(global_seed) ^ key);
  return s ? s : 0x9E3779B97F4A7C15ull;
}
struct XorShiftSampler : Sampler {
  uint64_t s;
  explicit XorShiftSampler(uint64_t state) : s(state) {}
  double operator()() override {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return static_cast<double>(s >> 40) * 0x1p-24;
  }
};
struct ArraySampler : Sampler {   // feeds fixed uniforms (known-answer tests)
  const double* u; uint32_t n, i = 0;
  ArraySampler(const double* u_, uint32_t n_) : u(u_), n(n_) {}
  double operator()() override { return i < n ? u[i++] : (i++, 0.5); }
};

inline float UniformF(Sampler& s) { return static_cast<float>(s()); }          // sampling.h:156-161
inline float UniformF(float max, Sampler& s) { return static_cast<float>(s()) * max; }  // :163-168

// sampling.h:234-265 (HemispherePSA), 267-300 (CosinePower); results are casts, not renormalised.
V3 HemispherePSA(V3 w, Sampler& smp, const Math& M) {
  V3 u, v; OrthonormalBasis(w, u, v);
  const float r0 = UniformF(smp);
  const float r1 = UniformF(smp);
  const float cos_theta = std::sqrt(r0);
  const float sin_theta = std::sqrt(1 - r0);
  const float phi = 2 * kPIf * r1;
  float sp, cp; M.sincos(phi, sp, cp);
  return u * sin_theta * cp + v * sin_theta * sp + w * cos_theta;
}
V3 CosinePower(V3 w, float exponent, Sampler& smp, const Math& M) {
  V3 u, v; OrthonormalBasis(w, u, v);
  const float r0 = UniformF(smp);
  const float r1 = UniformF(smp);
  const float cos_theta = M.powf_(r0, 1 / (exponent + 1));
  const float sin_theta = std::sqrt(1 - cos_theta * cos_theta);
  const float phi = 2 * kPIf * r1;
  float sp, cp; M.sincos(phi, sp, cp);
  return u * sin_theta * cp + v * sin_theta * sp + w * cos_theta;
}
inline V3 PerfectReflection(V3 incident, V3 normal, float signed_cos) {   // geometry.h:38-47
  return 2 * signed_cos * normal - incident;
}

// ------------------------------------------------------------------------------------------
// AABB (include/amber/prelude/aabb.h:35-171, src/amber/prelude/aabb.cc:28-62)
// ------------------------------------------------------------------------------------------
struct AABB { V3 mn, mx; };
inline AABB EmptyBox() { return AABB{splat(FLT_MAX), splat(-FLT_MAX)}; }
inline AABB Union(const AABB& a, const AABB& b) {                        // aabb.h:117-132
  return AABB{v3(std::min(a.mn.x, b.mn.x), std::min(a.mn.y, b.mn.y), std::min(a.mn.z, b.mn.z)),
              v3(std::max(a.mx.x, b.mx.x), std::max(a.mx.y, b.mx.y), std::max(a.mx.z, b.mx.z))};
}
inline float SurfaceArea(const AABB& bb) {                              // aabb.h:164-171
  const V3 s = v3(bb.mx.x - bb.mn.x, bb.mx.y - bb.mn.y, bb.mx.z - bb.mn.z);
  return 2 * (s.x * s.y + s.y * s.z + s.z * s.x);
}
// SSE semantics: _mm_min_ps(a,b) = a < b ? a : b ; _mm_max_ps(a,b) = a > b ? a : b (b on NaN)
inline float sse_min(float a, float b) { return a < b ? a : b; }
inline float sse_max(float a, float b) { return a > b ? a : b; }
bool SlabTest(const AABB& bb, const Ray& ray, float t_max, float& t_in, float& t_out) {  // aabb.cc:28-62
  float t_min = 0;
  const float ix = 1.0f / ray.d.x, iy = 1.0f / ray.d.y, iz = 1.0f / ray.d.z;
  const float t0x = (bb.mn.x - ray.o.x) * ix, t0y = (bb.mn.y - ray.o.y) * iy, t0z = (bb.mn.z - ray.o.z) * iz;
  const float t1x = (bb.mx.x - ray.o.x) * ix, t1y = (bb.mx.y - ray.o.y) * iy, t1z = (bb.mx.z - ray.o.z) * iz;
  t_min = std::max({t_min, sse_min(t0x, t1x), sse_min(t0y, t1y), sse_min(t0z, t1z)});
  t_max = std::min({t_max, sse_max(t0x, t1x), sse_max(t0y, t1y), sse_max(t0z, t1z)});
  t_in = t_min; t_out = t_max;
  return t_min <= t_max;
}

// ------------------------------------------------------------------------------------------
// Primitives (src/amber/scene/primitive_*.cc)
// ------------------------------------------------------------------------------------------
struct Object {
  uint32_t kind = 0, material = 0;
  V3 a{0, 0, 0}, b{0, 0, 0}, c{0, 0, 0};   // triangle: v0 v1 v2 ; sphere: a=center ; disk/cyl: a=center b=normal
  float radius = 0, height = 0;
  V3 normal{0, 0, 0};                       // triangle: Normalize(Cross(v1-v0, v2-v0)), primitive_triangle.cc:59-68
  AABB cyl_bb = EmptyBox();
  uint32_t index = 0;                       // insertion index
};

// algebra.h:31-52
bool SolveQuadratic(float a, float b, float c, float& alpha, float& beta) {
  const float d = b * b - 4 * a * c;
  if (d < 0) return false;
  const float sqrt_d = std::sqrt(d);
  alpha = -b - sqrt_d;
  beta = -b + sqrt_d;
  if (std::abs(alpha) < std::abs(beta)) { alpha = c / beta * 2; beta /= 2 * a; }
  else { beta = c / alpha * 2; alpha /= 2 * a; }
  return true;
}

Hit IntersectTriangle(const Object& o, const Ray& ray) {               // primitive_triangle.cc:97-128
  const V3 E1 = o.b - o.a;
  const V3 E2 = o.c - o.a;
  const V3 P = Cross(ray.d, E2);
  const float det = Dot(P, E1);
  const V3 T = ray.o - o.a;
  const float u = Dot(P, T) / det;
  if (u > 1 || u < 0) return Hit();
  const V3 Q = Cross(T, E1);
  const float v = Dot(Q, ray.d) / det;
  if (v > 1 || v < 0) return Hit();
  if (u + v > 1) return Hit();
  const float t = Dot(Q, E2) / det;
  if (t < kEPS) return Hit();                                           // float vs long double 1e-6L
  return MakeHitU(o.a + u * E1 + v * E2, o.normal, t);
}
Hit IntersectSphere(const Object& o, const Ray& ray) {                 // primitive_sphere.cc:75-107
  const float a = 1.0f;
  const float b = -2 * Dot(o.a - ray.o, ray.d);
  const float c = SquaredLength(o.a - ray.o) - o.radius * o.radius;
  float alpha, beta;
  if (!SolveQuadratic(a, b, c, alpha, beta)) return Hit();
  if (alpha > kEPS) return MakeHitN(ray.o + alpha * ray.d, ray.o + alpha * ray.d - o.a, alpha);
  if (beta > kEPS) return MakeHitN(ray.o + beta * ray.d, ray.o + beta * ray.d - o.a, beta);
  return Hit();
}
Hit IntersectDisk(const Object& o, const Ray& ray) {                   // primitive_disk.cc:94-114
  const float cos_theta = Dot(ray.d, o.b);
  if (cos_theta == 0) return Hit();
  const float t = Dot(o.a - ray.o, o.b) / cos_theta;
  if (t < kEPS) return Hit();
  const float sq = SquaredLength(ray.o + t * ray.d - o.a);
  if (sq > o.radius * o.radius) return Hit();
  return MakeHitU(ray.o + t * ray.d, o.b, t);
}
Hit IntersectCylinder(const Object& o, const Ray& ray) {               // primitive_cylinder.cc:100-142
  const V3 OC = o.a - ray.o;
  const V3 u = ray.d - Dot(ray.d, o.b) * o.b;
  const V3 v = OC - Dot(OC, o.b) * o.b;
  const float a = SquaredLength(u);
  const float b = -2 * Dot(u, v);
  const float c = SquaredLength(v) - o.radius * o.radius;
  float alpha, beta;
  if (!SolveQuadratic(a, b, c, alpha, beta)) return Hit();
  if (alpha > kEPS) {
    const float h = Dot(alpha * ray.d - OC, o.b);
    if (h >= 0 && h <= o.height)
      return MakeHitN(ray.o + alpha * ray.d, ray.o + alpha * ray.d - o.a - h * o.b, alpha);
  }
  if (beta > kEPS) {
    const float h = Dot(beta * ray.d - OC, o.b);
    if (h >= 0 && h <= o.height)
      return MakeHitN(ray.o + beta * ray.d, ray.o + beta * ray.d - o.a - h * o.b, beta);
  }
  return Hit();
}
Hit Intersect(const Object& o, const Ray& ray) {                       // scene/object.h:136-141
  switch (o.kind) {
    case ORACLE_PRIM_TRIANGLE: return IntersectTriangle(o, ray);
    case ORACLE_PRIM_SPHERE: return IntersectSphere(o, ray);
    case ORACLE_PRIM_DISK: return IntersectDisk(o, ray);
    default: return IntersectCylinder(o, ray);
  }
}
AABB DiskBox(V3 center, V3 normal, float radius) {                     // primitive_disk.cc:81-92
  const V3 f = v3(std::sqrt(1 - normal.x * normal.x), std::sqrt(1 - normal.y * normal.y),
                  std::sqrt(1 - normal.z * normal.z));
  return AABB{center - radius * f, center + radius * f};
}
AABB BoundingBox(const Object& o) {
  switch (o.kind) {
    case ORACLE_PRIM_TRIANGLE:                                          // primitive_triangle.cc:80-95
      return AABB{v3(std::min({o.a.x, o.b.x, o.c.x}), std::min({o.a.y, o.b.y, o.c.y}), std::min({o.a.z, o.b.z, o.c.z})),
                  v3(std::max({o.a.x, o.b.x, o.c.x}), std::max({o.a.y, o.b.y, o.c.y}), std::max({o.a.z, o.b.z, o.c.z}))};
    case ORACLE_PRIM_SPHERE:                                            // primitive_sphere.cc:69-73
      return AABB{o.a - splat(o.radius), o.a + splat(o.radius)};
    case ORACLE_PRIM_DISK: return DiskBox(o.a, o.b, o.radius);
    default: return o.cyl_bb;                                           // primitive_cylinder.cc:73-84
  }
}
V3 Center(const Object& o) {
  switch (o.kind) {
    case ORACLE_PRIM_TRIANGLE:                                          // primitive_triangle.cc:70-78
      return v3((o.a.x + o.b.x + o.c.x) / 3, (o.a.y + o.b.y + o.c.y) / 3, (o.a.z + o.b.z + o.c.z) / 3);
    case ORACLE_PRIM_CYLINDER: return o.a + o.height / 2 * o.b;         // primitive_cylinder.cc:86-90
    default: return o.a;
  }
}
float TriangleArea(const Object& o) { return Length(Cross(o.b - o.a, o.c - o.a)) / 2; }   // primitive_triangle.cc:130-134
Object MakeTriangle(V3 v0, V3 v1, V3 v2, uint32_t material) {
  Object o; o.kind = ORACLE_PRIM_TRIANGLE; o.material = material; o.a = v0; o.b = v1; o.c = v2;
  o.normal = Normalize(Cross(v1 - v0, v2 - v0));
  return o;
}
void FinishObject(Object& o) {
  if (o.kind == ORACLE_PRIM_TRIANGLE) o.normal = Normalize(Cross(o.b - o.a, o.c - o.a));
  if (o.kind == ORACLE_PRIM_CYLINDER) {
    o.cyl_bb = Union(Union(EmptyBox(), DiskBox(o.a, o.b, o.radius)), DiskBox(o.a + o.height * o.b, o.b, o.radius));
  }
}

// ------------------------------------------------------------------------------------------
// BVH (include/amber/raytracer/acceleration_bvh.h:134-403) and List (acceleration_list.h:51-68)
// ------------------------------------------------------------------------------------------
// Conservative culling (ORACLE_ACCEL_BVH_CONS): a ray prepared for the widened slab test (ConsRay, below the BVH).
struct ConsRay {
  double o[3], d[3];
  bool flat[3];          // d == 0 on the axis: the ray stays at o there
  double E;              // extra half width of every box for THIS ray (direction-length drift of the sphere test)
  double slack_t;        // absolute slack of every comparison between a box parameter and a reference hit distance
};

struct BVH {
  using It = std::vector<Object>::iterator;
  struct Node {
    std::unique_ptr<Node> left, right;
    It first, last;
    AABB bb;             // the reference's box: union of Primitive::BoundingBox
    AABB cb;             // ORACLE_ACCEL_BVH_CONS only: union of ConservativeBox over the subtree
  };
  struct Split { float cost = FLT_MAX; It middle; AABB bl = EmptyBox(), br = EmptyBox(); };

  std::vector<Object> objects;
  std::unique_ptr<Node> root;
  std::atomic<uint32_t> n_nodes{0}, n_leaves{0}, max_depth{0};

  static AABB Box(It first, It last) {                                  // :182-194
    AABB bb = EmptyBox();
    for (It i = first; i != last; ++i) bb = Union(bb, BoundingBox(*i));
    return bb;
  }
  static float SAH(const AABB& p, const AABB& l, const AABB& r, std::size_t nl, std::size_t nr) {  // :298-312
    return 2 * 2.0f + (nl * SurfaceArea(l) + nr * SurfaceArea(r)) / SurfaceArea(p) * 1.0f;
  }
  // The reference passes the axis as a std::function<real_type(const Vector3&)> (:243-246); a template parameter calls the same
  // comparisons in the same order (std::sort / std::lower_bound are deterministic functions of the comparator's answers), 3x faster.
  template <int kAxis> static float Axis(V3 v) { return kAxis == 0 ? v.x : (kAxis == 1 ? v.y : v.z); }
  template <int kAxis>
  static Split FindSplitAxis(It first, It last, const AABB& bb) {       // :243-296
    std::sort(first, last, [](const Object& a, const Object& b) { return Axis<kAxis>(Center(a)) < Axis<kAxis>(Center(b)); });
    const std::size_t n_splits =
        std::min<std::size_t>(15.0f, std::log2(std::distance(first, last)));
    Split split;
    for (std::size_t i = 0; i < n_splits; i++) {
      const float split_point = Axis<kAxis>(bb.mn) + (Axis<kAxis>(bb.mx) - Axis<kAxis>(bb.mn)) * (i + 1) / (n_splits + 1);
      const It middle = std::lower_bound(first, last, split_point,
          [](const Object& object, const float sp) { return Axis<kAxis>(Center(object)) < sp; });
      const AABB bl = Box(first, middle), br = Box(middle, last);
      const std::size_t nl = std::distance(first, middle), nr = std::distance(middle, last);
      const float cost = SAH(bb, bl, br, nl, nr);
      if (cost < split.cost) { split.cost = cost; split.middle = middle; split.bl = bl; split.br = br; }
    }
    return split;
  }
  static Split FindSplit(It first, It last, const AABB& bb) {           // :197-241
    const Split sx = FindSplitAxis<0>(first, last, bb);
    const Split sy = FindSplitAxis<1>(first, last, bb);
    const Split sz = FindSplitAxis<2>(first, last, bb);
    if (sx.cost < sy.cost && sx.cost < sz.cost) {
      std::sort(first, last, [](const Object& a, const Object& b) { return Center(a).x < Center(b).x; });
      return sx;
    } else if (sy.cost < sx.cost) {
      std::sort(first, last, [](const Object& a, const Object& b) { return Center(a).y < Center(b).y; });
      return sy;
    } else {
      std::sort(first, last, [](const Object& a, const Object& b) { return Center(a).z < Center(b).z; });
      return sz;
    }
  }
  // Subtrees work on disjoint ranges of `objects`, so the two recursive calls of a large node may run on two threads: the tree
  // (topology, boxes, object order) is the one the single-threaded reference recursion builds.  1M spheres: 19 s -> 4 s on 8 cores.
  std::unique_ptr<Node> Create(It first, It last, const AABB& bb, uint32_t depth) {   // :158-180
    n_nodes++;
    for (uint32_t seen = max_depth.load(); depth > seen && !max_depth.compare_exchange_weak(seen, depth);) {}
    const Split split = FindSplit(first, last, bb);
    auto node = std::make_unique<Node>();
    node->bb = bb;
    node->cb = EmptyBox();
    if (split.cost > std::distance(first, last) * 1.0f) {
      n_leaves++;
      node->first = first; node->last = last;
    } else {
      node->first = node->last = It();
      if (depth < 4 && std::distance(first, last) > 20000) {
        auto left = std::async(std::launch::async, [&] { return Create(first, split.middle, split.bl, depth + 1); });
        node->right = Create(split.middle, last, split.br, depth + 1);
        node->left = left.get();
      } else {
        node->left = Create(first, split.middle, split.bl, depth + 1);
        node->right = Create(split.middle, last, split.br, depth + 1);
      }
    }
    return node;
  }
  explicit BVH(std::vector<Object>&& objs) : objects(std::move(objs)) {  // :134-138
    root = Create(objects.begin(), objects.end(), Box(objects.begin(), objects.end()), 0);
  }
  static void CastNode(const Node* n, const Ray& ray, float distance, Hit& out_hit, const Object*& out_obj) {  // :340-403
    if (n->first != n->last) {
      Hit closest; const Object* closest_obj = nullptr;
      for (It i = n->first; i != n->last; ++i) {
        const Hit hit = Intersect(*i, ray);
        if (hit && hit.t < distance) { distance = hit.t; closest = hit; closest_obj = &*i; }
      }
      out_hit = closest; out_obj = closest_obj; return;
    }
    bool lh = false, rh = false; float lin = FLT_MAX, rin = FLT_MAX, dummy;
    if (n->left) lh = SlabTest(n->left->bb, ray, distance, lin, dummy);
    if (n->right) rh = SlabTest(n->right->bb, ray, distance, rin, dummy);
    if (!lh && !rh) { out_hit = Hit(); out_obj = nullptr; return; }
    if (lh && !rh) { CastNode(n->left.get(), ray, distance, out_hit, out_obj); return; }
    if (!lh && rh) { CastNode(n->right.get(), ray, distance, out_hit, out_obj); return; }
    const Node* near_ = lin < rin ? n->left.get() : n->right.get();
    const Node* far_ = lin < rin ? n->right.get() : n->left.get();
    Hit nh; const Object* no = nullptr;
    CastNode(near_, ray, distance, nh, no);
    if (!nh) { CastNode(far_, ray, distance, out_hit, out_obj); return; }
    if (nh.t < std::max(lin, rin)) { out_hit = nh; out_obj = no; return; }
    Hit fh; const Object* fo = nullptr;
    CastNode(far_, ray, nh.t, fh, fo);
    if (!fh) { out_hit = nh; out_obj = no; } else { out_hit = fh; out_obj = fo; }
  }

  // ---- ORACLE_ACCEL_BVH_CONS: the same tree, culled so that the answer is the List scan's --------------------------------
  // The reference's BVH culls with the GEOMETRIC boxes of its primitives (Primitive::BoundingBox) while its primitive tests are
  // binary32 arithmetic that accepts rays slightly OUTSIDE the geometry (below), returns the first-visited object on a distance
  // tie, and stops at the near child when its hit lies in front of the far child's box (:386-391).  So Cast through the
  // reference's BVH and through its List (acceleration_list.h:51-68) are two different functions of the ray; on the 1M-sphere
  // scene they differ on 2e-7 of the rays.  The engine under test implements List.  List over 1M objects costs a millisecond per
  // ray, so this mode computes the List answer at BVH speed: every object whose reference test can possibly accept the ray is
  // tested with the reference's arithmetic (Intersect), and the winner is chosen by List's rule -- the smallest distance, the
  // lower insertion index on a tie.  It is a restatement of nothing in the reference: it is the checker's way to evaluate
  // acceleration_list.h:51-68 on large scenes, and tests/test_oracle_conservative_bvh.py proves it equal to the plain scan.
  //
  // "Can possibly accept": ConservativeBox(o) contains every point o + t d (exact arithmetic) at which the reference's test of
  // object o returns a hit for a ray of unit direction; a ray with |d|^2 = 1 + delta may miss it by ConsRay.E.  The margins are
  // those DESIGN.md section 5 derives for engine BVH, each taken FOUR TIMES as large (so that a GPU == oracle comparison also
  // tests whether the engine's own margins suffice):
  //  * sphere (primitive_sphere.cc:75-107, algebra.h:31-52): the discriminant b*b - 4*c is binary32 arithmetic on b ~ 2D,
  //    c ~ D^2 (D = distance origin -> centre), absolute error ~ 12 eps D^2: accepted impact parameters p^2 <= r^2 + ~12 eps D^2,
  //    and the float roots move by up to sqrt of that along the ray.  Box radius r' = sqrt(r^2 + 64 eps Dmax^2), Dmax = scene diagonal.
  //  * triangle (primitive_triangle.cc:97-128): u and v carry an absolute error ~ 6 eps |T||E|/|det| each; along a needle
  //    (e_max / e_min >> 1) the test accepts points ~ 36 eps |T| e_max / e_min beyond the short edge: boxes widened by 4x that.
  //  * disk (primitive_disk.cc:94-114): geometric for any normal length; cylinder (primitive_cylinder.cc:100-142) with an axis
  //    of length nu: radial distance <= r and axial distance <= height / nu from the base centre: sphere of radius hypot(r, height / nu).
  //  * every box padded by 2^-14 of (scene extent + coordinate magnitude): roundings of the accepted point itself.
  struct ConsScene {
    V3 center{0, 0, 0}; double half_diag = 0, inv_rmin = 0, extent = 0;
  } cons;
  bool has_cons = false;

  static AABB ConservativeBox(const Object& o, double sphere_slack2, double tri_reach, double extent) {
    double mn[3], mx[3];
    auto grow = [&](V3 p) { const double q[3] = {p.x, p.y, p.z}; for (int c = 0; c < 3; c++) { mn[c] = std::min(mn[c], q[c]); mx[c] = std::max(mx[c], q[c]); } };
    for (int c = 0; c < 3; c++) { mn[c] = 1e300; mx[c] = -1e300; }
    auto ball = [&](V3 ctr, double r) { const double q[3] = {ctr.x, ctr.y, ctr.z}; for (int c = 0; c < 3; c++) { mn[c] = q[c] - r; mx[c] = q[c] + r; } };
    switch (o.kind) {
      case ORACLE_PRIM_TRIANGLE: {
        grow(o.a); grow(o.b); grow(o.c);
        const V3 e1 = o.b - o.a, e2 = o.c - o.a, e3 = o.c - o.b;
        const double l1 = double(e1.x) * e1.x + double(e1.y) * e1.y + double(e1.z) * e1.z, l2 = double(e2.x) * e2.x + double(e2.y) * e2.y + double(e2.z) * e2.z,
                     l3 = double(e3.x) * e3.x + double(e3.y) * e3.y + double(e3.z) * e3.z;
        const double emax = std::sqrt(std::max({l1, l2, l3})), emin = std::sqrt(std::min({l1, l2, l3}));
        double m = tri_reach;                                             // degenerate triangle: the whole reach
        if (emin > 0.0 && std::isfinite(emax)) m = std::min(tri_reach, 4.0 * 36.0 * 5.9604644775390625e-08 * tri_reach * emax / emin);
        for (int c = 0; c < 3; c++) { mn[c] -= m; mx[c] += m; }
        break;
      }
      case ORACLE_PRIM_SPHERE: ball(o.a, std::sqrt(double(o.radius) * o.radius + sphere_slack2) * 1.00001); break;
      case ORACLE_PRIM_DISK: ball(o.a, std::fabs(double(o.radius)) * 1.00001); break;
      default: {
        const double nu = std::sqrt(double(o.b.x) * o.b.x + double(o.b.y) * o.b.y + double(o.b.z) * o.b.z);
        const double axial = nu > 1e-30 ? std::fabs(double(o.height)) / nu : 1e30;
        ball(o.a, std::min(1e30, std::hypot(double(o.radius), axial) * 1.00001));
      }
    }
    AABB bb;
    float* lo[3] = {&bb.mn.x, &bb.mn.y, &bb.mn.z}; float* hi[3] = {&bb.mx.x, &bb.mx.y, &bb.mx.z};
    for (int c = 0; c < 3; c++) {
      const double pad = 6.103515625e-05 * (extent + std::max(std::fabs(mn[c]), std::fabs(mx[c]))) + 1e-30;   // 2^-14
      *lo[c] = std::nextafter(static_cast<float>(mn[c] - pad), -FLT_MAX);   // the conversion may round inward: one more step out
      *hi[c] = std::nextafter(static_cast<float>(mx[c] + pad), FLT_MAX);
      if (!(mn[c] == mn[c]) || !(mx[c] == mx[c])) { *lo[c] = -FLT_MAX; *hi[c] = FLT_MAX; }                    // NaN geometry: never culled
    }
    return bb;
  }
  AABB FillCons(Node* n, double sphere_slack2, double tri_reach, double extent) {
    if (n->first != n->last) {
      n->cb = EmptyBox();
      for (It i = n->first; i != n->last; ++i) n->cb = Union(n->cb, ConservativeBox(*i, sphere_slack2, tri_reach, extent));
    } else {
      n->cb = EmptyBox();
      if (n->left) n->cb = Union(n->cb, FillCons(n->left.get(), sphere_slack2, tri_reach, extent));
      if (n->right) n->cb = Union(n->cb, FillCons(n->right.get(), sphere_slack2, tri_reach, extent));
    }
    return n->cb;
  }
  void BuildConservative() {
    const AABB all = root->bb;                                           // geometric bounds of the scene
    const double dx = double(all.mx.x) - all.mn.x, dy = double(all.mx.y) - all.mn.y, dz = double(all.mx.z) - all.mn.z;
    const double diag = std::sqrt(dx * dx + dy * dy + dz * dz);
    const double sphere_slack2 = 64.0 * 5.9604644775390625e-08 * diag * diag;
    cons.extent = std::max({dx, dy, dz});
    const AABB cb = FillCons(root.get(), sphere_slack2, diag, cons.extent);
    cons.center = v3(0.5f * (cb.mn.x + cb.mx.x), 0.5f * (cb.mn.y + cb.mx.y), 0.5f * (cb.mn.z + cb.mx.z));
    const double ex = double(cb.mx.x) - cb.mn.x, ey = double(cb.mx.y) - cb.mn.y, ez = double(cb.mx.z) - cb.mn.z;
    cons.half_diag = 0.5 * std::sqrt(ex * ex + ey * ey + ez * ez) * 1.001;
    double rmin = 1e300; bool spheres = false;
    for (const Object& o : objects) if (o.kind == ORACLE_PRIM_SPHERE) { spheres = true; rmin = std::min(rmin, std::fabs(double(o.radius))); }
    cons.inv_rmin = spheres ? (rmin > 0 ? 1.0 / rmin : 1e300) : 0.0;
    has_cons = true;
  }
  // Direction length: the reference never renormalises sampled directions (vector3.h:236-239) and its sphere test assumes
  // |d| = 1 (a = 1, primitive_sphere.cc:80-83).  With |d|^2 = 1 + delta, s the distance of the closest approach along the ray
  // and p the impact parameter, its discriminant is 4 (r^2 - p^2 + delta s^2): it accepts p^2 <= r^2 + delta s^2, i.e. misses the
  // sphere's box by up to sqrt(r^2 + delta s^2) - r <= min(delta S^2 / r_min, sqrt(delta) S) with S >= s (S = |o - centre| + half
  // the scene diagonal); and its distance |d| s -+ sqrt(..) differs from the parameter of the geometric crossing, (s -+ h) / |d|,
  // by about |delta| s / |d|.  Both four times the engine's: E = 4 min(..), slack_t = 8 |delta| S / min(|d|, 1).
  ConsRay Prepare(const Ray& ray) const {
    ConsRay r;
    const double o[3] = {ray.o.x, ray.o.y, ray.o.z}, d[3] = {ray.d.x, ray.d.y, ray.d.z};
    for (int c = 0; c < 3; c++) { r.o[c] = o[c]; r.d[c] = d[c]; r.flat[c] = d[c] == 0.0; }
    const double len2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], delta = len2 - 1.0;
    const double cx = o[0] - cons.center.x, cy = o[1] - cons.center.y, cz = o[2] - cons.center.z;
    const double S = 1.001 * (std::sqrt(cx * cx + cy * cy + cz * cz) + cons.half_diag);
    const double dpos = delta > 0 ? delta : 0.0;
    r.E = cons.inv_rmin > 0 ? 4.0 * std::min(dpos * S * S * cons.inv_rmin, std::sqrt(dpos) * S) : 0.0;
    const double len = std::sqrt(std::min(len2, 1.0));
    r.slack_t = 8.0 * std::fabs(delta) * S / (len > 0 ? len : 1e-300) + 1e-9 * S;
    if (!(r.E == r.E)) r.E = 1e300;                                      // NaN ray: nothing is culled (every exact test then fails on its own)
    if (!(r.slack_t == r.slack_t)) r.slack_t = 1e300;
    return r;
  }
  // The ray's parameter interval inside cb widened by E, in binary64 (the box planes and the ray are binary32 values, so lo - o is
  // exact or rounds once; the quotient rounds once: each plane parameter is within 2^-51 of its exact value, covered by the
  // relative 1e-9).  Written so that a NaN anywhere means "visit".
  static bool SlabCons(const AABB& cb, const ConsRay& r, double t_best, double& t_in) {
    const double mn[3] = {cb.mn.x, cb.mn.y, cb.mn.z}, mx[3] = {cb.mx.x, cb.mx.y, cb.mx.z};
    double tn = -1e300, tf = 1e300;
    for (int c = 0; c < 3; c++) {
      const double lo = mn[c] - r.E, hi = mx[c] + r.E;
      if (r.flat[c]) { if (r.o[c] < lo || r.o[c] > hi) return false; continue; }
      double t0 = (lo - r.o[c]) / r.d[c], t1 = (hi - r.o[c]) / r.d[c];
      if (t1 < t0) std::swap(t0, t1);
      t0 -= 1e-9 * std::fabs(t0); t1 += 1e-9 * std::fabs(t1);
      if (t0 > tn) tn = t0;
      if (t1 < tf) tf = t1;
    }
    t_in = tn;
    if (tn > tf) return false;                                           // the ray misses the widened box
    if (tf + r.slack_t < 0.0) return false;                              // the box lies behind the origin (accepted hits have t > kEPS)
    if (tn - r.slack_t > t_best * (1.0 + 1e-6)) return false;            // the box begins behind the closest hit so far (ties are visited)
    return true;
  }
  static void CastCons(const Node* n, const Ray& ray, const ConsRay& cr, float& distance, Hit& best, const Object*& best_obj) {
    if (n->first != n->last) {
      for (It i = n->first; i != n->last; ++i) {
        const Hit hit = Intersect(*i, ray);
        // acceleration_list.h:58-64 scans in insertion order with a strict <: the smallest distance, then the lowest index
        if (hit && (hit.t < distance || (best_obj && hit.t == distance && i->index < best_obj->index))) { distance = hit.t; best = hit; best_obj = &*i; }
      }
      return;
    }
    bool lh = false, rh = false; double lin = 0, rin = 0;
    if (n->left) lh = SlabCons(n->left->cb, cr, distance, lin);
    if (n->right) rh = SlabCons(n->right->cb, cr, distance, rin);
    const Node* first = n->left.get(); const Node* second = n->right.get();
    if (lh && rh && rin < lin) std::swap(first, second);
    if (!lh) { first = rh ? n->right.get() : nullptr; second = nullptr; }
    else if (!rh) second = nullptr;
    if (first) CastCons(first, ray, cr, distance, best, best_obj);
    if (second) {
      double tin;
      if (SlabCons(second->cb, cr, distance, tin)) CastCons(second, ray, cr, distance, best, best_obj);   // re-tested against the closer hit found meanwhile
    }
  }
};

// ------------------------------------------------------------------------------------------
// Materials (src/amber/scene/material_*.cc, include/amber/scene/material_*.h)
// ------------------------------------------------------------------------------------------
struct Material {
  uint32_t kind = 0;
  V3 rho{0, 0, 0};
  float param = 0;   // exponent / ior
  float r0 = 0;      // refraction: Fresnel(ior), material_refraction.cc:265-269
};
float Fresnel(float ior, const Math& M) { return static_cast<float>(M.pow_i((ior - 1) / (ior + 1), 2)); }
float Schlick(float r0, float cos_theta, const Math& M) {              // material_refraction.cc:271-275 (double pow)
  return static_cast<float>(r0 + (1 - r0) * M.pow_i(1 - cos_theta, 5));
}
struct Scatter { V3 dir{0, 0, 0}; V3 weight{0, 0, 0}; };               // rendering/scatter.h:32-48

V3 Radiance(const Material& m, V3 normal, V3 dir_out) {                // material_diffuse_light.h:127-139, material.h:101-109
  if (m.kind != ORACLE_MAT_DIFFUSE_LIGHT) return splat(0);
  if (Dot(dir_out, normal) <= 0) return splat(0);
  return m.rho;
}
Scatter SampleLight(const Material& m, V3 normal, V3 dir_out, Sampler& smp, const Math& M) {
  Scatter sc;
  switch (m.kind) {
    case ORACLE_MAT_LAMBERTIAN: {                                       // material_lambertian.cc:61-70
      const V3 w = Dot(dir_out, normal) > 0 ? normal : -normal;
      sc.dir = HemispherePSA(w, smp, M);
      sc.weight = 1.0f * m.rho;                                         // material_basic.h:327-338
      return sc;
    }
    case ORACLE_MAT_PHONG: {                                            // material_phong.cc:81-106
      const V3 refl = PerfectReflection(dir_out, normal, Dot(dir_out, normal));
      const float signed_cos_o = Dot(dir_out, normal);
      // The reference re-samples until the direction is on dir_out's side -- forever when that cannot happen (a
      // normal that is not of unit length can put the whole lobe on the other side).  The device engine must
      // terminate, so both sides accept attempt number ORACLE_PHONG_MAX_TRIES as it is (probability 2^-1024 otherwise).
      for (int attempt = 1;; ++attempt) {
        const V3 dir_in = CosinePower(refl, m.param, smp, M);
        const float signed_cos_i = Dot(dir_in, normal);
        if (signed_cos_o * signed_cos_i <= 0 && attempt < ORACLE_PHONG_MAX_TRIES) continue;
        sc.dir = dir_in;
        sc.weight = ((m.param + 2) / (m.param + 1) * std::abs(signed_cos_i)) * m.rho;
        return sc;
      }
    }
    case ORACLE_MAT_SPECULAR: {                                         // material_specular.cc:62-70
      sc.dir = PerfectReflection(dir_out, normal, Dot(dir_out, normal));
      sc.weight = 1.0f * m.rho;
      return sc;
    }
    case ORACLE_MAT_REFRACTION: {                                       // material_refraction.cc:177-220
      const float signed_cos_alpha = Dot(dir_out, normal);
      const float ior = signed_cos_alpha > 0 ? 1 / m.param : m.param;
      const float squared_cos_beta = 1 - (1 - signed_cos_alpha * signed_cos_alpha) * (ior * ior);
      const V3 dir_r = PerfectReflection(dir_out, normal, signed_cos_alpha);
      if (squared_cos_beta < 0) { sc.dir = dir_r; sc.weight = 1.0f * m.rho; return sc; }
      const float cos_alpha = std::abs(signed_cos_alpha);
      const float cos_beta = std::sqrt(squared_cos_beta);
      const V3 dir_t = -ior * dir_out + ((signed_cos_alpha < 0 ? 1 : -1) * cos_beta + ior * signed_cos_alpha) * normal;
      const float rho_r = Schlick(m.r0, cos_alpha, M);
      const float rho_t = (1 - rho_r) * (ior * ior);
      const float rho = rho_r + rho_t;
      const float p_r = (rho_r / rho + 0.5f) / 2;
      const float p_t = (rho_t / rho + 0.5f) / 2;
      if (UniformF(smp) < p_r) { sc.dir = dir_r; sc.weight = (rho_r / p_r) * m.rho; }
      else { sc.dir = dir_t; sc.weight = (rho_t / p_t) * m.rho; }
      return sc;
    }
    case ORACLE_MAT_EYE:                                                // material_eye.h:146-155
      sc.dir = -dir_out; sc.weight = splat(1); return sc;
    default:                                                            // DiffuseLight, material_diffuse_light.h:185-194
      return sc;
  }
}

// ------------------------------------------------------------------------------------------
// Thin lens + sensor (src/amber/scene/lens_thin.cc:32-148, src/amber/rendering/sensor.cc:94-120)
// ------------------------------------------------------------------------------------------
struct M3 {
  float e[9];
  V3 operator()(V3 v) const {                                           // matrix3.h:101-109
    return v3(e[0] * v.x + e[1] * v.y + e[2] * v.z, e[3] * v.x + e[4] * v.y + e[5] * v.z,
              e[6] * v.x + e[7] * v.y + e[8] * v.z);
  }
  M3 Inverse() const {                                                  // matrix3.h:111-143
    const float e11 = e[0], e12 = e[1], e13 = e[2], e21 = e[3], e22 = e[4], e23 = e[5], e31 = e[6], e32 = e[7], e33 = e[8];
    const float d = +e11 * (e22 * e33 - e23 * e32) - e21 * (e12 * e33 - e13 * e32) + e31 * (e12 * e23 - e13 * e22);
    M3 r;
    if (d == 0) { for (float& x : r.e) x = std::numeric_limits<float>::quiet_NaN(); return r; }
    const float dinv = 1 / d;
    r.e[0] = +dinv * (e22 * e33 - e23 * e32); r.e[1] = -dinv * (e12 * e33 - e13 * e32); r.e[2] = +dinv * (e12 * e23 - e13 * e22);
    r.e[3] = -dinv * (e21 * e33 - e23 * e31); r.e[4] = +dinv * (e11 * e33 - e13 * e31); r.e[5] = -dinv * (e11 * e23 - e13 * e21);
    r.e[6] = +dinv * (e21 * e32 - e22 * e31); r.e[7] = -dinv * (e11 * e32 - e12 * e31); r.e[8] = +dinv * (e11 * e22 - e12 * e21);
    return r;
  }
};
struct ThinLens {
  int kind = 0;                     // 0 thin (lens_thin.cc), 1 pinhole (lens_pinhole.cc)
  V3 origin; M3 global_, local_;
  std::vector<Object> blades;       // lens-owned aperture triangles (lens_thin.cc:46-55)
  float focus_distance, sensor_distance, p_area;
};
ThinLens MakeThinLens(const float t[16], float focal_length, float focus_distance, float radius,
                      uint32_t n_blades, uint32_t eye_material, const Math& M) {   // lens_thin.cc:32-57
  ThinLens L;
  // origin_ = transform(Vector3()) : matrix4.h:98-107
  L.origin = v3(t[0] * 0.0f + t[1] * 0.0f + t[2] * 0.0f + t[3], t[4] * 0.0f + t[5] * 0.0f + t[6] * 0.0f + t[7],
                t[8] * 0.0f + t[9] * 0.0f + t[10] * 0.0f + t[11]);
  const float g[9] = {t[0], t[1], t[2], t[4], t[5], t[6], t[8], t[9], t[10]};
  std::memcpy(L.global_.e, g, sizeof g);
  L.local_ = L.global_.Inverse();
  L.focus_distance = focus_distance;
  L.sensor_distance = 1 / (1 / focal_length - 1 / focus_distance);
  L.p_area = 1;
  const std::size_t n = n_blades;
  for (std::size_t i = 0; i < n; i++) {
    const float alpha = 2 * kPIf / n * i;
    const float beta = 2 * kPIf / n * (i + 1);
    float sa, ca, sb, cb;
    if (M.mode == ORACLE_MATH_LIBM) { ca = std::cos(alpha); sa = std::sin(alpha); cb = std::cos(beta); sb = std::sin(beta); }
    else { PortableSinCos(alpha, &sa, &ca); PortableSinCos(beta, &sb, &cb); }
    L.blades.push_back(MakeTriangle(L.origin + L.global_(radius * v3(ca, sa, 0)),
                                    L.origin + L.global_(radius * v3(cb, sb, 0)), L.origin, eye_material));
  }
  L.p_area /= TriangleArea(L.blades.front()) * n;
  return L;
}

struct Sensor { uint64_t w, h; float sw, sh; };

struct EyeRay { V3 origin, normal, dir; float weight; uint32_t blade; };
// BasicThin::GenerateRay lens_thin.cc:70-107 ; Sensor::PixelBound::Uniform sensor.cc:111-120.
// The two jitter draws are constructor arguments of Vector2 (sensor.cc:114-117): their order is
// unspecified by C++; the reference's recorded outputs come from g++ 11.4 which evaluates the
// SECOND argument (Y) first.  The oracle fixes that order: draw#4 -> Y, draw#5 -> X.
EyeRay GenerateEyeRay(const ThinLens& L, const Sensor& S, uint64_t px, uint64_t py, Sampler& smp, const Math& M) {
  if (L.kind == 1) {                                                     // BasicPinhole::GenerateRay lens_pinhole.cc:48-68
    const float jy = UniformF(smp);
    const float jx = UniformF(smp);
    const float uvx = (px + jx) / S.w;
    const float uvy = (py + jy) / S.h;
    const V3 sensor_point = v3((uvx - 0.5f) * S.sw, (uvy - 0.5f) * S.sh, L.sensor_distance);
    const V3 ray_dir = Normalize(L.global_(-sensor_point));              // Ray(origin, Vector3) normalises
    // PDFDirection lens_pinhole.cc:93-106
    const V3 direction = L.local_(ray_dir);
    const V3 point = L.sensor_distance / direction.z * direction;
    const float p_area = 1 / (S.sw * S.sh);
    const float geometry_factor = direction.z * direction.z / SquaredLength(point);
    const float pdf_dir = p_area / geometry_factor;
    EyeRay e;
    e.origin = L.origin; e.normal = Normalize(L.global_(v3(0, 0, -1))); e.dir = ray_dir; e.blade = 0;
    e.weight = 1 / static_cast<float>(1.0L) / pdf_dir;                   // 1 / PDFArea(kDiracDelta -> real_type) / PDFDirection
    return e;
  }
  const std::size_t nb = L.blades.size();
  const std::size_t pos = std::min<std::size_t>(nb - 1, std::floor(UniformF(static_cast<float>(nb), smp)));
  const Object& tri = L.blades[pos];
  // Triangle::SampleSurfacePoint primitive_triangle.cc:136-150
  float u = UniformF(smp);
  float v = UniformF(smp);
  if (u + v >= 1) { u = 1 - u; v = 1 - v; }
  const V3 ap_origin = (1 - u - v) * tri.a + u * tri.b + v * tri.c;
  const V3 aperture_point = L.local_(ap_origin - L.origin);
  const float jy = UniformF(smp);
  const float jx = UniformF(smp);
  const float uvx = (px + jx) / S.w;
  const float uvy = (py + jy) / S.h;
  const float sx = (uvx - 0.5f) * S.sw;                                 // UVToPoint sensor.cc:94-101
  const float sy = (uvy - 0.5f) * S.sh;
  const V3 sensor_point = v3(sx, sy, L.sensor_distance);
  const V3 direction = Normalize(-L.focus_distance / L.sensor_distance * sensor_point - aperture_point);
  const double factor = M.pow_i(Normalize(sensor_point - aperture_point).z / direction.z, 4);
  const V3 ray_dir = Normalize(L.global_(direction));                   // Ray(origin, Vector3) normalises, ray.h:52-56
  // PDFDirection lens_thin.cc:138-148 : float * double / double -> double -> real_type
  const V3 dloc = L.local_(ray_dir);
  const float pdf_dir = static_cast<float>(
      static_cast<float>(S.w * S.h) / (S.sw * S.sh) * M.pow_i(L.sensor_distance, 2) / M.pow_i(dloc.z, 4));
  EyeRay e;
  e.origin = ap_origin; e.normal = tri.normal; e.dir = ray_dir; e.blade = static_cast<uint32_t>(pos);
  e.weight = static_cast<float>(factor / L.p_area / pdf_dir);          // double / float / float -> real_type
  return e;
}


// ------------------------------------------------------------------------------------------
// Light tracing support (src/amber/rendering/algorithm_lt.cc:125-163, scene/light_set.h:61-123,
// lens_thin.cc:109-130, lens_pinhole.cc:70-85, sensor.cc:46-59, Primitive::SampleSurfacePoint)
// ------------------------------------------------------------------------------------------
Scatter SampleImportance(const Material& m, V3 normal, V3 dir_out, Sampler& smp, const Math& M) {
  if (m.kind != ORACLE_MAT_REFRACTION) return SampleLight(m, normal, dir_out, smp, M);   // symmetric forwarders / Eye / Light
  Scatter sc;                                                                             // material_refraction.cc:222-263
  const float signed_cos_alpha = Dot(dir_out, normal);
  const float ior = signed_cos_alpha > 0 ? 1 / m.param : m.param;
  const float squared_cos_beta = 1 - (1 - signed_cos_alpha * signed_cos_alpha) * (ior * ior);
  const V3 dir_r = PerfectReflection(dir_out, normal, signed_cos_alpha);
  if (squared_cos_beta < 0) { sc.dir = dir_r; sc.weight = 1.0f * m.rho; return sc; }
  const float cos_alpha = std::abs(signed_cos_alpha);
  const float cos_beta = std::sqrt(squared_cos_beta);
  const V3 dir_t = -ior * dir_out + ((signed_cos_alpha < 0 ? 1 : -1) * cos_beta + ior * signed_cos_alpha) * normal;
  const float rho_r = Schlick(m.r0, cos_alpha, M);
  const float rho_t = 1 - rho_r;                                       // no radiance scaling for importance transport
  const float p_r = (rho_r + 0.5f) / 2;
  const float p_t = (rho_t + 0.5f) / 2;
  if (UniformF(smp) < p_r) { sc.dir = dir_r; sc.weight = (rho_r / p_r) * m.rho; }
  else { sc.dir = dir_t; sc.weight = (rho_t / p_t) * m.rho; }
  return sc;
}

float SurfaceAreaOf(const Object& o) {
  switch (o.kind) {
    case ORACLE_PRIM_TRIANGLE: return TriangleArea(o);
    case ORACLE_PRIM_SPHERE: return 4 * kPIf * o.radius * o.radius;                 // primitive_sphere.cc:109-113
    case ORACLE_PRIM_DISK: return kPIf * o.radius * o.radius;                       // primitive_disk.cc:116-120
    default: return 2 * kPIf * o.radius * o.height;                                 // primitive_cylinder.cc:144-148
  }
}
// sampling.h:176-183: theta = Uniform<T>(2 * kPI, sampler) -- the product is long double, narrowed to T
inline void Circle(Sampler& smp, const Math& M, float& cx, float& cy) {
  const float theta = UniformF(static_cast<float>(2 * kPI), smp);
  float s, c; M.sincos(theta, s, c);
  cx = c; cy = s;
}
// returns origin and direction (the surface normal) of Primitive::SampleSurfacePoint
void SampleSurfacePoint(const Object& o, Sampler& smp, const Math& M, V3& origin, V3& normal) {
  switch (o.kind) {
    case ORACLE_PRIM_TRIANGLE: {                                        // primitive_triangle.cc:136-150
      float u = UniformF(smp), v = UniformF(smp);
      if (u + v >= 1) { u = 1 - u; v = 1 - v; }
      origin = (1 - u - v) * o.a + u * o.b + v * o.c; normal = o.normal; return;
    }
    case ORACLE_PRIM_SPHERE: {                                          // primitive_sphere.cc:115-122, SphereSA sampling.h:185-199
      const float r0 = static_cast<float>(smp()) * (1.0f - (-1.0f)) + (-1.0f);       // Uniform<T>(-1, 1, sampler)
      const float r1 = UniformF(smp);
      const float cos_theta = r0;
      const float sin_theta = std::sqrt(1 - r0 * r0);
      const float phi = 2 * kPIf * r1;
      float sp, cp; M.sincos(phi, sp, cp);
      normal = v3(cos_theta * cp, cos_theta * sp, sin_theta);
      origin = o.a + o.radius * normal; return;
    }
    case ORACLE_PRIM_DISK: {                                            // primitive_disk.cc:122-136
      const float radius = std::sqrt(UniformF(o.radius * o.radius, smp));
      V3 u, v; OrthonormalBasis(o.b, u, v);
      float ax, ay; Circle(smp, M, ax, ay);
      origin = o.a + (u * ax + v * ay) * radius; normal = o.b; return;
    }
    default: {                                                          // primitive_cylinder.cc:150-164
      const float height = UniformF(o.height, smp);
      V3 u, v; OrthonormalBasis(o.b, u, v);
      float ax, ay; Circle(smp, M, ax, ay);
      const V3 n = u * ax + v * ay;
      origin = o.a + o.b * height + n * o.radius;
      normal = Normalize(n);                                            // Ray(origin, Vector3) normalises, ray.h:52-56
      return;
    }
  }
}

struct Light { uint32_t object; float cum_power; V3 irradiance; };
struct LightSet {                                                       // scene/light_set.h:61-82
  std::vector<Light> lights;
  float total() const { return lights.empty() ? 0.0f : lights.back().cum_power; }
};

// Sensor::ResponsePixel sensor.cc:46-59 ; returns false when the point is off the sensor
bool ResponsePixel(const Sensor& S, float px, float py, uint64_t& x, uint64_t& y) {
  const float uvx = px / S.sw + 0.5f, uvy = py / S.sh + 0.5f;
  if (std::min(uvx, uvy) < 0 || std::max(uvx, uvy) >= 1) return false;
  x = std::min<uint64_t>(S.w - 1, uvx * S.w);                           // UVToPixel sensor.cc:86-92 (float -> integer truncation)
  y = std::min<uint64_t>(S.h - 1, uvy * S.h);
  return true;
}
// Lens::Response for Ray(position, direction_out): scene/scene.h:299-307
bool LensResponse(const ThinLens& L, const Sensor& S, V3 position, V3 direction_out, const Math& M, uint64_t& x, uint64_t& y, float& value) {
  const V3 direction = L.local_(direction_out);
  if (L.kind == 1) {                                                    // lens_pinhole.cc:70-85
    const V3 point = L.sensor_distance / direction.z * direction;
    if (!ResponsePixel(S, point.x, point.y, x, y)) return false;
    value = 1; return true;
  }
  if (direction.z >= 0) return false;                                   // lens_thin.cc:111-114
  const V3 aperture_point = L.local_(position - L.origin);
  const V3 sensor_point = -L.sensor_distance / L.focus_distance * aperture_point + L.sensor_distance / direction.z * direction;
  if (!ResponsePixel(S, sensor_point.x, sensor_point.y, x, y)) return false;
  value = static_cast<float>(M.pow_i(Normalize(sensor_point - aperture_point).z / direction.z, 4));
  return true;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// Scene
// ------------------------------------------------------------------------------------------
struct oracle_scene {
  int accel = ORACLE_ACCEL_BVH;
  std::vector<Object> objects;        // insertion order
  std::vector<Material> materials;
  ThinLens lens;
  LightSet light_set;
  std::unique_ptr<BVH> bvh;

  void Finish() {
    for (uint32_t i = 0; i < objects.size(); i++) objects[i].index = i;
    BuildLightSet();
    if (accel == ORACLE_ACCEL_BVH || accel == ORACLE_ACCEL_BVH_CONS) bvh = std::make_unique<BVH>(std::vector<Object>(objects));
    if (accel == ORACLE_ACCEL_BVH_CONS) bvh->BuildConservative();
  }
  // Scene::Create collects the SurfaceType::Light objects (scene/scene.h:177-182); LightSet sorts them by power and
  // accumulates (light_set.h:61-82).  Power = Sum(SurfaceArea * Irradiance) (scene/object.h:99-103), Irradiance = radiance * pi.
  void BuildLightSet() {
    struct Item { uint32_t index; float power; V3 irr; V3 irradiance() const { return irr; } };
    std::vector<Item> items;
    for (const Object& o : objects) {
      const Material& m = materials[o.material];
      if (m.kind != ORACLE_MAT_DIFFUSE_LIGHT) continue;
      const V3 irr = m.rho * kPIf;
      const V3 p = SurfaceAreaOf(o) * irr;
      items.push_back(Item{o.index, p.x + p.y + p.z, irr});
    }
    std::sort(items.begin(), items.end(), [](const Item& x, const Item& y) { return x.power < y.power; });
    float power = 0;
    for (const Item& it : items) { power += it.power; light_set.lights.push_back(Light{it.index, power, it.irradiance()}); }
  }
  // Scene::Cast scene/scene.h:236-244 -> Acceleration::Cast(ray, FLT_MAX) acceleration.h:46-51
  bool Cast(const Ray& ray, Hit& hit, const Object*& obj) const { return CastWith(accel, ray, hit, obj); }
  bool CastWith(int how, const Ray& ray, Hit& hit, const Object*& obj) const {
    if (how == ORACLE_ACCEL_BVH) {
      BVH::CastNode(bvh->root.get(), ray, FLT_MAX, hit, obj);
    } else if (how == ORACLE_ACCEL_BVH_CONS) {                          // acceleration_list.h:51-68 evaluated through the tree (BVH::CastCons)
      float distance = FLT_MAX; hit = Hit(); obj = nullptr;
      const ConsRay cr = bvh->Prepare(ray);
      BVH::CastCons(bvh->root.get(), ray, cr, distance, hit, obj);
    } else {                                                            // acceleration_list.h:51-68
      float distance = FLT_MAX; hit = Hit(); obj = nullptr;
      for (const Object& o : objects) {
        const Hit h = Intersect(o, ray);
        if (h && h.t < distance) { distance = h.t; hit = h; obj = &o; }
      }
    }
    return static_cast<bool>(hit);
  }
};

namespace {

// sig_obj / sig_t: FNV-1a (32 bit) over the object index of every cast (0xffffffff for a miss) / over the bits of every
// hit distance -- a path's discrete history and its exact arithmetic, for path-level parity counts (oracle_path_signatures)
struct PathResult { V3 measurement; uint32_t casts, hits; uint32_t sig_obj, sig_t; };
inline uint32_t Fnv32(uint32_t h, uint32_t v) { for (int k = 0; k < 4; k++) { h ^= (v >> (8 * k)) & 0xffu; h *= 16777619u; } return h; }

// Diagnostic (oracle_direction_length_stats, single-threaded): histogram of | |d|^2 - 1 | over the rays that are cast.
// The reference never renormalises sampled directions (vector3.h:236-239); engine BVH's sphere bounds depend on how
// far from unit length they drift.
struct DirStats { uint64_t rays = 0, above[6] = {0, 0, 0, 0, 0, 0}; double max_dev = 0; };
DirStats* g_dir_stats = nullptr;

// Sees every Scene::Cast of a traced path (oracle_collect_rays, oracle_classify_path); returning false abandons the path.
struct CastObserver { virtual bool operator()(const Ray& ray, const Hit& hit, const Object* obj, uint32_t cast_number) = 0; virtual ~CastObserver() {} };

// PathTracing::Thread::Render algorithm_pt.cc:125-160.  max_depth == 0 means RR-only (reference).
PathResult TracePath(const oracle_scene& sc, const Sensor& S, uint64_t px, uint64_t py, Sampler& smp,
                     const Math& M, uint32_t max_depth, oracle_bounce* trace, uint32_t max_trace,
                     float* eye_out, CastObserver* observer = nullptr) {
  const EyeRay eye = GenerateEyeRay(sc.lens, S, px, py, smp, M);
  if (eye_out) {
    eye_out[0] = eye.origin.x; eye_out[1] = eye.origin.y; eye_out[2] = eye.origin.z;
    eye_out[3] = eye.dir.x; eye_out[4] = eye.dir.y; eye_out[5] = eye.dir.z; eye_out[6] = eye.weight;
  }
  V3 measurement = splat(0);
  Ray ray{eye.origin, eye.dir};
  V3 weight = splat(eye.weight);                                        // Leading<Radiant>(.., Radiant(weight))
  PathResult r{splat(0), 0, 0, 2166136261u, 2166136261u};
  for (;;) {
    if (g_dir_stats) {
      const double dev = std::fabs(double(ray.d.x) * ray.d.x + double(ray.d.y) * ray.d.y + double(ray.d.z) * ray.d.z - 1.0);
      static const double kEdges[6] = {1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 1e-4};
      g_dir_stats->rays++;
      for (int k = 0; k < 6; k++) if (dev > kEdges[k]) g_dir_stats->above[k]++;
      if (dev > g_dir_stats->max_dev) g_dir_stats->max_dev = dev;
    }
    Hit hit; const Object* obj = nullptr;
    sc.Cast(ray, hit, obj);
    r.casts++;
    if (observer && !(*observer)(ray, hit, obj, r.casts)) break;
    r.sig_obj = Fnv32(r.sig_obj, hit ? static_cast<uint32_t>(obj->index) : 0xffffffffu);
    if (hit) r.sig_t = Fnv32(r.sig_t, f2u(hit.t));
    if (!hit) {
      if (trace && r.casts <= max_trace) { oracle_bounce& b = trace[r.casts - 1]; std::memset(&b, 0, sizeof b); b.object = -1; b.t = hit.t; }
      break;
    }
    r.hits++;
    const Material& m = sc.materials[obj->material];
    const V3 weight_before = weight;
    measurement = measurement + weight * Radiance(m, hit.n, -ray.d);
    const Scatter scat = SampleLight(m, hit.n, -ray.d, smp, M);
    const float p_rr = std::min<float>(static_cast<float>(kRussianRoulette), Max3(scat.weight));
    if (trace && r.casts <= max_trace) {
      oracle_bounce& b = trace[r.casts - 1];
      b.object = static_cast<int32_t>(obj->index); b.t = hit.t;
      b.pos[0] = hit.pos.x; b.pos[1] = hit.pos.y; b.pos[2] = hit.pos.z;
      b.weight[0] = weight_before.x; b.weight[1] = weight_before.y; b.weight[2] = weight_before.z;
      b.measurement[0] = measurement.x; b.measurement[1] = measurement.y; b.measurement[2] = measurement.z;
    }
    if (UniformF(smp) >= p_rr) break;
    if (max_depth && r.casts >= max_depth) break;                       // build-side extension (config 5)
    ray = Ray{hit.pos, scat.dir};
    weight = weight * (scat.weight / splat(p_rr));
  }
  r.measurement = measurement;
  return r;
}

}  // namespace


namespace {

struct SplatRecord { uint32_t path, sample, bounce, pixel; float rgb[3]; };

// LightTracing::Thread::Render algorithm_lt.cc:125-163 for one light path; appends the splats it makes.
uint32_t TraceLightPath(const oracle_scene& sc, const Sensor& S, uint32_t path, uint32_t sample, Sampler& smp, const Math& M,
                        uint32_t max_depth, std::vector<SplatRecord>& out) {
  const LightSet& ls = sc.light_set;
  if (ls.lights.empty()) return 0;
  // LightSet::GenerateRay light_set.h:84-104
  const float x = UniformF(ls.total(), smp);
  std::size_t pos = 0;
  while (pos < ls.lights.size() && ls.lights[pos].cum_power < x) pos++;          // std::lower_bound
  if (pos >= ls.lights.size()) pos = ls.lights.size() - 1;                        // objects_.at(pos) would throw; unreachable for u < 1
  const Light& light = ls.lights[pos];
  const Object& lobj = sc.objects[light.object];
  V3 origin, normal;
  SampleSurfacePoint(lobj, smp, M, origin, normal);
  const V3 dir = HemispherePSA(normal, smp, M);
  const float pdf_area = (light.irradiance.x + light.irradiance.y + light.irradiance.z) / ls.total();   // light_set.h:107-111
  Ray ray{origin, dir};
  V3 weight = light.irradiance / splat(pdf_area);
  const float size_f = static_cast<float>(S.w * S.h);
  uint32_t casts = 0;
  for (;;) {
    Hit hit; const Object* obj = nullptr;
    sc.Cast(ray, hit, obj);
    casts++;
    if (!hit) break;
    const Material& m = sc.materials[obj->material];
    if (m.kind == ORACLE_MAT_EYE) {                                       // algorithm_lt.cc:141-147
      uint64_t px, py; float value;
      if (LensResponse(sc.lens, S, hit.pos, -ray.d, M, px, py, value)) {
        const V3 add = (weight * splat(value)) / splat(size_f);           // weight * response.Value() / image.Size()
        out.push_back(SplatRecord{path, sample, casts, static_cast<uint32_t>(px + py * S.w), {add.x, add.y, add.z}});
      }
    }
    const Scatter scat = SampleImportance(m, hit.n, -ray.d, smp, M);
    const float p_rr = std::min<float>(static_cast<float>(kRussianRoulette), Max3(scat.weight));
    if (UniformF(smp) >= p_rr) break;
    if (max_depth && casts >= max_depth) break;
    ray = Ray{hit.pos, scat.dir};
    weight = weight * (scat.weight / splat(p_rr));
  }
  return casts;
}

}  // namespace

extern "C" {

oracle_scene* oracle_scene_create(const oracle_object* objects, uint32_t n_objects,
                                  const oracle_material* materials, uint32_t n_materials,
                                  const oracle_thin_lens* lens, int accel) {
  auto* sc = new oracle_scene();
  const bool blades_last = (accel & ORACLE_BLADES_LAST) != 0;             // cli::ImportScene order, import.cc:155-157
  sc->accel = accel & 0xff;
  const Math M{ORACLE_MATH_LIBM};
  for (uint32_t i = 0; i < n_materials; i++) {
    Material m; m.kind = materials[i].kind; m.rho = v3(materials[i].rho[0], materials[i].rho[1], materials[i].rho[2]);
    m.param = materials[i].param;
    if (m.kind == ORACLE_MAT_REFRACTION) m.r0 = Fresnel(m.param, M);
    sc->materials.push_back(m);
  }
  Material eye; eye.kind = ORACLE_MAT_EYE; eye.rho = splat(1);
  sc->materials.push_back(eye);
  const uint32_t eye_id = static_cast<uint32_t>(sc->materials.size() - 1);
  if (lens->n_blades == 0) {   // MakePinholeLens(transform, sensor_distance = focal_length), lens_pinhole.cc:31-46
    const float* t = lens->transform;
    ThinLens L;
    L.kind = 1;
    L.origin = v3(t[0] * 0.0f + t[1] * 0.0f + t[2] * 0.0f + t[3], t[4] * 0.0f + t[5] * 0.0f + t[6] * 0.0f + t[7],
                  t[8] * 0.0f + t[9] * 0.0f + t[10] * 0.0f + t[11]);
    const float g[9] = {t[0], t[1], t[2], t[4], t[5], t[6], t[8], t[9], t[10]};
    std::memcpy(L.global_.e, g, sizeof g);
    L.local_ = L.global_.Inverse();
    L.focus_distance = 0; L.sensor_distance = lens->focal_length; L.p_area = 1;
    L.blades.push_back(MakeTriangle(L.origin, L.origin, L.origin, eye_id));
    sc->lens = L;
  } else {
    sc->lens = MakeThinLens(lens->transform, lens->focal_length, lens->focus_distance, lens->radius, lens->n_blades, eye_id, M);
  }
  if (!blades_last) for (const Object& b : sc->lens.blades) sc->objects.push_back(b);     // cornel_box.cc:62-64
  for (uint32_t i = 0; i < n_objects; i++) {
    Object o; o.kind = objects[i].kind; o.material = objects[i].material;
    const float* p = objects[i].p;
    o.a = v3(p[0], p[1], p[2]);
    if (o.kind == ORACLE_PRIM_TRIANGLE) { o.b = v3(p[3], p[4], p[5]); o.c = v3(p[6], p[7], p[8]); }
    else if (o.kind == ORACLE_PRIM_SPHERE) { o.radius = p[3]; }
    else { o.b = v3(p[3], p[4], p[5]); o.radius = p[6]; o.height = p[7]; }
    FinishObject(o);
    sc->objects.push_back(o);
  }
  if (blades_last) for (const Object& b : sc->lens.blades) sc->objects.push_back(b);
  sc->Finish();
  return sc;
}

// etude::CornelBox src/amber/etude/cornel_box.cc:38-204 (literals are double in the reference and
// narrow to real_type at the Vector3 / RGB constructors).
oracle_scene* oracle_scene_cornell_box(float focal_length, float aperture_radius, uint32_t n_blades, int accel) {
  std::vector<oracle_object> objs;
  std::vector<oracle_material> mats;
  auto mat = [&](uint32_t kind, double r, double g, double b, double param) {
    oracle_material m; m.kind = kind; m.rho[0] = static_cast<float>(r); m.rho[1] = static_cast<float>(g);
    m.rho[2] = static_cast<float>(b); m.param = static_cast<float>(param); mats.push_back(m);
    return static_cast<uint32_t>(mats.size() - 1);
  };
  auto tri = [&](uint32_t m, double x0, double y0, double z0, double x1, double y1, double z1, double x2, double y2, double z2) {
    oracle_object o; o.kind = ORACLE_PRIM_TRIANGLE; o.material = m;
    const double v[9] = {x0, y0, z0, x1, y1, z1, x2, y2, z2};
    for (int i = 0; i < 9; i++) o.p[i] = static_cast<float>(v[i]);
    objs.push_back(o);
  };
  auto sph = [&](uint32_t m, double x, double y, double z, double r) {
    oracle_object o; o.kind = ORACLE_PRIM_SPHERE; o.material = m; std::memset(o.p, 0, sizeof o.p);
    o.p[0] = static_cast<float>(x); o.p[1] = static_cast<float>(y); o.p[2] = static_cast<float>(z); o.p[3] = static_cast<float>(r);
    objs.push_back(o);
  };
  uint32_t m;
  m = mat(ORACLE_MAT_DIFFUSE_LIGHT, 1e11, 1e11, 1e11, 0);              // :66-79 light source
  tri(m, 0.01, 0.99, 0.01, -0.01, 0.99, 0.01, -0.01, 0.99, -0.01);
  tri(m, -0.01, 0.99, -0.01, 0.01, 0.99, -0.01, 0.01, 0.99, 0.01);
  m = mat(ORACLE_MAT_LAMBERTIAN, .5, 0, 0, 0);                          // :81-94 left wall
  tri(m, -1, 1, 1, -1, -1, 1, -1, -1, -1);
  tri(m, -1, -1, -1, -1, 1, -1, -1, 1, 1);
  m = mat(ORACLE_MAT_LAMBERTIAN, 0, .5, 0, 0);                          // :96-109 right wall
  tri(m, 1, 1, 1, 1, 1, -1, 1, -1, -1);
  tri(m, 1, -1, -1, 1, -1, 1, 1, 1, 1);
  m = mat(ORACLE_MAT_PHONG, .95, .95, .95, 256);                        // :111-124 back wall
  tri(m, 1, 1, -1, -1, 1, -1, -1, -1, -1);
  tri(m, -1, -1, -1, 1, -1, -1, 1, 1, -1);
  m = mat(ORACLE_MAT_LAMBERTIAN, .5, .5, .5, 0);                        // :126-139 floor
  tri(m, 1, -1, 1, 1, -1, -1, -1, -1, -1);
  tri(m, -1, -1, -1, -1, -1, 1, 1, -1, 1);
  m = mat(ORACLE_MAT_LAMBERTIAN, .5, .5, .5, 0);                        // :141-154 ceiling
  tri(m, 1, 1, 1, -1, 1, 1, -1, 1, -1);
  tri(m, -1, 1, -1, 1, 1, -1, 1, 1, 1);
  m = mat(ORACLE_MAT_REFRACTION, 1, 1, 1, 1.333);                       // :156-181 water
  tri(m, 1, -0.5, 1, 1, -0.5, -1, -1, -0.5, -1);
  tri(m, -1, -0.5, -1, -1, -0.5, 1, 1, -0.5, 1);
  tri(m, 1, -0.5, 1, -1, -0.5, 1, -1, -1, 1);
  tri(m, -1, -1, 1, 1, -1, 1, 1, -0.5, 1);
  m = mat(ORACLE_MAT_LAMBERTIAN, .5, .5, .5, 0);                        // :183-186 diffuse sphere
  sph(m, 0.4, -0.6, -0.5, 0.4);
  m = mat(ORACLE_MAT_SPECULAR, .95, .95, .95, 0);                       // :188-191 specular sphere
  sph(m, -0.4, -0.7, 0.1, 0.3);
  m = mat(ORACLE_MAT_REFRACTION, 1, 1, 1, 1.125);                       // :193-196 refraction sphere
  sph(m, 0.1, -0.8, 0.6, 0.2);
  oracle_thin_lens lens;
  const float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 4, 0, 0, 0, 1};  // :50-61
  std::memcpy(lens.transform, T, sizeof T);
  lens.focal_length = focal_length; lens.focus_distance = 4; lens.radius = aperture_radius; lens.n_blades = n_blades;
  return oracle_scene_create(objs.data(), static_cast<uint32_t>(objs.size()), mats.data(),
                             static_cast<uint32_t>(mats.size()), &lens, accel);
}

void oracle_scene_destroy(oracle_scene* s) { delete s; }

uint32_t oracle_scene_object_count(const oracle_scene* s) { return static_cast<uint32_t>(s->objects.size()); }
uint32_t oracle_scene_material_count(const oracle_scene* s) { return static_cast<uint32_t>(s->materials.size()); }
void oracle_scene_get_object(const oracle_scene* s, uint32_t i, oracle_object* out, float normal_out[3]) {
  const Object& o = s->objects[i];
  std::memset(out, 0, sizeof *out);
  out->kind = o.kind; out->material = o.material;
  out->p[0] = o.a.x; out->p[1] = o.a.y; out->p[2] = o.a.z;
  if (o.kind == ORACLE_PRIM_TRIANGLE) { out->p[3] = o.b.x; out->p[4] = o.b.y; out->p[5] = o.b.z; out->p[6] = o.c.x; out->p[7] = o.c.y; out->p[8] = o.c.z; }
  else if (o.kind == ORACLE_PRIM_SPHERE) { out->p[3] = o.radius; }
  else { out->p[3] = o.b.x; out->p[4] = o.b.y; out->p[5] = o.b.z; out->p[6] = o.radius; out->p[7] = o.height; }
  normal_out[0] = o.normal.x; normal_out[1] = o.normal.y; normal_out[2] = o.normal.z;
}
void oracle_scene_get_material(const oracle_scene* s, uint32_t i, oracle_material* out, float* r0_out) {
  const Material& m = s->materials[i];
  out->kind = m.kind; out->rho[0] = m.rho.x; out->rho[1] = m.rho.y; out->rho[2] = m.rho.z; out->param = m.param;
  *r0_out = m.r0;
}
void oracle_scene_get_lens(const oracle_scene* s, float origin[3], float global_[9], float local_[9],
                           float* focus_distance, float* sensor_distance, float* p_area) {
  origin[0] = s->lens.origin.x; origin[1] = s->lens.origin.y; origin[2] = s->lens.origin.z;
  std::memcpy(global_, s->lens.global_.e, 36); std::memcpy(local_, s->lens.local_.e, 36);
  *focus_distance = s->lens.focus_distance; *sensor_distance = s->lens.sensor_distance; *p_area = s->lens.p_area;
}
void oracle_scene_bvh_stats(const oracle_scene* s, uint32_t* n_nodes, uint32_t* n_leaves, uint32_t* max_depth) {
  *n_nodes = s->bvh ? s->bvh->n_nodes.load() : 0u; *n_leaves = s->bvh ? s->bvh->n_leaves.load() : 0u; *max_depth = s->bvh ? s->bvh->max_depth.load() : 0u;
}

// The tree itself, for checks of a builder that claims to be the reference's (the engine's AMBER_ENGINE_REFERENCE_BVH): objects_ after the
// build (position -> insertion index), and FNV-1a64 over a pre-order walk -- per node a tag (1 inner, 0 leaf), the six box floats, and for a
// leaf its range [first, first + count) of objects_.
void oracle_scene_bvh_order(const oracle_scene* s, uint32_t* out) {
  if (!s->bvh) return;
  for (size_t i = 0; i < s->bvh->objects.size(); i++) out[i] = s->bvh->objects[i].index;
}
uint64_t oracle_scene_bvh_digest(const oracle_scene* s) {
  if (!s->bvh) return 0;
  uint64_t h = 14695981039346656037ull;
  auto mix = [&](const void* p, size_t n) { const unsigned char* b = static_cast<const unsigned char*>(p); for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
  std::vector<const BVH::Node*> stack{s->bvh->root.get()};
  while (!stack.empty()) {
    const BVH::Node* n = stack.back(); stack.pop_back();
    const bool leaf = n->first != n->last;
    const uint32_t tag = leaf ? 0u : 1u;
    const float bb[6] = {n->bb.mn.x, n->bb.mn.y, n->bb.mn.z, n->bb.mx.x, n->bb.mx.y, n->bb.mx.z};
    mix(&tag, 4); mix(bb, 24);
    if (leaf) {
      const uint32_t range[2] = {static_cast<uint32_t>(n->first - s->bvh->objects.begin()), static_cast<uint32_t>(n->last - n->first)};
      mix(range, 8);
    } else { stack.push_back(n->right.get()); stack.push_back(n->left.get()); }
  }
  return h;
}

// PathTracing::Render (algorithm_pt.cc:82-95) through ParallelMean (rendering/parallel.h:57-68) with
// ThreadCount()==1: passes are summed by the binary-counter Accumulator (accumulator.h:136-166) and
// divided by the pass count (Mean, :88-95).  Pixel loop: y outer, x inner (algorithm_pt.cc:112-123).
void oracle_render_mt(const oracle_scene* sc, const oracle_sensor* sensor, uint64_t seed, uint32_t spp,
                      int math, float* out_rgb, oracle_counters* counters) {
  const Sensor S{sensor->width, sensor->height, sensor->scene_width, sensor->scene_height};
  const Math M{math};
  MTSampler smp(seed);
  const std::size_t n = static_cast<std::size_t>(S.w) * S.h * 3;
  std::vector<std::vector<float>> buffer;   // Accumulator::buffer_
  std::size_t size = 0;
  oracle_counters cnt{0, 0, 0};
  for (uint32_t pass = 0; pass < spp; pass++) {
    std::vector<float> image(n, 0.0f);
    for (uint64_t y = 0; y < S.h; y++)
      for (uint64_t x = 0; x < S.w; x++) {
        const PathResult r = TracePath(*sc, S, x, y, smp, M, 0, nullptr, 0, nullptr);
        float* p = &image[(x + y * S.w) * 3];
        p[0] = r.measurement.x; p[1] = r.measurement.y; p[2] = r.measurement.z;
        cnt.casts += r.casts; cnt.hits += r.hits; cnt.paths++;
      }
    // Accumulator::Add(0, value)
    size += 1;
    for (std::size_t i = 0;; i++) {
      const std::size_t mask = static_cast<std::size_t>(1) << i;
      if (size & mask) {
        if (i == buffer.size()) buffer.emplace_back(std::move(image)); else buffer[i] = std::move(image);
        break;
      }
      for (std::size_t k = 0; k < n; k++) image[k] += buffer[i][k];     // value += buffer_[i]
    }
  }
  std::vector<float> sum(n, 0.0f);                                       // Sum(): initial_ + set bits ascending
  for (std::size_t i = 0; i < buffer.size(); i++)
    if (size & (static_cast<std::size_t>(1) << i))
      for (std::size_t k = 0; k < n; k++) sum[k] += buffer[i][k];
  const float div = static_cast<float>(size);                            // Mean(): Sum() / size_
  for (std::size_t k = 0; k < n; k++) out_rgb[k] = sum[k] / div;
  if (counters) *counters = cnt;
}

void oracle_render_xorshift(const oracle_scene* sc, const oracle_sensor* sensor, uint64_t global_seed,
                            uint32_t first_sample, uint32_t n_samples, uint32_t y0, uint32_t y1,
                            int math, uint32_t max_depth, uint32_t n_threads, uint32_t chunk,
                            float* sum_rgb, oracle_counters* counters) {
  const Sensor S{sensor->width, sensor->height, sensor->scene_width, sensor->scene_height};
  const Math M{math};
  if (n_threads == 0) n_threads = 1;
  std::vector<oracle_counters> cnts(n_threads, oracle_counters{0, 0, 0});
  auto work = [&](uint32_t tid) {
    oracle_counters cnt{0, 0, 0};        // thread-local: the threads' slots of `cnts` share cache lines, and three updates per path on a shared line held 16 threads to 7x one
    for (uint64_t y = y0 + tid; y < y1; y += n_threads)
      for (uint64_t x = 0; x < S.w; x++) {
        const uint32_t pixel = static_cast<uint32_t>(x + y * S.w);
        float* p = &sum_rgb[static_cast<std::size_t>(pixel) * 3];
        // accumulation granule (include/amber_hip.h AMBER_ACCUM_CHUNK): sequential sum inside a chunk of
        // `chunk` samples, chunk sums added to the running pixel value in chunk order; 0 = one chunk
        const uint32_t step = chunk ? chunk : n_samples;
        for (uint32_t c0 = 0; c0 < n_samples; c0 += step) {
          V3 sum = splat(0);
          const uint32_t c1 = c0 + step < n_samples ? c0 + step : n_samples;
          for (uint32_t s = first_sample + c0; s < first_sample + c1; s++) {
            XorShiftSampler smp(XorShiftSeed(global_seed, pixel, s));
            const PathResult r = TracePath(*sc, S, x, y, smp, M, max_depth, nullptr, 0, nullptr);
            sum = sum + r.measurement;
            cnt.casts += r.casts; cnt.hits += r.hits; cnt.paths++;
          }
          p[0] += sum.x; p[1] += sum.y; p[2] += sum.z;
        }
      }
    cnts[tid] = cnt;
  };
  std::vector<std::thread> threads;
  for (uint32_t t = 1; t < n_threads; t++) threads.emplace_back(work, t);
  work(0);
  for (auto& t : threads) t.join();
  if (counters) {
    oracle_counters tot{0, 0, 0};
    for (const auto& c : cnts) { tot.casts += c.casts; tot.hits += c.hits; tot.paths += c.paths; }
    *counters = tot;
  }
}

void oracle_path_signatures(const oracle_scene* sc, const oracle_sensor* sensor, uint64_t global_seed, uint32_t first_sample,
                            uint32_t n_samples, uint32_t y0, uint32_t y1, int math, uint32_t max_depth, uint32_t n_threads,
                            uint64_t* sig) {
  const Sensor S{sensor->width, sensor->height, sensor->scene_width, sensor->scene_height};
  const Math M{math};
  if (n_threads == 0) n_threads = 1;
  auto work = [&](uint32_t tid) {
    for (uint64_t y = y0 + tid; y < y1; y += n_threads)
      for (uint64_t x = 0; x < S.w; x++) {
        const uint32_t pixel = static_cast<uint32_t>(x + y * S.w);
        uint64_t* out = sig + ((y - y0) * S.w + x) * n_samples;
        for (uint32_t k = 0; k < n_samples; k++) {
          XorShiftSampler smp(XorShiftSeed(global_seed, pixel, first_sample + k));
          const PathResult r = TracePath(*sc, S, x, y, smp, M, max_depth, nullptr, 0, nullptr);
          out[k] = static_cast<uint64_t>(r.sig_obj) | (static_cast<uint64_t>(r.sig_t) << 32);
        }
      }
  };
  std::vector<std::thread> threads;
  for (uint32_t t = 1; t < n_threads; t++) threads.emplace_back(work, t);
  work(0);
  for (auto& t : threads) t.join();
}

uint32_t oracle_trace_path(const oracle_scene* sc, const oracle_sensor* sensor, uint64_t global_seed,
                           uint32_t px, uint32_t py, uint32_t sample, int math, uint32_t max_depth,
                           oracle_bounce* out, uint32_t max_bounces, float eye_ray_out[7]) {
  const Sensor S{sensor->width, sensor->height, sensor->scene_width, sensor->scene_height};
  const Math M{math};
  XorShiftSampler smp(XorShiftSeed(global_seed, static_cast<uint32_t>(px + py * S.w), sample));
  return TracePath(*sc, S, px, py, smp, M, max_depth, out, max_bounces, eye_ray_out).casts;
}

int32_t oracle_cast(const oracle_scene* sc, const float origin[3], const float dir[3], float* t, float pos[3], float normal[3]) {
  const Ray ray{v3(origin[0], origin[1], origin[2]), v3(dir[0], dir[1], dir[2])};
  Hit hit; const Object* obj = nullptr;
  sc->Cast(ray, hit, obj);
  *t = hit.t; pos[0] = hit.pos.x; pos[1] = hit.pos.y; pos[2] = hit.pos.z; normal[0] = hit.n.x; normal[1] = hit.n.y; normal[2] = hit.n.z;
  return hit ? static_cast<int32_t>(obj->index) : -1;
}

int oracle_scene_set_accel(oracle_scene* sc, int accel) {
  if (accel == ORACLE_ACCEL_LIST) { sc->accel = accel; return 0; }
  if (accel == ORACLE_ACCEL_BVH && sc->bvh) { sc->accel = accel; return 0; }
  if (accel == ORACLE_ACCEL_BVH_CONS && sc->bvh && sc->bvh->has_cons) { sc->accel = accel; return 0; }
  return -1;
}

namespace {
// Eight spheres against one ray: bit k set unless the discriminant of sphere k is negative -- b, c and b*b - 4*a*c exactly as
// IntersectSphere / SolveQuadratic form them (a = 1: 4*a*c is 4*c), one binary32 rounding per operation, no contraction.
inline int SphereDiscriminantNotNegative8(const float* cx, const float* cy, const float* cz, const float* rad, __m256 ox, __m256 oy, __m256 oz,
                                          __m256 dx, __m256 dy, __m256 dz) {
  const __m256 x = _mm256_sub_ps(_mm256_loadu_ps(cx), ox), y = _mm256_sub_ps(_mm256_loadu_ps(cy), oy), z = _mm256_sub_ps(_mm256_loadu_ps(cz), oz);
  const __m256 dot = _mm256_add_ps(_mm256_add_ps(_mm256_mul_ps(x, dx), _mm256_mul_ps(y, dy)), _mm256_mul_ps(z, dz));
  const __m256 b = _mm256_mul_ps(_mm256_set1_ps(-2.0f), dot);
  const __m256 sq = _mm256_add_ps(_mm256_add_ps(_mm256_mul_ps(x, x), _mm256_mul_ps(y, y)), _mm256_mul_ps(z, z));
  const __m256 r = _mm256_loadu_ps(rad);
  const __m256 c = _mm256_sub_ps(sq, _mm256_mul_ps(r, r));
  const __m256 disc = _mm256_sub_ps(_mm256_mul_ps(b, b), _mm256_mul_ps(_mm256_set1_ps(4.0f), c));
  return _mm256_movemask_ps(_mm256_cmp_ps(disc, _mm256_setzero_ps(), _CMP_NLT_UQ));   // !(disc < 0): NaN goes on, like "if (d < 0) return false"
}
}  // namespace

void oracle_cast_many(const oracle_scene* sc, int accel, uint64_t n, const float* origins, const float* dirs, uint32_t n_threads,
                      int32_t* object_out, float* t_out) {
  if (n_threads == 0) n_threads = 1;
  auto ray_of = [&](uint64_t i) { return Ray{v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]), v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2])}; };
  const bool blocked = accel == ORACLE_ACCEL_LIST && sc->objects.size() >= 4096;
  if (!blocked) {
    auto work = [&](uint32_t tid) {
      for (uint64_t i = tid; i < n; i += n_threads) {
        Hit hit; const Object* obj = nullptr;
        sc->CastWith(accel, ray_of(i), hit, obj);
        object_out[i] = hit ? static_cast<int32_t>(obj->index) : -1; t_out[i] = hit.t;
      }
    };
    std::vector<std::thread> threads;
    for (uint32_t t = 1; t < n_threads; t++) threads.emplace_back(work, t);
    work(0);
    for (auto& t : threads) t.join();
    return;
  }
  // acceleration_list.h:51-68 over a large scene: blocks of objects (outer) against a thread's rays (inner).  Objects are met in
  // insertion order by every ray, so "hit.t < distance" keeps the lowest index of a tie exactly as the scan does.
  const std::vector<Object>& objs = sc->objects;
  constexpr std::size_t kBlock = 2048;
  auto work = [&](uint32_t tid) {
    const uint64_t i0 = n * tid / n_threads, i1 = n * (tid + 1) / n_threads;
    std::vector<float> dist(i1 - i0, FLT_MAX);
    for (uint64_t i = i0; i < i1; i++) { object_out[i] = -1; t_out[i] = Hit().t; }
    std::vector<float> cx(kBlock + 8), cy(kBlock + 8), cz(kBlock + 8), rad(kBlock + 8);
    std::vector<uint32_t> sphere_at(kBlock + 8), other;
    for (std::size_t b0 = 0; b0 < objs.size(); b0 += kBlock) {
      const std::size_t b1 = std::min(objs.size(), b0 + kBlock);
      std::size_t ns = 0; other.clear();
      for (std::size_t k = b0; k < b1; k++) {
        if (objs[k].kind == ORACLE_PRIM_SPHERE) { cx[ns] = objs[k].a.x; cy[ns] = objs[k].a.y; cz[ns] = objs[k].a.z; rad[ns] = objs[k].radius; sphere_at[ns++] = static_cast<uint32_t>(k); }
        else other.push_back(static_cast<uint32_t>(k));
      }
      const std::size_t ns8 = (ns + 7) / 8 * 8;
      for (std::size_t k = ns; k < ns8; k++) { cx[k] = cy[k] = cz[k] = 0; rad[k] = 0; sphere_at[k] = 0xffffffffu; }
      std::vector<uint32_t> cand;
      for (uint64_t i = i0; i < i1; i++) {
        const Ray ray = ray_of(i);
        cand = other;
        const __m256 ox = _mm256_set1_ps(ray.o.x), oy = _mm256_set1_ps(ray.o.y), oz = _mm256_set1_ps(ray.o.z);
        const __m256 dx = _mm256_set1_ps(ray.d.x), dy = _mm256_set1_ps(ray.d.y), dz = _mm256_set1_ps(ray.d.z);
        for (std::size_t k = 0; k < ns8; k += 8) {
          int m = SphereDiscriminantNotNegative8(&cx[k], &cy[k], &cz[k], &rad[k], ox, oy, oz, dx, dy, dz);
          while (m) { const int j = __builtin_ctz(m); m &= m - 1; if (sphere_at[k + j] != 0xffffffffu) cand.push_back(sphere_at[k + j]); }
        }
        std::sort(cand.begin(), cand.end());                              // insertion order
        float& distance = dist[i - i0];
        for (uint32_t k : cand) {
          const Hit h = Intersect(objs[k], ray);
          if (h && h.t < distance) { distance = h.t; object_out[i] = static_cast<int32_t>(objs[k].index); t_out[i] = h.t; }
        }
      }
    }
  };
  std::vector<std::thread> threads;
  for (uint32_t t = 1; t < n_threads; t++) threads.emplace_back(work, t);
  work(0);
  for (auto& t : threads) t.join();
}

uint64_t oracle_collect_rays(const oracle_scene* sc, const oracle_sensor* sensor, uint64_t global_seed, uint32_t first_sample, uint32_t n_samples,
                             uint32_t y0, uint32_t y1, int math, uint32_t max_depth, uint64_t max_rays, float* origins, float* dirs) {
  const Sensor S{sensor->width, sensor->height, sensor->scene_width, sensor->scene_height};
  const Math M{math};
  struct Collect : CastObserver {
    uint64_t n = 0, cap = 0; float* o = nullptr; float* d = nullptr;
    bool operator()(const Ray& ray, const Hit&, const Object*, uint32_t) override {
      if (n < cap) { o[3 * n] = ray.o.x; o[3 * n + 1] = ray.o.y; o[3 * n + 2] = ray.o.z; d[3 * n] = ray.d.x; d[3 * n + 1] = ray.d.y; d[3 * n + 2] = ray.d.z; }
      n++; return true;
    }
  } collect;
  collect.cap = max_rays; collect.o = origins; collect.d = dirs;
  for (uint64_t y = y0; y < y1; y++)
    for (uint64_t x = 0; x < S.w; x++)
      for (uint32_t k = 0; k < n_samples; k++) {
        XorShiftSampler smp(XorShiftSeed(global_seed, static_cast<uint32_t>(x + y * S.w), first_sample + k));
        TracePath(*sc, S, x, y, smp, M, max_depth, nullptr, 0, nullptr, &collect);
      }
  return collect.n;
}

int oracle_classify_path(const oracle_scene* sc, const oracle_sensor* sensor, uint64_t global_seed, uint32_t px, uint32_t py, uint32_t sample,
                         int math, uint32_t max_depth, int accel_a, int accel_b, uint32_t out[10]) {
  const Sensor S{sensor->width, sensor->height, sensor->scene_width, sensor->scene_height};
  const Math M{math};
  struct Classify : CastObserver {
    const oracle_scene* sc = nullptr; int b = 0; uint32_t* out = nullptr; bool found = false;
    bool operator()(const Ray& ray, const Hit& ha, const Object* oa, uint32_t cast_number) override {
      Hit hb; const Object* ob = nullptr;
      sc->CastWith(b, ray, hb, ob);
      const uint32_t ia = ha ? oa->index : 0xffffffffu, ib = hb ? ob->index : 0xffffffffu;
      if (ia == ib && (!ha || f2u(ha.t) == f2u(hb.t))) return true;
      Hit hl; const Object* ol = nullptr;
      sc->CastWith(ORACLE_ACCEL_LIST, ray, hl, ol);
      out[0] = cast_number; out[1] = ia; out[2] = ib; out[3] = hl ? ol->index : 0xffffffffu;
      out[4] = f2u(ha.t); out[5] = f2u(hb.t); out[6] = f2u(hl.t);
      float tin, tout;
      out[7] = hl && SlabTest(BoundingBox(*ol), ray, FLT_MAX, tin, tout) ? 1u : 0u;
      out[8] = (ha && hb && f2u(ha.t) == f2u(hb.t)) ? 1u : 0u;
      out[9] = hl && SlabTest(BoundingBox(*ol), ray, hl.t, tin, tout) ? 1u : 0u;
      found = true;
      return false;
    }
  } classify;
  classify.sc = sc; classify.b = accel_b; classify.out = out;
  // TracePath casts through sc->accel: switched for the duration of this call (single-threaded tool, see the header)
  XorShiftSampler smp(XorShiftSeed(global_seed, static_cast<uint32_t>(px + py * S.w), sample));
  struct AccelGuard { oracle_scene* s; int old; ~AccelGuard() { s->accel = old; } } guard{const_cast<oracle_scene*>(sc), sc->accel};
  const_cast<oracle_scene*>(sc)->accel = accel_a;
  TracePath(*sc, S, px, py, smp, M, max_depth, nullptr, 0, nullptr, &classify);
  return classify.found ? 1 : 0;
}

int oracle_intersect(const oracle_object* obj, const float origin[3], const float dir[3], float* t, float pos[3], float normal[3]) {
  Object o; o.kind = obj->kind; o.material = obj->material; const float* p = obj->p;
  o.a = v3(p[0], p[1], p[2]);
  if (o.kind == ORACLE_PRIM_TRIANGLE) { o.b = v3(p[3], p[4], p[5]); o.c = v3(p[6], p[7], p[8]); }
  else if (o.kind == ORACLE_PRIM_SPHERE) { o.radius = p[3]; }
  else { o.b = v3(p[3], p[4], p[5]); o.radius = p[6]; o.height = p[7]; }
  FinishObject(o);
  const Hit hit = Intersect(o, Ray{v3(origin[0], origin[1], origin[2]), v3(dir[0], dir[1], dir[2])});
  *t = hit.t; pos[0] = hit.pos.x; pos[1] = hit.pos.y; pos[2] = hit.pos.z; normal[0] = hit.n.x; normal[1] = hit.n.y; normal[2] = hit.n.z;
  return hit ? 1 : 0;
}

int oracle_aabb_intersect(const float bmin[3], const float bmax[3], const float origin[3], const float dir[3],
                          float t_max, float* t_in, float* t_out) {
  const AABB bb{v3(bmin[0], bmin[1], bmin[2]), v3(bmax[0], bmax[1], bmax[2])};
  return SlabTest(bb, Ray{v3(origin[0], origin[1], origin[2]), v3(dir[0], dir[1], dir[2])}, t_max, *t_in, *t_out) ? 1 : 0;
}

uint32_t oracle_sample_material(const oracle_material* m, const float normal[3], const float dir_out[3],
                                const double* u, uint32_t n_u, int math, float dir_in[3], float weight[3]) {
  const Math M{math};
  Material mm; mm.kind = m->kind; mm.rho = v3(m->rho[0], m->rho[1], m->rho[2]); mm.param = m->param;
  if (mm.kind == ORACLE_MAT_REFRACTION) mm.r0 = Fresnel(mm.param, Math{ORACLE_MATH_LIBM});
  ArraySampler smp(u, n_u);
  const Scatter s = SampleLight(mm, v3(normal[0], normal[1], normal[2]), v3(dir_out[0], dir_out[1], dir_out[2]), smp, M);
  dir_in[0] = s.dir.x; dir_in[1] = s.dir.y; dir_in[2] = s.dir.z; weight[0] = s.weight.x; weight[1] = s.weight.y; weight[2] = s.weight.z;
  return smp.i;
}
void oracle_radiance(const oracle_material* m, const float normal[3], const float dir_out[3], float out[3]) {
  Material mm; mm.kind = m->kind; mm.rho = v3(m->rho[0], m->rho[1], m->rho[2]); mm.param = m->param;
  const V3 r = Radiance(mm, v3(normal[0], normal[1], normal[2]), v3(dir_out[0], dir_out[1], dir_out[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oracle_eye_ray(const oracle_scene* sc, const oracle_sensor* sensor, uint32_t px, uint32_t py, const double u[5],
                    int math, float origin[3], float normal[3], float dir[3], float* weight, uint32_t* blade) {
  const Sensor S{sensor->width, sensor->height, sensor->scene_width, sensor->scene_height};
  ArraySampler smp(u, 5);
  const EyeRay e = GenerateEyeRay(sc->lens, S, px, py, smp, Math{math});
  origin[0] = e.origin.x; origin[1] = e.origin.y; origin[2] = e.origin.z;
  normal[0] = e.normal.x; normal[1] = e.normal.y; normal[2] = e.normal.z;
  dir[0] = e.dir.x; dir[1] = e.dir.y; dir[2] = e.dir.z; *weight = e.weight; *blade = e.blade;
}

void oracle_mt_uniforms(uint64_t seed, uint32_t n, uint64_t* raw, double* u, float* uf) {
  std::mt19937_64 e(seed); MTSampler s(seed);
  for (uint32_t i = 0; i < n; i++) { raw[i] = e(); u[i] = s(); uf[i] = static_cast<float>(u[i]); }
}
uint64_t oracle_xorshift_seed(uint64_t global_seed, uint32_t pixel, uint32_t sample) { return XorShiftSeed(global_seed, pixel, sample); }
void oracle_xorshift_uniforms(uint64_t state, uint32_t n, double* u) { XorShiftSampler s(state); for (uint32_t i = 0; i < n; i++) u[i] = s(); }

void oracle_sincos(float phi, int math, float* s, float* c) { Math{math}.sincos(phi, *s, *c); }
float oracle_pow(float x, float y, int math) { return Math{math}.powf_(x, y); }
double oracle_pow_i(float x, int n, int math) { return Math{math}.pow_i(x, n); }
uint64_t oracle_math_compare(int kind, int mode_a, int mode_b, uint32_t k0, uint32_t k1, float y, float first_bad[2]) {
  const Math A{mode_a}, B{mode_b};
  uint64_t bad = 0;
  uint64_t rs = SplitMix64(k0);
  auto next = [&rs]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
  for (uint32_t k = k0; k < k1; k++) {
    float a0 = 0, a1 = 0;
    bool differ;
    if (kind == 0 || kind == 2) {
      a0 = kind == 0 ? (2.0f * kPIf) * (static_cast<float>(k) * 0x1p-24f)
                     : (static_cast<float>(static_cast<double>(next() >> 11) * 0x1p-53) - 0.5f) * 239.9f;
      float sa, ca, sb, cb;
      A.sincos(a0, sa, ca); B.sincos(a0, sb, cb);
      differ = f2u(sa) != f2u(sb) || f2u(ca) != f2u(cb);
    } else {
      if (kind == 1) { a0 = static_cast<float>(k) * 0x1p-24f; a1 = y; }
      else {   // random positive x over the whole exponent range, y in (-y, y)
        a0 = u2f(static_cast<uint32_t>(next() >> 33));                         // [0, 2^31): +0 .. NaN patterns of positive sign
        a1 = (static_cast<float>(static_cast<double>(next() >> 11) * 0x1p-53) - 0.5f) * 2.0f * y;
      }
      const float pa = A.powf_(a0, a1), pb = B.powf_(a0, a1);
      differ = f2u(pa) != f2u(pb) && !(pa != pa && pb != pb);                  // any NaN equals any NaN
    }
    if (differ) { if (!bad && first_bad) { first_bad[0] = a0; first_bad[1] = a1; } bad++; }
  }
  return bad;
}

// Light tracing in XorShift mode: passes [first_sample, first_sample+n); every pass traces W*H light paths (path i
// seeded by (lt_seed, i, pass)) and adds their splats to a zeroed pass image in path order; pass images are added to
// sum_rgb in pass order.  Returns the number of splats; optionally copies the first max_records of them (7 x u32 each:
// path, sample, bounce, pixel, rgb bits) in accumulation order.
uint64_t oracle_render_lt_xorshift(const oracle_scene* sc, const oracle_sensor* sensor, uint64_t global_seed, uint32_t first_sample,
                                   uint32_t n_samples, int math, uint32_t max_depth, float* sum_rgb, oracle_counters* counters,
                                   uint32_t* records, uint64_t max_records) {
  const Sensor S{sensor->width, sensor->height, sensor->scene_width, sensor->scene_height};
  const Math M{math};
  const uint64_t lt_seed = global_seed + 0x6C74ull;                      // decorrelated from the path tracer's streams
  const uint32_t n_paths = static_cast<uint32_t>(S.w * S.h);
  oracle_counters cnt{0, 0, 0};
  uint64_t n_rec = 0;
  std::vector<float> pass(static_cast<std::size_t>(n_paths) * 3);
  for (uint32_t s = first_sample; s < first_sample + n_samples; s++) {
    std::vector<SplatRecord> recs;
    for (uint32_t i = 0; i < n_paths; i++) {
      XorShiftSampler smp(XorShiftSeed(lt_seed, i, s));
      cnt.casts += TraceLightPath(*sc, S, i, s, smp, M, max_depth, recs);
      cnt.paths++;
    }
    std::fill(pass.begin(), pass.end(), 0.0f);
    for (const SplatRecord& r : recs) {
      float* p = &pass[static_cast<std::size_t>(r.pixel) * 3];
      p[0] += r.rgb[0]; p[1] += r.rgb[1]; p[2] += r.rgb[2];
      if (records && n_rec < max_records) {
        uint32_t* q = records + n_rec * 7;
        q[0] = r.path; q[1] = r.sample; q[2] = r.bounce; q[3] = r.pixel; std::memcpy(q + 4, r.rgb, 12);
      }
      n_rec++;
    }
    if (!recs.empty()) for (std::size_t k = 0; k < pass.size(); k++) sum_rgb[k] += pass[k];
  }
  if (counters) *counters = cnt;
  return n_rec;
}

// Output stage (application.cc:98-108): postprocess::Filmic (filmic.cc:30-66) then postprocess::Gamma(2.2) (gamma.cc:36-52).
// hdr_value_type = float_t = float (postprocess/forward.h:33): every operation below is one binary32 operation in the
// reference's order (Vector3 op scalar is component-wise, vector3.h; the constant products kB*kC, kD*kE, kD*kF and kE/kF
// are binary32 constant expressions); std::pow(float, float) is glibc's powf; std::min<float>(1, x) returns 1 for a NaN x;
// the LDR constructor converts float -> uint_fast8_t by truncation (255 * [0, 1] is always in range).
void oracle_tonemap(const float* rgb, uint32_t width, uint32_t height, uint8_t* out_rgb8) {
  const float kA = 0.22f, kB = 0.30f, kC = 0.10f, kD = 0.20f, kE = 0.01f, kF = 0.30f, kW = 0.70f, kExposure = 16.0f;
  const float bc = kB * kC, de = kD * kE, df = kD * kF, ef = kE / kF;
  auto map = [&](float h) {                                             // Filmic::Map filmic.cc:59-66
    const float num = h * (h * kA + bc) + de;
    const float den = h * (h * kA + kB) + df;
    return num / den - ef;
  };
  const float white = map(kW);
  const float inv_gamma = 1 / 2.2f;                                      // 1 / gamma_ (gamma.cc:47-49)
  const std::size_t n = static_cast<std::size_t>(width) * height * 3;
  for (std::size_t k = 0; k < n; k++) {
    const float mapped = map(rgb[k] * kExposure) / white;               // filmic.cc:52
    const float v = 255 * std::min<float>(1, std::pow(mapped, inv_gamma));
    out_rgb8[k] = v >= 0 ? static_cast<uint8_t>(v) : 0;
  }
}

uint64_t oracle_fnv1a64(const void* data, uint64_t n_bytes) {
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint64_t h = 14695981039346656037ull;
  for (uint64_t i = 0; i < n_bytes; i++) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}

}  // extern "C"


// Diagnostic: | |d|^2 - 1 | of every ray of a single-threaded XorShift-mode render.  out[0] = rays, out[1..6] = rays
// above 1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 1e-4, out[7] = maximum.
extern "C" void oracle_direction_length_stats(const oracle_scene* sc, const oracle_sensor* sensor, uint64_t seed, uint32_t first_sample,
                                              uint32_t n_samples, int math, double out[8]) {
  const Sensor S{sensor->width, sensor->height, sensor->scene_width, sensor->scene_height};
  const Math M{math};
  DirStats st;
  g_dir_stats = &st;
  for (uint32_t y = 0; y < S.h; y++)
    for (uint32_t x = 0; x < S.w; x++)
      for (uint32_t s = first_sample; s < first_sample + n_samples; s++) {
        XorShiftSampler smp(XorShiftSeed(seed, x + y * S.w, s));
        (void)TracePath(*sc, S, x, y, smp, M, 0, nullptr, 0, nullptr);
      }
  g_dir_stats = nullptr;
  out[0] = static_cast<double>(st.rays);
  for (int k = 0; k < 6; k++) out[1 + k] = static_cast<double>(st.above[k]);
  out[7] = st.max_dev;
}
