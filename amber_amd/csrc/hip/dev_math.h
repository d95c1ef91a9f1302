// dev_math.h -- part of pt_device.h (included from there, in order; not a stand-alone header): Vector3, the per-(pixel, sample) XorShift sampler, glibc's sincosf / powf kernels restated for the device.
#pragma once

namespace amber_dev {

// ---------------------------------------------------------------------------------------------
// Vector3 (include/amber/prelude/vector3.h:36-342): component-wise ops, scalar splat
// ---------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
// Wave-uniform reads of scene records go through the constant address space: hipcc then emits scalar loads
// (s_load_dwordx4/x8/x16, operands in SGPRs) even though the kernel also stores to global memory in its loop.
typedef const uint32_t __attribute__((address_space(4)))* ConstWords;
typedef float F2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float cw_f(ConstWords w, int k) { return __uint_as_float(w[k]); }
__device__ __forceinline__ F2 cw_f2(ConstWords w, int k) { return F2{cw_f(w, k), cw_f(w, k + 1)}; }
__device__ __forceinline__ V3 cw_v3(ConstWords w, int k) { return V3{cw_f(w, k), cw_f(w, k + 1), cw_f(w, k + 2)}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 v) { return V3{s * v.x, s * v.y, s * v.z}; }
__device__ __forceinline__ V3 operator*(V3 v, float s) { return V3{v.x * s, v.y * s, v.z * s}; }
__device__ __forceinline__ V3 operator/(V3 v, float s) { return V3{v.x / s, v.y / s, v.z / s}; }
__device__ __forceinline__ V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ float Dot(V3 u, V3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
__device__ __forceinline__ float SquaredLength(V3 v) { return Dot(v, v); }
// __builtin_sqrtf lowers to the correctly rounded sequence (v_sqrt_f32 + one-ulp fix-up); __fsqrt_rn does NOT on ROCm 7.2
__device__ __forceinline__ float Sqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ V3 Normalize(V3 v) { const float l = Sqrt(SquaredLength(v)); return V3{v.x / l, v.y / l, v.z / l}; }
__device__ __forceinline__ V3 Cross(V3 u, V3 v) {
  return V3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}
// std::max({x,y,z}) (vector3.h:276-281): first element wins unless a later one is strictly greater
__device__ __forceinline__ float Max3(V3 v) { float m = v.x; if (m < v.y) m = v.y; if (m < v.z) m = v.z; return m; }
__device__ __forceinline__ float Abs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ void OrthonormalBasis(V3 w, V3& u, V3& v) {      // vector3.h:330-342
  const bool xs = Abs(w.x) < Abs(w.y);
  u = Normalize(Cross(w, xs ? v3(1.f, 0.f, 0.f) : v3(0.f, 1.f, 0.f)));
  v = Normalize(Cross(w, u));
}
__device__ __forceinline__ V3 MatMul(const float* e, V3 v) {               // matrix3.h:101-109
  return v3(e[0] * v.x + e[1] * v.y + e[2] * v.z, e[3] * v.x + e[4] * v.y + e[5] * v.z, e[6] * v.x + e[7] * v.y + e[8] * v.z);
}
__device__ __forceinline__ bool IsFinite(float x) { return Abs(x) < __builtin_inff(); }   // false for NaN and inf

// ---------------------------------------------------------------------------------------------
// per-(pixel,sample) XorShift sampler (DESIGN.md "Sampler"): splitmix64-hashed seed, Marsaglia
// xorshift64 (13,7,17), uniform = top 24 bits * 2^-24 -- exact in binary32, never 1.0.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t SplitMix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t XorShiftSeed(uint64_t hashed_global_seed, uint32_t pixel, uint32_t sample) {
  const uint64_t key = (static_cast<uint64_t>(pixel) << 32) | sample;
  const uint64_t s = SplitMix64(hashed_global_seed ^ key);
  return s ? s : 0x9E3779B97F4A7C15ull;
}
__device__ __forceinline__ float Uniform(uint64_t& s) {
  s ^= s << 13; s ^= s >> 7; s ^= s << 17;
  return static_cast<float>(static_cast<uint32_t>(s >> 40)) * 0x1p-24f;
}

// ---------------------------------------------------------------------------------------------
// sin / cos / pow.  The reference calls glibc: g++ -O2 merges std::cos(phi), std::sin(phi) (sampling.h:249-250, 283-284)
// into ONE sincosf call, std::pow(r0, 1 / (e + 1)) (sampling.h:279) is powf.  glibc 2.35's binary32 functions are
// double-precision kernels (the algorithms and tables of ARM's optimized-routines: s_sincosf.c, sincosf_poly.h, e_powf.c,
// e_powf_log2_data.c, e_exp2f_data.c) whose x86-64 FMA variant -- the one every FMA-capable host selects -- fuses every
// a*b+c of the source and rounds to binary32 once at the end.  gfx950 has IEEE binary64 mul/fma, so the engine executes
// exactly those operations: bit-identical to the live libm for every argument the path can produce (the oracle's
// GLIBC mode is the same restatement, proven equal to libm.so.6 over the whole argument set in tests/test_math_modes.py).
// Domain of SinCos: |x| < 120 (the path needs [0, 2 pi]); NaN beyond.
// -DAMBER_BUILD_PORTABLE_MATH builds round 1's + - * / forms instead (Cephes-style; 29 % of the 2^24 possible phi differ from
// glibc in the last bit) -- kept only to measure the distance between the two.
// ---------------------------------------------------------------------------------------------
#ifdef AMBER_BUILD_PORTABLE_MATH
#define AMBER_MATH_MODE 1
__device__ __forceinline__ void SinCos(float x, float& s_out, float& c_out) {
  const float FOPI = 1.27323954473516f;
  const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
  int j = static_cast<int>(FOPI * x);
  j += (j & 1);
  const float y = static_cast<float>(j);
  const float r = ((x - y * DP1) - y * DP2) - y * DP3;
  const float z = r * r;
  const float ps = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * r + r;
  const float pc = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z
                   - 0.5f * z + 1.0f;
  const int q = (j >> 1) & 3;
  const float s = (q & 1) ? pc : ps;
  const float c = (q & 1) ? ps : pc;
  s_out = (q & 2) ? -s : s;
  c_out = (q == 1 || q == 2) ? -c : c;
}
__device__ __forceinline__ float Pow(float x, float y) {
  if (y == 0.0f) return 1.0f;
  if (x == 0.0f) return y > 0.0f ? 0.0f : __builtin_inff();
  if (x == 1.0f) return 1.0f;
  uint32_t bits = __float_as_uint(x);
  int e = static_cast<int>((bits >> 23) & 0xff);
  if (e == 0) { x = x * 16777216.0f; bits = __float_as_uint(x); e = static_cast<int>((bits >> 23) & 0xff) - 24; }
  e -= 127;
  float m = __uint_as_float((bits & 0x007fffffu) | 0x3f800000u);
  if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
  const float f = m - 1.0f;
  const float s = f / (2.0f + f);
  const float z = s * s;
  float p = 0.0909090909f;
  p = p * z + 0.111111111f;
  p = p * z + 0.142857143f;
  p = p * z + 0.2f;
  p = p * z + 0.333333333f;
  p = p * z + 1.0f;
  const float ln_m = 2.0f * s * p;
  const float log2x = static_cast<float>(e) + ln_m * 1.44269504f;
  const float w = y * log2x;
  if (w >= 128.0f) return __builtin_inff();
  if (w < -149.0f) return 0.0f;
  const float nf = __builtin_floorf(w + 0.5f);
  const float g = w - nf;
  const float t = g * 0.693147181f;
  float q = 1.98412698e-4f;
  q = q * t + 1.38888889e-3f;
  q = q * t + 8.33333333e-3f;
  q = q * t + 4.16666667e-2f;
  q = q * t + 1.66666667e-1f;
  q = q * t + 0.5f;
  q = q * t + 1.0f;
  q = q * t + 1.0f;
  int n = static_cast<int>(nf);
  if (n < -126) { q = q * __uint_as_float(static_cast<uint32_t>(n + 126 + 127) << 23); n = -126; }
  return q * __uint_as_float(static_cast<uint32_t>(n + 127) << 23);
}
#else
#define AMBER_MATH_MODE 2
__device__ const double kGlibcLog2Tab[16][2] = {   // __powf_log2_data.tab: {invc, logc}
  {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
  {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
  {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
  {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
  {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
  {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
  {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
  {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
__device__ const uint64_t kGlibcExp2Tab[32] = {    // __exp2f_data.tab
  0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
  0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
  0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
  0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
  0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
  0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};
// __sincosf (s_sincosf.c) for |y| < 120.  The source's two short paths are folded into reduce_fast, which computes the
// same values there: for |y| < 0.75 it finds n = 0 and x - 0 * hpi = x exactly.  Its second coefficient table (n & 2)
// holds the negated cosine coefficients and sign[] = {1, -1, -1, 1} flips x: round-to-nearest is symmetric, so the
// first table's results with the signs applied afterwards are the same bits.
__device__ __forceinline__ void SinCos(float y, float& s_out, float& c_out) {
  const double x0 = static_cast<double>(y);
  const double r = x0 * 0x1.45f306dc9c883p+23;                                  // hpi_inv = 2^24 * 2 / pi
  const int n = (static_cast<int>(r) + 0x800000) >> 24;                          // nearest multiple of pi / 2
  const double x = __builtin_fma(-static_cast<double>(n), 0x1.921fb54442d18p+0, x0);
  const double x2 = x * x;
  const double s1 = __builtin_fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);    // sincosf_poly (sysdeps/x86_64/fpu/sincosf_poly.h)
  const double c2 = __builtin_fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
  const double c1 = __builtin_fma(x2, -0x1.ffffffd0c621cp-2, 1.0);
  const double x3 = x2 * x, x4 = x2 * x2;
  const double x5 = x2 * x3, x6 = x2 * x4;
  const double s = __builtin_fma(x3, -0x1.555545995a603p-3, x);
  const double c = __builtin_fma(x4, 0x1.55553e1068f19p-5, c1);
  float sv = static_cast<float>(__builtin_fma(x5, s1, s));
  float cv = static_cast<float>(__builtin_fma(x6, c2, c));
  if ((n + 1) & 2) sv = -sv;                                                     // sign[n & 3]
  if (n & 2) cv = -cv;                                                           // table[1]
  const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;                     // abstop12
  if (top < 0x398u) { sv = y; cv = 1.0f; }                                       // |y| < 2^-12: sin = y, cos = 1 (n = 0)
  s_out = (n & 1) ? cv : sv;
  c_out = (n & 1) ? sv : cv;
  if (!(top < 0x42fu)) { s_out = __builtin_nanf(""); c_out = __builtin_nanf(""); }   // |y| >= 120, inf, NaN: outside the restated domain
}
// __powf (e_powf.c) for x >= +0 and finite y (CosinePower's r0^(1/(e+1)): r0 in [0, 1), y > 0); negative or
// non-finite arguments follow glibc's special cases.
__device__ __forceinline__ int GlibcCheckInt(uint32_t iy) {
  const int e = static_cast<int>(iy >> 23 & 0xffu);
  if (e < 0x7f) return 0;
  if (e > 0x7f + 23) return 2;
  if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
  if (iy & (1u << (0x7f + 23 - e))) return 1;
  return 2;
}
// A double constant that is the ADDEND of an fma must sit in a VGPR pair (one SGPR operand per VOP3 instruction).  Left to
// itself the compiler materialises such constants once, outside the persistent loop, and -- under the kernels' register
// caps -- spills them to scratch, reloading them on every evaluation (pt_megakernel: three 8-byte scratch loads per Pow).
// Passing the constant through an empty asm next to its use makes it two v_mov instead.
#define AMBER_NEAR_CONSTANT(bits64_) NearConstantBits<static_cast<uint32_t>((bits64_) & 0xffffffffull), static_cast<uint32_t>((bits64_) >> 32)>()
template <uint32_t kLo, uint32_t kHi>
__device__ __forceinline__ double NearConstantBits() {
  uint32_t lo, hi;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(lo), "=v"(hi) : "n"(kLo), "n"(kHi));
  return __hiloint2double(static_cast<int>(hi), static_cast<int>(lo));
}
__device__ __forceinline__ float Pow(float x, float y) {
  uint32_t sign_bias = 0u;
  uint32_t ix = __float_as_uint(x);
  const uint32_t iy = __float_as_uint(y);
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || 2u * iy - 1u >= 2u * 0x7f800000u - 1u) {
    if (2u * iy - 1u >= 2u * 0x7f800000u - 1u) {                               // y is 0, inf or NaN
      if (2u * iy == 0u) return 1.0f;
      if (ix == 0x3f800000u) return 1.0f;
      if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
      if (2u * ix == 2u * 0x3f800000u) return 1.0f;
      if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
      return y * y;
    }
    if (2u * ix - 1u >= 2u * 0x7f800000u - 1u) {                               // x is 0, inf or NaN
      float x2 = x * x;
      if ((ix & 0x80000000u) && GlibcCheckInt(iy) == 1) x2 = -x2;
      return (iy & 0x80000000u) ? 1.0f / x2 : x2;
    }
    if (ix & 0x80000000u) {                                                     // finite x < 0
      const int yint = GlibcCheckInt(iy);
      if (yint == 0) return __builtin_nanf("");
      if (yint == 1) sign_bias = 1u << 16;
      ix &= 0x7fffffffu;
    }
    if (ix < 0x00800000u) { ix = __float_as_uint(__uint_as_float(ix) * 0x1p23f); ix &= 0x7fffffffu; ix -= 23u << 23; }   // subnormal x
  }
  // log2_inline: x = 2^k z, z in [OFF, 2 OFF), one of 16 subintervals with centre c; log2(x) = log1p(z/c - 1)/ln2 + log2(c) + k
  const uint32_t tmp = ix - 0x3f330000u;
  const uint32_t i = (tmp >> 19) & 15u;
  const uint32_t top = tmp & 0xff800000u;
  const double z = static_cast<double>(__uint_as_float(ix - top));
  const int k = static_cast<int32_t>(top) >> 23;
  const double2 t_log = *reinterpret_cast<const double2*>(kGlibcLog2Tab[i]);
  const double r = __builtin_fma(z, t_log.x, -1.0);
  const double y0 = t_log.y + static_cast<double>(k);
  const double r2 = r * r;
  double yy = __builtin_fma(0x1.27616c9496e0bp-2, r, AMBER_NEAR_CONSTANT(0xbfd71969a075c67aull) /* -0x1.71969a075c67ap-2 */);
  const double pp = __builtin_fma(0x1.ec70a6ca7baddp-2, r, AMBER_NEAR_CONSTANT(0xbfe7154748bef6c8ull) /* -0x1.7154748bef6c8p-1 */);
  const double r4 = r2 * r2;
  double q = __builtin_fma(0x1.71547652ab82bp0, r, y0);
  q = __builtin_fma(pp, r2, q);
  yy = __builtin_fma(yy, r4, q);
  const double ylogx = static_cast<double>(y) * yy;
  if ((static_cast<uint64_t>(__double_as_longlong(ylogx)) >> 47 & 0xffffull) >= (0x405f800000000000ull >> 47)) {   // |y log2 x| >= 126
    const float sgn = sign_bias ? -1.0f : 1.0f;
    if (ylogx > 0x1.fffffffd1d571p+6) return sgn * __builtin_inff();
    if (ylogx <= -150.0) return sgn * 0.0f;
    if (ylogx < -149.0) return sgn * 0x1p-149f;
  }
  // exp2_inline: x = k/32 + r, 2^x = 2^(k/32) * 2^r
  const double shift = 0x1.8p+52 / 32;
  double kd = ylogx + shift;
  const uint64_t ki = static_cast<uint64_t>(__double_as_longlong(kd));
  kd -= shift;
  const double rr = ylogx - kd;
  uint64_t t = kGlibcExp2Tab[ki & 31ull];
  t += (ki + sign_bias) << 47;
  const double sc = __longlong_as_double(static_cast<long long>(t));
  const double zz = __builtin_fma(0x1.c6af84b912394p-5, rr, AMBER_NEAR_CONSTANT(0x3fcebfce50fac4f3ull) /* 0x1.ebfce50fac4f3p-3 */);
  const double rr2 = rr * rr;
  double e = __builtin_fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
  e = __builtin_fma(zz, rr2, e);
  return static_cast<float>(e * sc);
}
#endif
// std::pow(float, int) promotes to double (C++11): glibc's double pow.  x*x is exact (24 + 24 bits), so (x*x)^2 is the
// correctly rounded x^4; x^5 is formed from the exact (hi, lo) pair of x^4 and rounds once too (portable build: two
// roundings, as round 1).  glibc's pow is not correctly rounded: it differs from these in the last bit of the DOUBLE for
// 1e-3 of the arguments, which never survived the conversion to binary32 in 2e8 trials.
__device__ __forceinline__ double Pow4(float x) { const double d = x; const double d2 = d * d; return d2 * d2; }
#ifdef AMBER_BUILD_PORTABLE_MATH
__device__ __forceinline__ double Pow5(float x) { const double d = x; const double d2 = d * d; return (d2 * d2) * d; }
#else
__device__ __forceinline__ double Pow5(float x) {
  const double d = x, d2 = d * d;
  const double h = d2 * d2, l = __builtin_fma(d2, d2, -h);
  const double p = h * d, pl = __builtin_fma(h, d, -p);
  return p + __builtin_fma(l, d, pl);
}
#endif

}  // namespace amber_dev
