// CPU check of bvh_build.h's binary16 plane encoding (tests/test_bvh_planes_cpu.py compiles and runs it with hipcc; no GPU needed):
// for random boxes at many scales and offsets the stored min plane is <= the box's min, the max plane >= its max, both are the
// TIGHTEST representable values with that property (up to the guard), and no stored value is a binary16 denormal, inf or NaN.
#include <cstdio>
#include <random>
#include "../../amber_amd/csrc/hip/bvh_build.h"

int main() {
  using namespace amber_bvh;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  long bad = 0, n = 0, loose = 0;
  for (int trial = 0; trial < 400; trial++) {
    const double scale = std::pow(10.0, -4.0 + 10.0 * U(rng));                 // half extents from 1e-4 to 1e6
    const double centre = (U(rng) - 0.5) * std::pow(10.0, -2.0 + 9.0 * U(rng)) * (trial % 5 == 0 ? 0.0 : 1.0);
    const float gmid = static_cast<float>(centre), half = static_cast<float>(scale);
    for (int k = 0; k < 500; k++) {
      double a = centre + (2.0 * U(rng) - 1.0) * scale, b = centre + (2.0 * U(rng) - 1.0) * scale;
      if (k % 7 == 0) a = centre + (U(rng) - 0.5) * scale * 1e-5;               // planes next to the centre: the denormal range of u
      if (k % 11 == 0) b = a;                                                  // a flat box
      if (a > b) std::swap(a, b);
      const float mn = static_cast<float>(a), mx = static_cast<float>(b);
      if (!(mn <= mx)) continue;
      const uint32_t w = PlaneWord(mn, mx, gmid, half);
      const uint16_t lo = w & 0xffffu, hi = w >> 16;
      for (uint16_t h : {lo, hi}) {
        const int e = (h >> 10) & 31, m = h & 1023;
        if (e == 31 || (e == 0 && m != 0)) { bad++; std::printf("non-normal value %04x\n", h); }
      }
      const long double pl = (long double)gmid + (long double)F16Value(lo) * half, ph = (long double)gmid + (long double)F16Value(hi) * half;
      if (!(pl <= (long double)mn) || !(ph >= (long double)mx)) { bad++; if (bad < 10) std::printf("not conservative: [%g, %g] stored [%Lg, %Lg]\n", mn, mx, pl, ph); }
#if AMBER_BVH_F16
      // tightness: one representable step inwards must violate the bound (or be within the guard of it)
      const long double guard = (fabsl((long double)gmid) + fabsl((long double)half) * 2.0L) * 0x1p-49L;
      const long double pl2 = (long double)gmid + (long double)F16Value(F16Step(lo, true)) * half, ph2 = (long double)gmid + (long double)F16Value(F16Step(hi, false)) * half;
      if (pl2 <= (long double)mn - guard && F16Step(lo, true) != lo) loose++;
      if (ph2 >= (long double)mx + guard && F16Step(hi, false) != hi) loose++;
#endif
      n++;
    }
  }
  // a scene with no extent on an axis, away from the origin: the grid of that axis must keep a usable step, the planes
  // of the flat box must be found in a few steps, stay conservative and inside the promised reach |value| <= 1 + one binary16 step
  for (double coord : {1234.5, -0.37, 6.0e5, 0.0, 1e-20}) {
    float mid, half;
    F16AxisGrid(coord, coord, mid, half);
    if (!(half > 0.0f) || (coord != 0.0 && !(half >= std::fabs(mid) * 9e-7f))) { bad++; std::printf("degenerate axis at %g: half %g\n", coord, half); }
    const float c = static_cast<float>(coord);
    const uint32_t w = PlaneWord(c, c, mid, half);
    const double vlo = F16Value(w & 0xffffu), vhi = F16Value(w >> 16);
    if (!(double(mid) + vlo * half <= c) || !(double(mid) + vhi * half >= c) || std::fabs(vlo) > 1.001 || std::fabs(vhi) > 1.001) {
      bad++; std::printf("degenerate axis at %g: planes %g %g (values %g %g)\n", coord, double(mid) + vlo * half, double(mid) + vhi * half, vlo, vhi);
    }
    n++;
  }
  std::printf("%ld boxes, %ld violations, %ld planes not tight\n", n, bad, loose);
  return (bad == 0 && loose == 0 && n > 100000) ? 0 : 1;
}
