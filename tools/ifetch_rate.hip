// ifetch_rate.hip -- is a long straight-line VALU stream limited by instruction fetch rather than by the VALU?
// Each kernel runs ITER trips over a body of REPT independent v_fma (VOP3, 8-byte) or v_fmac (VOP2, 4-byte)
// instructions; a short body stays in the wave's instruction buffer, a long one (>= 16 KB) has to stream from the
// instruction cache.  Prints SIMD cycles per wave-instruction for each (encoding, body length).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ifetch_rate tools/ifetch_rate.hip && /tmp/ifetch_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY_VOP3(n) asm volatile(".rept " #n "\n v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n .endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y))
#define BODY_VOP2(n) asm volatile(".rept " #n "\n v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n .endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y))

#define KERNEL(name, BODY, n)                                                              \
  __global__ void __launch_bounds__(256) name(float* out, const float* in, int iters) {    \
    float a = in[0] + threadIdx.x, b = in[1], c = in[2], d = in[3], x = in[4], y = in[5];  \
    for (int i = 0; i < iters; i++) { BODY(n); }                                           \
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;                                   \
  }
KERNEL(vop3_16, BODY_VOP3, 4)        // 16 instructions, 128 B
KERNEL(vop3_1k, BODY_VOP3, 256)      // 1024 instructions, 8 KB
KERNEL(vop3_4k, BODY_VOP3, 1024)     // 4096 instructions, 32 KB
KERNEL(vop2_16, BODY_VOP2, 4)
KERNEL(vop2_1k, BODY_VOP2, 256)      // 4 KB
KERNEL(vop2_4k, BODY_VOP2, 1024)     // 16 KB

template <typename K>
static double Run(K kernel, int n_instr, int waves_per_simd, float* d_out, const float* d_in, int cus, double ghz) {
  const int total = 1 << 22;                 // wave-instructions per wave
  const int iters = total / n_instr;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kernel<<<cus * waves_per_simd, 256>>>(d_out, d_in, 4);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kernel<<<cus * waves_per_simd, 256>>>(d_out, d_in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * ghz * 1e9 / (double(waves_per_simd) * iters * n_instr);
}

int main() {
  hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
  const int cus = prop.multiProcessorCount; const double ghz = prop.clockRate * 1e-6;
  float *d_out, *d_in; hipMalloc(&d_out, 4 * cus * 8 * 256); hipMalloc(&d_in, 64);
  const float h[8] = {1.f, 2.f, 3.f, 4.f, 0.999f, 0.001f, 0, 0}; hipMemcpy(d_in, h, 32, hipMemcpyHostToDevice);
  std::printf("%s, %d CUs, %.2f GHz nominal: SIMD cycles per wave-instruction\n", prop.gcnArchName, cus, ghz);
  for (int w : {1, 2, 4, 6, 8}) {
    std::printf("%d waves/SIMD:  VOP3 body 16 / 1024 / 4096 instr: %.2f %.2f %.2f   VOP2: %.2f %.2f %.2f\n", w,
                Run(vop3_16, 16, w, d_out, d_in, cus, ghz), Run(vop3_1k, 1024, w, d_out, d_in, cus, ghz), Run(vop3_4k, 4096, w, d_out, d_in, cus, ghz),
                Run(vop2_16, 16, w, d_out, d_in, cus, ghz), Run(vop2_1k, 1024, w, d_out, d_in, cus, ghz), Run(vop2_4k, 4096, w, d_out, d_in, cus, ghz));
  }
  return 0;
}
