// valu_rates.hip -- issue cost (SIMD cycles per wave64 instruction) of the VALU operations the path tracer leans on,
// measured on the device: every wave runs ITER iterations of UNROLL independent dependency chains of one operation,
// the grid fills every SIMD with 8 waves, cost = elapsed cycles * SIMDs / wave-instructions.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define ITER 4096
#define CHAINS 8

enum Op { FMA32, MUL32, ADD32, RCP32, SQRT32, RSQ32, DIV32_IEEE, SQRT32_IEEE, FMA64, ADD64F, MUL64F, DIV64_IEEE, SHL64, XORSHIFT64, XORSHIFT64_32,
          MULLO32, MULHI32, MAD64_32, CVT_F32_U32, MIN3, CNDMASK, LDS_READ, PK_FMA32, EXP32, LOG32, SIN32, N_OPS };
static const char* kNames[N_OPS] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_rcp_f32", "v_sqrt_f32 (approx)", "v_rsq_f32", "a / b (IEEE f32 sequence)",
                                    "sqrtf (IEEE f32 sequence)", "v_fma_f64", "v_add_f64", "v_mul_f64", "a / b (IEEE f64 sequence)", "v_lshlrev_b64",
                                    "xorshift64 step (u64 code)", "xorshift64 step (u32 halves, alignbit)", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32",
                                    "v_cvt_f32_u32", "v_min3_f32", "v_cndmask_b32 (+v_cmp)", "ds_read_b32 (broadcast)", "v_pk_fma_f32 (2 flops/lane)",
                                    "v_exp_f32", "v_log_f32", "v_sin_f32"};

template <int kOp>
__global__ void __launch_bounds__(256) rate_kernel(float* out, const float* in, int iters) {
  __shared__ float lds[64];
  if (threadIdx.x < 64) lds[threadIdx.x] = in[threadIdx.x];
  __syncthreads();
  float f[CHAINS]; double d[CHAINS]; uint64_t u[CHAINS]; uint32_t w[CHAINS];
  const float a = in[0], b = in[1];
  const double da = in[0], db = in[1];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) { f[c] = in[2 + c] + threadIdx.x; d[c] = f[c]; u[c] = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + c); w[c] = static_cast<uint32_t>(u[c]); }
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
      if (kOp == FMA32) f[c] = __builtin_fmaf(f[c], a, b);
      else if (kOp == MUL32) f[c] = f[c] * a;
      else if (kOp == ADD32) f[c] = f[c] + a;
      else if (kOp == RCP32) f[c] = __builtin_amdgcn_rcpf(f[c]);
      else if (kOp == SQRT32) f[c] = __builtin_amdgcn_sqrtf(f[c]);
      else if (kOp == RSQ32) f[c] = __builtin_amdgcn_rsqf(f[c]);
      else if (kOp == DIV32_IEEE) f[c] = a / f[c];
      else if (kOp == SQRT32_IEEE) f[c] = __builtin_sqrtf(f[c]);
      else if (kOp == FMA64) d[c] = __builtin_fma(d[c], da, db);
      else if (kOp == ADD64F) d[c] = d[c] + da;
      else if (kOp == MUL64F) d[c] = d[c] * da;
      else if (kOp == DIV64_IEEE) d[c] = da / d[c];
      else if (kOp == SHL64) u[c] = (u[c] << 13) + 1u;     // the add keeps the chain from folding; counted as 1 op below
      else if (kOp == XORSHIFT64) { uint64_t x = u[c]; x ^= x << 13; x ^= x >> 7; x ^= x << 17; u[c] = x; }
      else if (kOp == XORSHIFT64_32) {
        uint32_t lo = static_cast<uint32_t>(u[c]), hi = static_cast<uint32_t>(u[c] >> 32);
        hi ^= __builtin_amdgcn_alignbit(hi, lo, 19); lo ^= lo << 13;          // x ^= x << 13
        lo ^= __builtin_amdgcn_alignbit(hi, lo, 7); hi ^= hi >> 7;            // x ^= x >> 7
        hi ^= __builtin_amdgcn_alignbit(hi, lo, 15); lo ^= lo << 17;          // x ^= x << 17
        u[c] = (static_cast<uint64_t>(hi) << 32) | lo;
      }
      else if (kOp == MULLO32) w[c] = w[c] * 2654435761u;
      else if (kOp == MULHI32) w[c] = __umulhi(w[c], 2654435761u) + 3u;
      else if (kOp == MAD64_32) u[c] = static_cast<uint64_t>(static_cast<uint32_t>(u[c])) * 2654435761u + u[c];
      else if (kOp == CVT_F32_U32) { f[c] = static_cast<float>(w[c]); w[c] = __float_as_uint(f[c]); }
      else if (kOp == MIN3) f[c] = __builtin_fminf(__builtin_fminf(f[c], a), b);
      else if (kOp == CNDMASK) f[c] = f[c] > a ? b : f[c];
      else if (kOp == LDS_READ) f[c] = lds[__float_as_uint(f[c]) & 63u];
      else if (kOp == PK_FMA32) { f[c] = __builtin_fmaf(f[c], a, b); }   // paired below by the compiler when CHAINS is even
      else if (kOp == EXP32) f[c] = __builtin_amdgcn_exp2f(f[c]);
      else if (kOp == LOG32) f[c] = __builtin_amdgcn_logf(f[c]);
      else if (kOp == SIN32) f[c] = __builtin_amdgcn_sinf(f[c]);
    }
  }
  float acc = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc += f[c] + static_cast<float>(d[c]) + static_cast<float>(u[c] & 0xffff) + static_cast<float>(w[c] & 0xff);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int kOp>
static float Run(float* d_out, const float* d_in, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  rate_kernel<kOp><<<blocks, 256>>>(d_out, d_in, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  rate_kernel<kOp><<<blocks, 256>>>(d_out, d_in, ITER);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms;
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { std::fprintf(stderr, "no HIP device\n"); return 1; }
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  const int blocks = cus * 8;                       // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  float *d_out, *d_in;
  hipMalloc(&d_out, sizeof(float) * blocks * 256);
  hipMalloc(&d_in, sizeof(float) * 64);
  std::vector<float> in(64);
  for (int i = 0; i < 64; i++) in[i] = 1.0f + 0.001f * i;
  hipMemcpy(d_in, in.data(), sizeof(float) * 64, hipMemcpyHostToDevice);
  float ms[N_OPS];
#define RUN(op) ms[op] = Run<op>(d_out, d_in, blocks);
  RUN(FMA32) RUN(MUL32) RUN(ADD32) RUN(RCP32) RUN(SQRT32) RUN(RSQ32) RUN(DIV32_IEEE) RUN(SQRT32_IEEE) RUN(FMA64) RUN(ADD64F) RUN(MUL64F) RUN(DIV64_IEEE)
  RUN(SHL64) RUN(XORSHIFT64) RUN(XORSHIFT64_32) RUN(MULLO32) RUN(MULHI32) RUN(MAD64_32) RUN(CVT_F32_U32) RUN(MIN3) RUN(CNDMASK) RUN(LDS_READ) RUN(PK_FMA32)
  RUN(EXP32) RUN(LOG32) RUN(SIN32)
  std::printf("%s: %d CUs, %.2f GHz (nominal); %d waves/SIMD, %d chains x %d iterations per wave\n", prop.gcnArchName, cus, ghz, 8, CHAINS, ITER);
  std::printf("%-44s %10s %26s\n", "operation (one per chain step)", "ms", "SIMD cycles per wave-step");
  for (int op = 0; op < N_OPS; op++) {
    const double steps_per_simd = 8.0 * CHAINS * ITER;                       // wave-steps issued on one SIMD
    const double cycles = ms[op] * 1e-3 * ghz * 1e9;
    std::printf("%-44s %10.3f %26.2f\n", kNames[op], ms[op], cycles / steps_per_simd);
  }
  return 0;
}
