// Micro-benchmark: issue cost of the fp64 / conversion instructions the pointwise code uses, relative to
// v_fma_f64, at 4 waves per SIMD with 4 independent chains per wave (throughput regime).
//   hipcc --offload-arch=gfx950 -O3 op_rates.hip -o op_rates
#include <hip/hip_runtime.h>
#include <cstdio>
enum { OP_FMA, OP_RCP, OP_FREXP_MANT, OP_FREXP_EXP, OP_CVT_I2D, OP_RNDNE, OP_CNDMASK, OP_MIN, OP_SQRT, OP_RSQ, OP_LDEXP, OP_AND_OR };
template <int OP>
__device__ __forceinline__ double op(double x, double a, double b) {
  if constexpr (OP == OP_FMA) return __builtin_fma(x, a, b);
  else if constexpr (OP == OP_RCP) return __builtin_amdgcn_rcp(x) + b;
  else if constexpr (OP == OP_FREXP_MANT) return __builtin_amdgcn_frexp_mant(x) + a;
  else if constexpr (OP == OP_FREXP_EXP) return (double)__builtin_amdgcn_frexp_exp(x) + a;   // + cvt + add
  else if constexpr (OP == OP_CVT_I2D) return (double)(__double2loint(x) & 1023) + a;          // and + cvt + add
  else if constexpr (OP == OP_RNDNE) return __builtin_rint(x) * a;
  else if constexpr (OP == OP_CNDMASK) return (x > a) ? x * b : x + b;
  else if constexpr (OP == OP_MIN) return fmin(x, a) + b;
  else if constexpr (OP == OP_SQRT) return __builtin_amdgcn_sqrt(x) + a;
  else if constexpr (OP == OP_RSQ) return __builtin_amdgcn_rsq(x) + a;
  else if constexpr (OP == OP_LDEXP) return __builtin_ldexp(x, 1) * a;
  else return __hiloint2double((__double2hiint(x) & 0x000FFFFF) | 0x3FE00000, __double2loint(x)) + a;
}
template <int OP>
__global__ void k(double* out, int iters, double a, double b) {
  double x[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) x[c] = 1.0 + threadIdx.x * 1e-3 + c;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int c = 0; c < 4; ++c) x[c] = op<OP>(x[c], a, b);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}
template <int OP>
double run(const char* tag) {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 1024);
  const int iters = 2048;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<256, 1024>>>(d, 16, 0.999, 1e-3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<256, 1024>>>(d, iters, 0.999, 1e-3);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e6 / ((double)iters * 8 * 4) * 2.4 / 4.0;  // cycles per statement per wave slot at 2.4 GHz, 4 waves/SIMD
  printf("%-12s %.3f ms  %.1f SIMD cycles per statement\n", tag, ms, per);
  hipFree(d);
  return per;
}
int main() {
  run<OP_FMA>("fma");
  run<OP_RCP>("rcp+add");
  run<OP_FREXP_MANT>("frexp_m+add");
  run<OP_FREXP_EXP>("frexp_e+cvt+add");
  run<OP_CVT_I2D>("and+cvt+add");
  run<OP_RNDNE>("rndne+mul");
  run<OP_CNDMASK>("cmp+mul+add+cnd");
  run<OP_MIN>("min+add");
  run<OP_SQRT>("sqrt+add");
  run<OP_RSQ>("rsq+add");
  run<OP_LDEXP>("ldexp+mul");
  run<OP_AND_OR>("and_or+add");
  return 0;
}
