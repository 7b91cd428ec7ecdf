// Micro-benchmark: fp64 FMA dependent-chain vs independent-chain issue rate on gfx950,
// at 1, 2 and 4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 fp64_latency.hip -o fp64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k(double* out, int iters, double a, double b) {
  double x[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3 + c;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
void run(int threads_per_block, const char* tag) {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 1024);
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CHAINS><<<256, threads_per_block>>>(d, 16, 0.999, 1e-3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<CHAINS><<<256, threads_per_block>>>(d, iters, 0.999, 1e-3);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * 16 * CHAINS;
  const double waves_per_simd = threads_per_block / 64 / 4.0;
  // cycles per instruction per wave at 2.4 GHz nominal
  printf("%s chains=%d waves/SIMD=%.2g : %.3f ms, %.2f ns per fma per wave, ~%.1f cycles@2.4GHz, SIMD issue interval %.1f cycles\n", tag,
         CHAINS, waves_per_simd, ms, ms * 1e6 / instr_per_wave, ms * 1e6 / instr_per_wave * 2.4,
         ms * 1e6 / instr_per_wave * 2.4 / (waves_per_simd < 1 ? 1 : waves_per_simd));
  hipFree(d);
}
int main() {
  for (int tpb : {256, 512, 1024}) {
    run<1>(tpb, "dep ");
    run<2>(tpb, "ilp2");
    run<4>(tpb, "ilp4");
    run<8>(tpb, "ilp8");
  }
  return 0;
}
