// Micro-benchmark: SIMD cycles per wave-instruction of the fp32 / packed-fp32 / fp64 arithmetic the transform core is
// made of, at 1, 2 and 4 waves per SIMD (8 independent chains per wave: throughput, not latency).
//   hipcc --offload-arch=gfx950 -O3 pk_rates.hip -o pk_rates && ./pk_rates
// Question it answers: does v_pk_fma_f32 (128 FMAs per wave-instruction) cost the SIMD-32 pipe the cycles of ONE
// v_fma_f32 or of two?  (157.3 TF fp32 vector peak = 256 CUs x 4 SIMDs x 32 lanes x 2 x 2.4 GHz is reached WITHOUT packing.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
enum { OP_FMA32, OP_PKFMA32, OP_ADD32, OP_PKADD32, OP_FMA64, OP_ADD64, OP_PKMUL32 };
template <int OP>
__global__ void k(float* out, int iters, float a, float b) {
  v2f x[8];
  double d[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { x[c] = v2f{1.0f + threadIdx.x * 1e-3f + c, 0.5f + c}; d[c] = 1.0 + c + threadIdx.x * 1e-3; }
  const v2f av = {a, a}, bv = {b, b};
  const double ad = a, bd = b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if constexpr (OP == OP_FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c].x) : "v"(a), "v"(b));
        else if constexpr (OP == OP_PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(av), "v"(bv));
        else if constexpr (OP == OP_ADD32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c].x) : "v"(b));
        else if constexpr (OP == OP_PKADD32) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(bv));
        else if constexpr (OP == OP_PKMUL32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(av));
        else if constexpr (OP == OP_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"(ad), "v"(bd));
        else asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(bd));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += x[c].x + x[c].y + (float)d[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char* tag) {
  float* d; hipMalloc(&d, sizeof(float) * 256 * 1024 * 2);
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-14s", tag);
  for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD: blocks of 256 threads, wps blocks per CU
    const int blocks = 256 * wps;
    k<OP><<<blocks, 256>>>(d, 16, 0.999f, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 0.999f, 1e-3f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD = wps * iters * 64; time in ns per wave-instruction per SIMD
    const double ns = ms * 1e6 / ((double)wps * iters * 64);
    printf("  wps=%d: %6.3f ns/inst (%.2f cyc @2.4GHz)", wps, ns, ns * 2.4);
  }
  printf("\n");
  hipFree(d);
}
int main() {
  run<OP_FMA32>("v_fma_f32");
  run<OP_ADD32>("v_add_f32");
  run<OP_PKFMA32>("v_pk_fma_f32");
  run<OP_PKADD32>("v_pk_add_f32");
  run<OP_PKMUL32>("v_pk_mul_f32");
  run<OP_FMA64>("v_fma_f64");
  run<OP_ADD64>("v_add_f64");
  return 0;
}
