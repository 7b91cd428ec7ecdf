// Micro-benchmark: the memory side of k_col<MODE_STEP> alone.  A workgroup (256 threads) owns COLS columns of a
// CT-column tile of the tile-major operand T ([tile][row][CT]): it reads its COLS*8 bytes of every row piece,
// reads and rewrites the same number of hat_U elements (contiguous per column) and writes T back in place.
// The Q = CT/COLS workgroups of a tile get block numbers b, b+8, ... (one XCD), as in the engine.
//   hipcc --offload-arch=gfx950 -O3 -w col_tiles.hip -o col_tiles && ./col_tiles
#include <hip/hip_runtime.h>
#include <cstdio>
template <int COLS, typename F, int THREADS>
__global__ __launch_bounds__(THREADS) void k(F* __restrict__ T, F* __restrict__ H, int N, int CT, int xcd_aware) {
  const int Q = CT / COLS;
  const int b = blockIdx.x;
  int ct, hh;
  if (xcd_aware && Q > 1) { const int x = b & 7, j = b >> 3; ct = x + 8 * (j / Q); hh = j % Q; }
  else { ct = b / Q; hh = b % Q; }
  F* tile = T + (size_t)ct * N * CT + hh * COLS;
  F* hcol = H + ((size_t)ct * CT + hh * COLS) * N;
  F v[COLS];
  // stage-in: row pieces
  for (int r = threadIdx.x; r < N; r += THREADS) {
#pragma unroll
    for (int c = 0; c < COLS; c += 2) {
      if constexpr (sizeof(F) == 8) { const double2 x = *reinterpret_cast<const double2*>(tile + (size_t)r * CT + c); v[c] = x.x; v[c + 1] = x.y; }
      else { const float2 x = *reinterpret_cast<const float2*>(tile + (size_t)r * CT + c); v[c] = x.x; v[c + 1] = x.y; }
    }
    // spectral stage: hat_U read-modify-write, contiguous per column
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
      const F h = hcol[(size_t)c * N + r];
      const F nh = h * F(0.5) + v[c];
      hcol[(size_t)c * N + r] = nh;
      v[c] = nh * F(0.25);
    }
#pragma unroll
    for (int c = 0; c < COLS; c += 2) {
      if constexpr (sizeof(F) == 8) *reinterpret_cast<double2*>(tile + (size_t)r * CT + c) = make_double2(v[c], v[c + 1]);
      else *reinterpret_cast<float2*>(tile + (size_t)r * CT + c) = make_float2(v[c], v[c + 1]);
    }
  }
}
template <int COLS, typename F = double, int THREADS = 256>
void run(int N, int CT, int xcd) {
  const size_t n = (size_t)N * N;
  F *T, *H;
  hipMalloc(&T, n * sizeof(F)); hipMalloc(&H, n * sizeof(F));
  hipMemset(T, 0, n * sizeof(F)); hipMemset(H, 0, n * sizeof(F));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = N / COLS, reps = 20;
  for (int i = 0; i < 3; ++i) k<COLS, F, THREADS><<<grid, THREADS>>>(T, H, N, CT, xcd);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) k<COLS, F, THREADS><<<grid, THREADS>>>(T, H, N, CT, xcd);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("N=%d %s CT=%d cols/WG=%d threads=%d xcd-aware=%d: %.1f us  %.0f GB/s (4 transfers)\n", N, sizeof(F) == 8 ? "f64" : "f32", CT, COLS, THREADS, xcd, ms * 1e3, 4.0 * n * sizeof(F) / 1e9 / (ms * 1e-3));
  hipFree(T); hipFree(H);
}
int main() {
  run<2>(4096, 4, 1); run<2, double, 512>(8192, 4, 1); run<2, double, 512>(8192, 8, 1);
  run<2, float, 256>(4096, 8, 1); run<2, float, 512>(8192, 8, 1); run<4, float, 512>(8192, 8, 1); run<8, float, 512>(8192, 8, 1);
  run<2, float, 512>(8192, 4, 1); run<4, float, 512>(8192, 4, 1); run<2, float, 512>(8192, 16, 1);
  return 0;
}
