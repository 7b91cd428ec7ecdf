// Micro-benchmark: the memory side of k_col<MODE_STEP> alone.  A workgroup (256 threads) owns COLS columns of a
// CT-column tile of the tile-major operand T ([tile][row][CT]): it reads its COLS*8 bytes of every row piece,
// reads and rewrites the same number of hat_U elements (contiguous per column) and writes T back in place.
// The Q = CT/COLS workgroups of a tile get block numbers b, b+8, ... (one XCD), as in the engine.
//   hipcc --offload-arch=gfx950 -O3 -w col_tiles.hip -o col_tiles && ./col_tiles
#include <hip/hip_runtime.h>
#include <cstdio>
template <int COLS>
__global__ __launch_bounds__(256) void k(double* __restrict__ T, double* __restrict__ H, int N, int CT, int xcd_aware) {
  const int Q = CT / COLS;
  const int b = blockIdx.x;
  int ct, hh;
  if (xcd_aware && Q > 1) { const int x = b & 7, j = b >> 3; ct = x + 8 * (j / Q); hh = j % Q; }
  else { ct = b / Q; hh = b % Q; }
  double* tile = T + (size_t)ct * N * CT + hh * COLS;
  double* hcol = H + ((size_t)ct * CT + hh * COLS) * N;
  double v[COLS];
  // stage-in: row pieces
  for (int r = threadIdx.x; r < N; r += 256) {
#pragma unroll
    for (int c = 0; c < COLS; c += 2) {
      const double2 x = *reinterpret_cast<const double2*>(tile + (size_t)r * CT + c);
      v[c] = x.x; v[c + 1] = x.y;
    }
    // spectral stage: hat_U read-modify-write, contiguous per column
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
      const double h = hcol[(size_t)c * N + r];
      const double nh = h * 0.5 + v[c];
      hcol[(size_t)c * N + r] = nh;
      v[c] = nh * 0.25;
    }
#pragma unroll
    for (int c = 0; c < COLS; c += 2) *reinterpret_cast<double2*>(tile + (size_t)r * CT + c) = make_double2(v[c], v[c + 1]);
  }
}
template <int COLS>
void run(int N, int CT, int xcd) {
  const size_t n = (size_t)N * N;
  double *T, *H;
  hipMalloc(&T, n * 8); hipMalloc(&H, n * 8);
  hipMemset(T, 0, n * 8); hipMemset(H, 0, n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = N / COLS, reps = 20;
  for (int i = 0; i < 3; ++i) k<COLS><<<grid, 256>>>(T, H, N, CT, xcd);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) k<COLS><<<grid, 256>>>(T, H, N, CT, xcd);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("N=%d CT=%d cols/WG=%d xcd-aware=%d: %.1f us  %.0f GB/s (4 transfers)\n", N, CT, COLS, xcd, ms * 1e3, 4.0 * n * 8 / 1e9 / (ms * 1e-3));
  hipFree(T); hipFree(H);
}
int main() {
  for (int N : {4096, 8192}) {
    run<2>(N, 4, 1); run<2>(N, 4, 0); run<4>(N, 4, 1); run<2>(N, 2, 1); run<2>(N, 8, 1); run<4>(N, 8, 1); run<8>(N, 8, 1);
  }
  return 0;
}
