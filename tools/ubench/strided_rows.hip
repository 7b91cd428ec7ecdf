// Micro-benchmark: what the row kernels' access pattern to the tile-major operand T costs by itself.
// A workgroup owns C rows; lane t reads element (row, col = t + k*THREADS) -- CT consecutive lanes share one
// CT*sizeof(T)-byte piece of a tile, consecutive pieces of a row lie N*CT*sizeof(T) bytes apart (tile-major) --
// adds up and writes one value.  Layouts: rows of panels of R rows each ([panel][tile][row in panel][CT]);
// R = N is the tile-major layout of the engine, R = 1 is row-major.
//   hipcc --offload-arch=gfx950 -O3 strided_rows.hip -o strided_rows && ./strided_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <typename T>
__global__ __launch_bounds__(256) void k_read(const T* __restrict__ A, double* __restrict__ out, int N, int CT, int R, int C, int write,
                                              T* __restrict__ B, int PAD, int XCD) {
  // workgroups b, b+8, b+16, ... run on one XCD (one L2): give them the row groups that share 128-byte lines
  // (LINE_ROWS rows of CT*sizeof(T)-byte pieces), like row_of_block() of the engine
  const int LINE_ROWS = 128 / (CT * (int)sizeof(T)) > C ? 128 / (CT * (int)sizeof(T)) : C;
  const int Q = LINE_ROWS / C;
  const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
  const int row0 = XCD ? (xcd + 8 * (j / Q)) * LINE_ROWS + (j % Q) * C : b * C;
  double s = 0.0;
  if (XCD == 2) {
    // two rows per workgroup, every wavefront takes 32 columns of BOTH rows: its loads touch 64-byte chunks
    // (the rows' pieces of a tile are neighbours) instead of 32-byte pieces
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = row0 + (lane >> 5);
    const size_t pstride = (size_t)R * CT + PAD;
    const size_t pbase = (size_t)(r / R) * (pstride * (N / CT)) + (size_t)(r % R) * CT;
    for (int col = w * 32 + (lane & 31); col < N; col += 128) {
      const size_t off = pbase + (size_t)(col / CT) * pstride + (col % CT);
      const T v = A[off];
      if (write) B[off] = v * T(1.0001);
      else s += (double)v;
    }
    if (!write) out[blockIdx.x * 256 + threadIdx.x] = s;
    return;
  }
  for (int r = row0; r < row0 + C; ++r) {
    const size_t pstride = (size_t)R * CT + PAD;  // elements between consecutive tiles of a panel
    const size_t pbase = (size_t)(r / R) * (pstride * (N / CT)) + (size_t)(r % R) * CT;
    for (int col = threadIdx.x; col < N; col += 256) {
      const size_t off = pbase + (size_t)(col / CT) * pstride + (col % CT);
      const T v = A[off];
      if (write) B[off] = v * T(1.0001);
      else s += (double)v;
    }
  }
  if (!write) out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename T>
void run(int N, int CT, int R, int write, int PAD = 0, int XCD = 1) {
  const size_t n = (size_t)N * N;
  const size_t na = (size_t)(N / R) * ((size_t)R * CT + PAD) * (N / CT);
  T *A, *B; double* out;
  hipMalloc(&A, na * sizeof(T)); hipMalloc(&B, na * sizeof(T)); hipMalloc(&out, sizeof(double) * 256 * (N / 2));
  hipMemset(A, 0, na * sizeof(T));
  hipMemset(B, 0, na * sizeof(T));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int C = 2, reps = 20;
  for (int i = 0; i < 3; ++i) k_read<T><<<N / C, 256>>>(A, out, N, CT, R, C, write, B, PAD, XCD);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) k_read<T><<<N / C, 256>>>(A, out, N, CT, R, C, write, B, PAD, XCD);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double gb = (double)n * sizeof(T) * (write ? 2 : 1) / 1e9;
  printf("N=%d %s CT=%d R=%-5d pad=%-5d xcd-aware=%d %s: %.1f us  %.0f GB/s\n", N, sizeof(T) == 8 ? "f64" : "f32", CT, R, PAD, XCD, write ? "copy" : "read", ms * 1e3,
         gb / (ms * 1e-3));
  hipFree(A); hipFree(B); hipFree(out);
}
int main() {
  for (int write = 0; write < 2; ++write) {
    for (int x = 1; x < 3; ++x) {
      run<double>(4096, 4, 4096, write, 0, x);
      run<double>(8192, 4, 8192, write, 0, x);
      run<double>(8192, 8, 8192, write, 0, x);
      run<float>(4096, 8, 4096, write, 0, x);
      run<float>(8192, 8, 8192, write, 0, x);
    }
  }
  return 0;
}
