// chs_direct.hip -- "direct" transform engine: the 2-D orthonormal DCT-II/III as
// two dense products with the cosine matrix D[k][n] = f_k*2*cos(pi k (2n+1)/(2N)).
// Works for every N (the reference accepts any N, e.g. `benchmark.py -N 100`,
// tests/run-tests.sh:15) and doubles as an on-device cross-check of the fast
// engine.  O(N^3) per transform: a correctness path, not the headline path.
//
// scipy.fftpack.dctn/idctn(norm='ortho') (chsimpy/solver.py:159,201,208):
//   forward  Y = D X D^T ,  inverse  X = D^T Y D .
#include <cmath>
#include "chs_common.h"

#define GT 64   // output tile edge
#define GK 16   // depth of one LDS stage
#define GTH 256

// C = op(A) * op(B), all N x N row-major.  op(A)(i,k) = TA ? A[k][i] : A[i][k].
template <typename T, bool TA, bool TB>
__global__ __launch_bounds__(GTH) void k_gemm(const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C,
                                              int N, const DevState* __restrict__ st, int ignore_halt) {
  if (!ignore_halt && st->halt) return;
  __shared__ T sA[GK][GT + 1];  // sA[k][i]
  __shared__ T sB[GK][GT + 1];  // sB[k][j]
  const int i0 = blockIdx.y * GT, j0 = blockIdx.x * GT;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16 threads, 4 x 4 outputs each
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;

  for (int k0 = 0; k0 < N; k0 += GK) {
    // stage A tile (GT x GK) and B tile (GK x GT)
    for (int e = threadIdx.x; e < GT * GK; e += GTH) {
      int i, k;
      if (TA) { i = e % GT; k = e / GT; } else { k = e % GK; i = e / GK; }
      const int gi = i0 + i, gk = k0 + k;
      T v = T(0);
      if (gi < N && gk < N) v = TA ? A[(size_t)gk * N + gi] : A[(size_t)gi * N + gk];
      sA[k][i] = v;
    }
    for (int e = threadIdx.x; e < GT * GK; e += GTH) {
      int j, k;
      if (TB) { k = e % GK; j = e / GK; } else { j = e % GT; k = e / GT; }
      const int gj = j0 + j, gk = k0 + k;
      T v = T(0);
      if (gj < N && gk < N) v = TB ? B[(size_t)gj * N + gk] : B[(size_t)gk * N + gj];
      sB[k][j] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GK; ++k) {
      double a[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { a[q] = (double)sA[k][ty * 4 + q]; b[q] = (double)sB[k][tx * 4 + q]; }
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = fma(a[p], b[q], acc[p][q]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int gi = i0 + ty * 4 + p;
    if (gi >= N) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int gj = j0 + tx * 4 + q;
      if (gj < N) C[(size_t)gi * N + gj] = (T)acc[p][q];
    }
  }
}

template <typename T>
static int gemm(Engine* E, const void* A, const void* B, void* C, bool TA, bool TB, int ignore_halt) {
  const int N = E->N;
  dim3 grid((N + GT - 1) / GT, (N + GT - 1) / GT);
  const T* a = (const T*)A; const T* b = (const T*)B; T* c = (T*)C;
  if (!TA && !TB) k_gemm<T, false, false><<<grid, GTH, 0, E->stream>>>(a, b, c, N, E->dState, ignore_halt);
  else if (!TA && TB) k_gemm<T, false, true><<<grid, GTH, 0, E->stream>>>(a, b, c, N, E->dState, ignore_halt);
  else if (TA && !TB) k_gemm<T, true, false><<<grid, GTH, 0, E->stream>>>(a, b, c, N, E->dState, ignore_halt);
  else k_gemm<T, true, true><<<grid, GTH, 0, E->stream>>>(a, b, c, N, E->dState, ignore_halt);
  CHS_HIP(hipGetLastError());
  return CHS_OK;
}

int chs_direct_init(Engine* E) {
  const int N = E->N;
  // D[k][n] in extended precision, rounded once to the device type.
  std::vector<long double> ang;  // not needed; computed inline
  const long double PI = 3.14159265358979323846264338327950288419716939937510L;
  const size_t bytes = (size_t)N * N * E->esz;
  std::vector<char> host(bytes);
  for (int k = 0; k < N; ++k) {
    const long double f = (k == 0) ? sqrtl(1.0L / (4.0L * N)) : sqrtl(1.0L / (2.0L * N));
    for (int n = 0; n < N; ++n) {
      // reduce the argument exactly: k(2n+1) mod 4N in integers
      const long long m = ((long long)k * (2LL * n + 1)) % (4LL * N);
      const long double v = 2.0L * f * cosl(PI * (long double)m / (2.0L * N));
      if (E->dtype == CHS_F64) ((double*)host.data())[(size_t)k * N + n] = (double)v;
      else ((float*)host.data())[(size_t)k * N + n] = (float)v;
    }
  }
  CHS_HIP(hipMalloc(&E->dD, bytes));
  CHS_HIP(hipMemcpy(E->dD, host.data(), bytes, hipMemcpyHostToDevice));
  return CHS_OK;
}

void chs_direct_free(Engine* E) {
  if (E->dD) hipFree(E->dD);
  E->dD = nullptr;
}

// out = D in D^T (forward) or D^T in D (inverse); tmp is an N x N scratch.
int chs_direct_dct2d(Engine* E, const void* in, void* out, void* tmp, bool inverse) {
  int rc;
  const int ih = 0;
  if (E->dtype == CHS_F64) {
    if (!inverse) {
      if ((rc = gemm<double>(E, in, E->dD, tmp, false, true, ih))) return rc;   // T = X D^T
      if ((rc = gemm<double>(E, E->dD, tmp, out, false, false, ih))) return rc; // Y = D T
    } else {
      if ((rc = gemm<double>(E, E->dD, in, tmp, true, false, ih))) return rc;   // T = D^T Y
      if ((rc = gemm<double>(E, tmp, E->dD, out, false, false, ih))) return rc; // X = T D
    }
  } else {
    if (!inverse) {
      if ((rc = gemm<float>(E, in, E->dD, tmp, false, true, ih))) return rc;
      if ((rc = gemm<float>(E, E->dD, tmp, out, false, false, ih))) return rc;
    } else {
      if ((rc = gemm<float>(E, E->dD, in, tmp, true, false, ih))) return rc;
      if ((rc = gemm<float>(E, tmp, E->dD, out, false, false, ih))) return rc;
    }
  }
  return CHS_OK;
}
