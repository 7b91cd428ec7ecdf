// chs_math.h -- pointwise physics of the timestep, shared by every kernel that
// touches U.  Operation order follows the reference expressions literally;
// floating-point contraction is switched off inside these functions so that
// the fp64 results round exactly like numpy's elementwise chains.
#pragma once
#include <hip/hip_runtime.h>
#include "chs_common.h"

template <typename T> __device__ __forceinline__ T chs_log(T x);
template <> __device__ __forceinline__ double chs_log<double>(double x) { return log(x); }
template <> __device__ __forceinline__ float chs_log<float>(float x) { return logf(x); }

// EnergieEut, chsimpy/solver.py:166-175:
//   Uinv = 1-U; U1Uinv = U/Uinv; U2inv = Uinv-U
//   RT*log(U1Uinv) - BRT + (A0 + A1*U2inv)*U2inv - 2*A1*U*Uinv
template <typename T>
__device__ __forceinline__ T chs_mu(T U, T RT, T BRT, T A0, T A1) {
#pragma clang fp contract(off)
  const T Uinv = T(1) - U;
  const T U1Uinv = U / Uinv;
  const T U2inv = Uinv - U;
  const T t1 = RT * chs_log<T>(U1Uinv);
  const T t2 = (A0 + A1 * U2inv) * U2inv;
  const T t3 = ((T(2) * A1) * U) * Uinv;
  return ((t1 - BRT) + t2) - t3;
}

// Bulk free-energy density, chsimpy/solver.py:218-221 (the argument of np.mean):
//   RT*(U*(log(U)-B) + Uinv*log(Uinv)) + (A0 + A1*(Uinv-U))*U*Uinv
template <typename T>
__device__ __forceinline__ T chs_energy_density(T U, T RT, T B, T A0, T A1) {
#pragma clang fp contract(off)
  const T Uinv = T(1) - U;
  const T a = U * (chs_log<T>(U) - B);
  const T b = Uinv * chs_log<T>(Uinv);
  const T c = ((A0 + A1 * (Uinv - U)) * U) * Uinv;
  return RT * (a + b) + c;
}

// Adaptive-step integrand, chsimpy/solver.py:182-183:
//   delt_max / sqrt(1 + delt_alpha*|mu|^2), delt_alpha = 500/2^3
__device__ __forceinline__ double chs_dt_integrand(double mu, double delt_max) {
#pragma clang fp contract(off)
  const double a = 62.5 * (mu * mu);
  return delt_max / sqrt(1.0 + a);
}

// Semi-implicit spectral update, chsimpy/solver.py:201-206 with the grids of
// chsimpy/utils.py:41-48 formed on the fly:
//   leig = lam_i + lam_j; CHeig = 1 + lam2*leig*leig; Seig = lam1*leig
//   hat_U = (hat_U + Seig*hat_mu) / CHeig
template <typename T>
__device__ __forceinline__ T chs_spectral(T hatU, T hatMu, double li, double lj, double lam1, double lam2) {
#pragma clang fp contract(off)
  const double leig = li + lj;
  const double CHeig = 1.0 + (lam2 * leig) * leig;
  const double Seig = lam1 * leig;
  const double rhs = (double)hatU + Seig * (double)hatMu;
  return (T)(rhs / CHeig);
}
