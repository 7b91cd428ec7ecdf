// chs_math.h -- pointwise physics of the timestep, shared by every kernel that
// touches U.  Operation order follows the reference expressions literally;
// floating-point contraction is switched off inside these functions so that
// the fp64 results round exactly like numpy's elementwise chains.
#pragma once
#include <hip/hip_runtime.h>
#include "chs_common.h"

// ---------------------------------------------------------------------------
// fp64 logarithms.  ocml's log() is a ~100-instruction double-double routine; the
// timestep needs three logs per grid point, which would make the loop VALU-bound.
// These are lean (< 1.5 ulp measured on gfx950, tests/test_gpu_math.py) versions:
//   log(q) = 2 atanh(s), s = (q-1)/(q+1), |s| <= 0.1716 after reduction to
//   q in [sqrt(1/2), sqrt(2)];  log(q) = 2s + s*z*P(z), z = s^2, P of degree 6
//   (tools/log_poly.py, approximation error 6e-18).
// Domain handling follows numpy: log(0) = -inf, log(x<0) = NaN, NaN stays NaN, so
// that a field leaving (0,1) still surfaces as the NaN assertion of timedata.py:10.
// ---------------------------------------------------------------------------
#define CHS_LN2_HI 0.6931471803691238      /* 0x1.62e42fee00000p-1 (21 trailing zero bits) */
#define CHS_LN2_LO 1.9082149292705877e-10
#define CHS_SQRT2 1.4142135623730951
#define CHS_SQRT1_2 0.7071067811865476

// d / sig for sig in [0.5, 4): hardware reciprocal + two Newton steps + one residual correction
__device__ __forceinline__ double chs_div_pos(double d, double sig) {
  double r = __builtin_amdgcn_rcp(sig);
  double e = __builtin_fma(-sig, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-sig, r, 1.0);
  r = __builtin_fma(r, e, r);
  double q = d * r;
  const double rho = __builtin_fma(-sig, q, d);
  return __builtin_fma(rho, r, q);
}

// The same quotient with ONE Newton step: the residual correction squares whatever error the
// reciprocal still has (2^-14 would already do), so the result is as accurate as chs_div_pos.
__device__ __forceinline__ double chs_div_pos1(double d, double sig) {
  double r = __builtin_amdgcn_rcp(sig);
  const double e = __builtin_fma(-sig, r, 1.0);
  r = __builtin_fma(r, e, r);
  const double q = d * r;
  const double rho = __builtin_fma(-sig, q, d);
  return __builtin_fma(rho, r, q);
}

// s -> 2 atanh(s) for |s| <= 0.1716
__device__ __forceinline__ double chs_log_poly(double s) {
  const double z = s * s;
  double p = 0.1467141530063111;
  p = __builtin_fma(p, z, 0.15327184863231447);
  p = __builtin_fma(p, z, 0.18183028385247743);
  p = __builtin_fma(p, z, 0.22222209184316374);
  p = __builtin_fma(p, z, 0.2857142863817006);
  p = __builtin_fma(p, z, 0.3999999999987206);
  p = __builtin_fma(p, z, 0.6666666666666671);
  return __builtin_fma(s * z, p, s + s);
}

// log(x)
__device__ __forceinline__ double chs_log_f64(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const int c = (m < CHS_SQRT1_2) ? 1 : 0;
  m = __builtin_ldexp(m, c);                  // [sqrt(1/2), sqrt(2))
  e -= c;
  const double s = chs_div_pos(m - 1.0, m + 1.0);
  const double lq = chs_log_poly(s);
  const double k = (double)e;
  double res = __builtin_fma(k, CHS_LN2_HI, __builtin_fma(k, CHS_LN2_LO, lq));
  // The opaque asm pins the computation above this point: without it the compiler turns the
  // selects below into divergent branches around the whole routine.
  asm volatile("" : "+v"(res));
  const double bad = (x == 0.0) ? -__builtin_inf() : __builtin_nan("");
  res = (x > 0.0) ? res : bad;
  res = (x == __builtin_inf()) ? x : res;
  return res;
}

// log(x) for x known to be positive and finite (the caller checks the domain once per grid
// point instead of three selects per logarithm)
__device__ __forceinline__ double chs_log_pos_f64(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const int c = (m < CHS_SQRT1_2) ? 1 : 0;
  m = __builtin_ldexp(m, c);                  // [sqrt(1/2), sqrt(2))
  e -= c;
  const double s = chs_div_pos1(m - 1.0, m + 1.0);
  const double lq = chs_log_poly(s);
  const double k = (double)e;
  return __builtin_fma(k, CHS_LN2_HI, __builtin_fma(k, CHS_LN2_LO, lq));
}

// log(a / b) without forming the quotient
__device__ __forceinline__ double chs_log_ratio_f64(double a, double b) {
  double ma = __builtin_amdgcn_frexp_mant(a), mb = __builtin_amdgcn_frexp_mant(b);  // [0.5, 1)
  const int ea = __builtin_amdgcn_frexp_exp(a), eb = __builtin_amdgcn_frexp_exp(b);
  const int c1 = (ma > CHS_SQRT2 * mb) ? 1 : 0;   // q > sqrt2  -> halve q
  const int c2 = (CHS_SQRT2 * ma < mb) ? 1 : 0;   // q < 1/sqrt2 -> double q
  mb = __builtin_ldexp(mb, c1);
  ma = __builtin_ldexp(ma, c2);
  const int e = ea - eb + c1 - c2;
  const double s = chs_div_pos(ma - mb, ma + mb);  // ma - mb is exact (Sterbenz)
  const double lq = chs_log_poly(s);
  const double k = (double)e;
  double res = __builtin_fma(k, CHS_LN2_HI, __builtin_fma(k, CHS_LN2_LO, lq));
  // numpy gives NaN / -inf / +inf for a quotient that is negative / 0 / x/0; all of them
  // poison the step and trip the reference's NaN assertion, so one NaN covers them
  asm volatile("" : "+v"(res));  // keep the selects below as selects (see chs_log_f64)
  const bool ok = (a > 0.0) && (b > 0.0) && (a < __builtin_inf()) && (b < __builtin_inf());
  return ok ? res : __builtin_nan("");  // every non-finite case ends in the NaN assertion anyway
}

// Table-driven log for the fused row kernel's pointwise part (no division, no domain selects):
//   x = 2^e m, m in [1/2, 1) (frexp);  i = rint(256 m) in [128, 256];  r = fma(m, rc_i, -1) with
//   rc_i = fl(256/i) from the table (|r| <= 1/256);  log x = (e ln2 + lc_i) + log1p(r), lc_i = fl(-log rc_i),
//   log1p by its series to r^6 (truncation r^7/7 < 2e-18).  `tab` = chs_log_table copied to LDS.
// 13 floating-point + 3 integer instructions and one LDS read.
// Accuracy: for 0 < x <= 1 -- all the timestep ever asks for: x = U or 1-U -- every term has the same
// sign (e <= 0, lc_i <= 0) and the result is within 2.5 ulp (tests/test_gpu_math.py); x -> 1 gives
// e = 0, i = 256, lc = 0: log x = log1p(r) with r = x - 1 exactly.  For x > 1 the terms e ln2 > 0 and
// lc_i < 0 cancel: the ABSOLUTE error stays ~2e-16 max(1, |log x|) but the relative error grows as
// log x -> 0+; ln2 is a single constant (no hi/lo split) because of that sign structure.
// Domain: the index doubles as the domain check.  For finite x > 0 it lies in [128, 256]; x <= 0 gives
// m <= 0 and an index below 128 (-> huge as unsigned), which `domain` (a running unsigned maximum of
// i - 128, to be compared with 128 once per row) records: the caller turns the sums into NaN then,
// like numpy's log of a non-positive number does for the reference (timedata.py:10).  NaN and +inf
// propagate through r.
#include "chs_log_table.h"
#define CHS_LN2 0.6931471805599453
__device__ __forceinline__ double chs_log_unit_tab_f64(double x, const double2* tab, unsigned& domain) {
#pragma clang fp contract(off)
  const double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  const int e = __builtin_amdgcn_frexp_exp(x);
  const double u = __builtin_fma(m, 256.0, 0x1.8p52);  // the low word of 1.5*2^52 + n is n
  const unsigned i = (unsigned)(__double2loint(u) - CHS_LOGTAB_I0);
  domain = max(domain, i);
  const double2 t = tab[min(i, (unsigned)(CHS_LOGTAB_N - 1))];
  const double r = __builtin_fma(m, t.x, -1.0);
  double p = __builtin_fma(r, -1.0 / 6.0, 1.0 / 5.0);
  p = __builtin_fma(r, p, -1.0 / 4.0);
  p = __builtin_fma(r, p, 1.0 / 3.0);
  p = __builtin_fma(r, p, -0.5);
  const double lp = __builtin_fma(r * r, p, r);
  return __builtin_fma((double)e, CHS_LN2, t.y) + lp;
}

// fp32 engine: x = U, 1-U or their quotient, never denormal: the hardware log2 (1 ulp) times ln 2 instead of the
// library's range-checked sequence (~10 instructions per logarithm); log of a non-positive number gives NaN / -inf
// as the library's does.  Every fp32 path uses the same function (the fused kernels and the sweep kernels agree).
#ifndef CHS_F32_NATIVE_LOG
#define CHS_F32_NATIVE_LOG 1
#endif
__device__ __forceinline__ float chs_logf(float x) {
  if constexpr (CHS_F32_NATIVE_LOG != 0) return __builtin_amdgcn_logf(x) * 0.69314718055994530942f;
  return logf(x);
}
template <typename T> __device__ __forceinline__ T chs_log_pos(T x);
template <> __device__ __forceinline__ double chs_log_pos<double>(double x) { return chs_log_pos_f64(x); }
template <> __device__ __forceinline__ float chs_log_pos<float>(float x) { return chs_logf(x); }
template <typename T> __device__ __forceinline__ T chs_log_unit_tab(T x, const double2* tab, unsigned& domain);
template <> __device__ __forceinline__ double chs_log_unit_tab<double>(double x, const double2* tab, unsigned& domain) { return chs_log_unit_tab_f64(x, tab, domain); }
template <> __device__ __forceinline__ float chs_log_unit_tab<float>(float x, const double2*, unsigned& domain) {
  domain = max(domain, (x > 0.0f) ? 0u : ~0u);
  return chs_logf(x);
}

template <typename T> __device__ __forceinline__ T chs_log(T x);
template <> __device__ __forceinline__ double chs_log<double>(double x) { return chs_log_f64(x); }
template <> __device__ __forceinline__ float chs_log<float>(float x) { return chs_logf(x); }

// log(U / Uinv)
template <typename T> __device__ __forceinline__ T chs_log_ratio(T a, T b);
template <> __device__ __forceinline__ double chs_log_ratio<double>(double a, double b) { return chs_log_ratio_f64(a, b); }
template <> __device__ __forceinline__ float chs_log_ratio<float>(float a, float b) { return chs_logf(a / b); }

// EnergieEut, chsimpy/solver.py:166-175:
//   Uinv = 1-U; U1Uinv = U/Uinv; U2inv = Uinv-U
//   RT*log(U1Uinv) - BRT + (A0 + A1*U2inv)*U2inv - 2*A1*U*Uinv
template <typename T>
__device__ __forceinline__ T chs_mu(T U, T RT, T BRT, T A0, T A1) {
#pragma clang fp contract(off)
  const T Uinv = T(1) - U;
  const T U2inv = Uinv - U;
  const T t1 = RT * chs_log_ratio<T>(U, Uinv);  // log(U / Uinv)
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

// The same two expressions with log(U) and log(1-U) supplied by the caller, so that the
// fused row kernel evaluates two logarithms per grid point instead of three
// (log(U/(1-U)) = log U - log(1-U)).
template <typename T>
__device__ __forceinline__ T chs_energy_from_logs(T U, T Uinv, T lU, T lV, T RT, T B, T A0, T A1) {
#pragma clang fp contract(off)
  const T a = U * (lU - B);
  const T b = Uinv * lV;
  const T c = ((A0 + A1 * (Uinv - U)) * U) * Uinv;
  return RT * (a + b) + c;
}
template <typename T>
__device__ __forceinline__ T chs_mu_from_logs(T U, T Uinv, T lU, T lV, T RT, T BRT, T A0, T A1) {
#pragma clang fp contract(off)
  const T U2inv = Uinv - U;
  const T t1 = RT * (lU - lV);
  const T t2 = (A0 + A1 * U2inv) * U2inv;
  const T t3 = ((T(2) * A1) * U) * Uinv;
  return ((t1 - BRT) + t2) - t3;
}

// Fused-kernel variants: same expressions, the compiler may contract a*b+c into one fma (each
// contraction removes a rounding; the results move by < 1 ulp, far inside the 1e-9 tolerance).
template <typename T>
__device__ __forceinline__ T chs_energy_from_logs_fast(T U, T Uinv, T lU, T lV, T RT, T B, T A0, T A1) {
  const T a = U * (lU - B);
  const T b = Uinv * lV;
  const T c = ((A0 + A1 * (Uinv - U)) * U) * Uinv;
  return RT * (a + b) + c;
}
template <typename T>
__device__ __forceinline__ T chs_mu_from_logs_fast(T U, T Uinv, T lU, T lV, T RT, T BRT, T A0, T A1) {
  const T U2inv = Uinv - U;
  const T t1 = RT * (lU - lV);
  const T t2 = (A0 + A1 * U2inv) * U2inv;
  const T t3 = ((T(2) * A1) * U) * Uinv;
  return ((t1 - BRT) + t2) - t3;
}

// Adaptive-step integrand, chsimpy/solver.py:182-183:
//   delt_max / sqrt(1 + delt_alpha*|mu|^2), delt_alpha = 500/2^3
__device__ __forceinline__ double chs_dt_integrand(double mu, double delt_max) {
#pragma clang fp contract(off)
  const double a = 62.5 * (mu * mu);
  return delt_max / sqrt(1.0 + a);
}

// The same integrand where it is evaluated for every grid point of every second step (the fused row kernel): a
// reciprocal square root instead of a square root and a division.  fp64: v_rsq_f64 refined by two Newton steps
// (~2 ulp; the column sums agree with the oracle's to 1e-15); fp32 engine: the hardware's v_rsq_f32 (1 ulp), in the
// precision of its operand mu.
__device__ __forceinline__ double chs_dt_integrand_fast(double mu, double delt_max) {
  const double a = __builtin_fma(62.5 * mu, mu, 1.0);
  const double h = 0.5 * a;
  double y = __builtin_amdgcn_rsq(a);
  y = y * __builtin_fma(-h * y, y, 1.5);
  y = y * __builtin_fma(-h * y, y, 1.5);
  return delt_max * y;
}
__device__ __forceinline__ float chs_dt_integrand_fast(float mu, float delt_max) {
  return delt_max * __builtin_amdgcn_rsqf(__builtin_fmaf(62.5f * mu, mu, 1.0f));
}

// Semi-implicit spectral update, chsimpy/solver.py:201-206 with the grids of
// chsimpy/utils.py:41-48 formed on the fly:
//   leig = lam_i + lam_j; CHeig = 1 + lam2*leig*leig; Seig = lam1*leig
//   hat_U = (hat_U + Seig*hat_mu) / CHeig
// fp32 engine, F32MATH: the same update in single precision (the operands are fp32 values and so is the result;
// the fp64 version below spends ~40 double-precision instructions per coefficient on a quotient that is rounded
// to 24 bits on the way out).  Reciprocal + residual correction: within 1 ulp of the fp32 quotient; the parity
// margins of the fp32 runs do not move (N=512, 1000 steps: U 7.8e-5 -> 8.6e-5 against the fp64 oracle).
// Measured: N=8192 fp32 k_col 460 -> 433 us, N=4096 fp32 76 -> 83 us (the compiler packs it with more moves and 16
// more registers there), so k_col asks for it from CHS_F32_SPECTRAL_MIN_N upwards only.
#ifndef CHS_F32_SPECTRAL_MIN_N
#define CHS_F32_SPECTRAL_MIN_N 8192
#endif
__device__ __forceinline__ float chs_spectral_f32(float hatU, float hatMu, double li, double lj, double lam1, double lam2) {
  const float leig = (float)li + (float)lj;
  const float CHeig = __builtin_fmaf((float)lam2 * leig, leig, 1.0f);
  const float rhs = __builtin_fmaf((float)lam1 * leig, hatMu, hatU);
  const float r = __builtin_amdgcn_rcpf(CHeig);
  const float q = rhs * r;
  const float rho = __builtin_fmaf(-CHeig, q, rhs);
  return __builtin_fmaf(rho, r, q);
}
template <typename T, bool F32MATH = false>
__device__ __forceinline__ T chs_spectral(T hatU, T hatMu, double li, double lj, double lam1, double lam2) {
#pragma clang fp contract(off)
  if constexpr (sizeof(T) == 4 && F32MATH) return chs_spectral_f32(hatU, hatMu, li, lj, lam1, lam2);
  const double leig = li + lj;
#ifndef CHS_SPECTRAL_FMA
#define CHS_SPECTRAL_FMA 0
#endif
#if CHS_SPECTRAL_FMA
  // fused multiply-adds (one rounding less each than numpy's separate operations: differences of an ulp in hat_U)
  const double CHeig = __builtin_fma(lam2 * leig, leig, 1.0);
  const double rhs = __builtin_fma(lam1 * leig, (double)hatMu, (double)hatU);
#else
  const double CHeig = 1.0 + (lam2 * leig) * leig;
  const double Seig = lam1 * leig;
  const double rhs = (double)hatU + Seig * (double)hatMu;
#endif
  // CHeig >= 1: reciprocal + one Newton step + residual correction (the correction squares the
  // reciprocal's remaining error: < 1 ulp of the correctly rounded quotient, measured) instead of the
  // ~2.5x longer IEEE division sequence
  return (T)chs_div_pos1(rhs, CHeig);
}
