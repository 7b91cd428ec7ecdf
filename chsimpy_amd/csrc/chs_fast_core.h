// chs_fast_core.h -- device core of the fast transform engine (gfx950).
//
// One length-N orthonormal DCT-II (scipy.fftpack.dct(type=2, norm='ortho'), the 1-D
// factor of the dctn calls at chsimpy/solver.py:159,201) is computed by a GROUP of G
// lanes (G <= 64: inside one wavefront, exchanges behind wavefront fences; G = 128 / 256 at N >= 4096: two or four
// wavefronts of one workgroup, exchanges behind the workgroup barrier -- FCfg::WAVE_LOCAL, xsync) as
//     Makhoul reordering + packing  z[n] = v[2n] + i v[2n+1]        (implicit in the loads)
//  -> M = N/2 point complex FFT, decimation in frequency, 2 or 3 register-resident
//     radix passes with the operands exchanged through a group-private LDS scratch
//  -> real-FFT recombination of (Z[k], Z[M-k]) and the quarter-wave twiddle.
// Each lane keeps E = M/G complex values in registers, every one a Cx<T> (chs_cx.h: two fp64 registers, or one
// packed fp32 register pair on which a complex add is ONE instruction).  Two ownership tricks remove
// every exchange except the ones between radix passes:
//   * pass 0 owns MIRROR PAIRS of butterflies (m', L1-1-m'): the 32-byte quads
//     x[4q..4q+3] a lane loads contain exactly the operands of its two butterflies;
//   * the last pass owns MIRROR PAIRS (kappa, S2-kappa): Z[k] and Z[M-k] meet in one lane.
// The DCT-III (inverse, solver.py:208) is the exact transpose of this network run
// backwards (the orthonormal DCT matrix is orthogonal).
// tools/dct_model.py is the executable model of these index maps.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "chs_cx.h"

// ---------------------------------------------------------------------------
// compile-time configuration
// ---------------------------------------------------------------------------
template <typename T_, int N_, int G_, int THREADS_, int R0_, int RA_, int RB_, int RL_, int PAD1_, int PAD2_,
          int PADL_, int WPS_, int CT_ = THREADS_ / G_, int XPAIR_ = -1>
struct FCfg {
  using T = T_;
  using V = Cx<T_>;  // the complex value type of the core
  static constexpr int N = N_;
  static constexpr int M = N_ / 2;
  static constexpr int G = G_;          // lanes per transform (a "group")
  static constexpr int E = M / G_;      // complex values per lane
  // radices: pass 0, up to two middle passes (1 = absent), last pass
  static constexpr int R0 = R0_, RA = RA_, RB = RB_, RL = RL_;
  static constexpr int R2 = RL_;        // alias used by the recombination stage
  static constexpr int S1 = R0_;             // resolved frequency digits before pass A
  static constexpr int S2A = R0_ * RA_;      // ... before pass B
  static constexpr int SL = R0_ * RA_ * RB_; // number of last-pass butterflies
  static constexpr int S2 = SL;              // alias used by the recombination stage
  static constexpr int L1 = M / R0_;         // sub-transform length after pass 0
  static constexpr int L2 = L1 / RA_;        // ... after pass A
  static constexpr int L3 = L2 / RB_;        // ... after pass B  (== RL)
  static constexpr int NP0 = E / (2 * R0_);  // mirror pairs per lane, pass 0
  static constexpr int NBA = E / RA_;        // butterflies per lane, pass A
  static constexpr int NBB = E / RB_;        // butterflies per lane, pass B
  static constexpr int NP2 = E / (2 * RL_);  // mirror pairs per lane, last pass
  static constexpr int THREADS = THREADS_;
  static constexpr int C = THREADS_ / G_;    // transforms per workgroup
  static constexpr int CT = CT_;             // columns per tile of the T1/T2 layout (k_col needs C == CT)
  static_assert(CT_ % (THREADS_ / G_) == 0, "a tile must hold whole workgroups");
  static constexpr int WPS = WPS_;           // waves per SIMD the kernels are compiled for
  static constexpr int P1 = L1 + PAD1_;      // pitch of X1[kappa < S1][m < L1]
  static constexpr int P2 = L2 + PAD2_;      // pitch of X2[kappa < S2A][m < L2]
  static constexpr int PL = SL + PADL_;      // pitch of XL[m < RL][kappa < SL]
  static constexpr int SCR1 = (RA_ > 1) ? S1 * P1 : 0;
  static constexpr int SCR2 = (RB_ > 1) ? S2A * P2 : 0;
  static constexpr int SCRL = RL_ * PL;
  // PAIR: the real and the imaginary part of a value travel TOGETHER through the exchange scratch as one item of
  // 2*sizeof(T) bytes (half the LDS instructions and half the barriers of an exchange; the scratch then holds twice
  // as many elements); otherwise one after the other through the same scratch (half the LDS).  XPAIR_ < 0: the
  // default of the element type (fp32 paired, fp64 split); 0 / 1: as given (chs_fast_f32.hip, chs_fast_f64.hip).
  static constexpr bool PAIR = (XPAIR_ < 0) ? (sizeof(T_) == 4) : (XPAIR_ != 0);
  // fp32: the columns of T in SLOT order (chs_fast_kernels.h: slot_boff) -- a lane's four coefficients of a slot are one
  // 16-byte access instead of four 4-byte ones.  fp64 keeps the natural column order: there a slot would be a whole
  // 32-byte sector written in two half-sector instructions by every lane, where the natural order has four lanes fill a
  // sector in ONE instruction (measured: N=4096 fp64 row kernel 100 -> 120 us in slot order; N=8192 fp32 288 -> 253 us)
  static constexpr bool SLOT = (sizeof(T_) == 4);
  static constexpr int SCR = (SCR1 > SCR2 ? (SCR1 > SCRL ? SCR1 : SCRL) : (SCR2 > SCRL ? SCR2 : SCRL)) * (PAIR ? 2 : 1);
  static constexpr bool WAVE_LOCAL = (G_ <= 64);  // a group inside one wavefront needs no s_barrier
  static_assert(L3 == RL_, "radices must multiply to M");
  static_assert(RA_ > 1 || RB_ == 1, "use RA before RB");
  static_assert(NP0 >= 1 && NP2 >= 1, "E must be >= 2*R0 and >= 2*RL");
  static_assert(RA_ == 1 || NBA >= 1, "E must be >= RA");
  static_assert(RB_ == 1 || NBB >= 1, "E must be >= RB");
  static_assert((G_ <= 64 && (64 % G_) == 0) || (G_ % 64) == 0, "group size");
  static_assert(THREADS_ % G_ == 0 && THREADS_ % 64 == 0, "workgroup size");
};

// twiddle tables in global memory (complex interleaved: re, im)
template <typename T>
struct FTables {
  const T* tw0;  // [(k-1)*L1 + m] = omega_M^(m k),    k = 1..R0-1, m < L1
  const T* twa;  // [(k-1)*L2 + m] = omega_L1^(m k),   k = 1..RA-1, m < L2
  const T* twb;  // [(k-1)*L3 + m] = omega_L2^(m k),   k = 1..RB-1, m < L3
  const T* wp;   // [kk] = -i exp(-2 pi i kk / N),          kk = 0..M
  const T* t1;   // [kk] = s_kk/2 exp(-i pi kk/(2N))
  const T* t2;   // [kk] = conj(s_(M-kk)/2 exp(-i pi (M-kk)/(2N)))
  // fp32 engine, k_col's spectral stage: the eigenvalues and gradient weights of a recombination slot's four
  // coefficients {kk, N-kk, M-kk, M+kk} as ONE 16-byte entry each, so that they arrive as the two register
  // pairs the packed arithmetic wants; entry M+1 = the special lane's own slot {0, M/2, M, 3M/2}
  const T* lam4;  // [4*j + t] = lambda of coefficient t of slot j (utils.py:35)
  const T* sin4;  // [4*j + t] = sin^2(pi k_t / N)
};

// ---------------------------------------------------------------------------
// the constant roots of unity (16th roots) and the small DFTs
// ---------------------------------------------------------------------------
#define FC_SQRT1_2 0.70710678118654752440
#define FC_COS_PI_8 0.92387953251128675613
#define FC_SIN_PI_8 0.38268343236508977173

// index of exp(-2 pi i IDX/16) (CONJ flips the sign of the angle), reduced to 0..15
template <int IDX, bool CONJ>
struct W16 {
  static constexpr int idx = ((CONJ ? -IDX : IDX) % 16 + 16) % 16;
  static constexpr bool pure = (idx % 4) == 0;  // 1, -i, -1, +i: no multiplication
};

// a * exp(-2 pi i IDX/16)
template <class V, int IDX, bool CONJ>
__device__ __forceinline__ V mul_w16(V a) {
  constexpr int idx = W16<IDX, CONJ>::idx;
  using S = decltype(cx_re(a));
  const V one = cx_make(S(1), S(1));
  if constexpr (idx == 0) {
    return a;
  } else if constexpr (idx == 4) {   // -i: (i, -r)
    return pk_mul_k<1, 0, 0, 1, 0, 1, 0, 0>(a, one);
  } else if constexpr (idx == 8) {
    return pk_mul_k<0, 1, 0, 1, 1, 1, 0, 0>(a, one);
  } else if constexpr (idx == 12) {  // +i: (-i, r)
    return pk_mul_k<1, 0, 0, 1, 1, 0, 0, 0>(a, one);
  } else {
    // exp(-i a) = c - i s with a = 2 pi idx/16
    constexpr double cs[16] = {1.0, FC_COS_PI_8, FC_SQRT1_2, FC_SIN_PI_8, 0.0, -FC_SIN_PI_8, -FC_SQRT1_2, -FC_COS_PI_8,
                               -1.0, -FC_COS_PI_8, -FC_SQRT1_2, -FC_SIN_PI_8, 0.0, FC_SIN_PI_8, FC_SQRT1_2, FC_COS_PI_8};
    constexpr double sn[16] = {0.0, FC_SIN_PI_8, FC_SQRT1_2, FC_COS_PI_8, 1.0, FC_COS_PI_8, FC_SQRT1_2, FC_SIN_PI_8,
                               0.0, -FC_SIN_PI_8, -FC_SQRT1_2, -FC_COS_PI_8, -1.0, -FC_COS_PI_8, -FC_SQRT1_2, -FC_SIN_PI_8};
    return cx_mul_k(a, cx_make((S)cs[idx], (S)-sn[idx]));
  }
}

// (a, b) <- (a + w b, a - w b) for a pure 16th root w = exp(-2 pi i ROT/16) in {1, -i, -1, +i}: the rotation
// is folded into the two additions
template <class V, int ROT>
__device__ __forceinline__ void bfly2_rot(V& a, V& b) {
  static_assert(ROT % 4 == 0, "pure rotations only");
  constexpr int r = ((ROT % 16) + 16) % 16;
  const V x = a, y = b;
  if constexpr (r == 0) { a = cx_add(x, y); b = cx_sub(x, y); }
  else if constexpr (r == 4) { a = cx_add_mi(x, y); b = cx_add_pi(x, y); }
  else if constexpr (r == 8) { a = cx_sub(x, y); b = cx_add(x, y); }
  else { a = cx_add_pi(x, y); b = cx_add_mi(x, y); }
}

// In-place DFT of size R on z[0..R): y[k] = sum_j a[j] exp(-+2 pi i jk/R)
// (INV: conjugate roots, unnormalised).  Natural order in and out.
template <class V, int R, bool INV>
struct Dft;

template <class V, bool INV>
struct Dft<V, 1, INV> {
  static __device__ __forceinline__ void run(V*) {}
};

template <class V, bool INV>
struct Dft<V, 2, INV> {
  static __device__ __forceinline__ void run(V* z) { bfly2_rot<V, 0>(z[0], z[1]); }
};

template <class V, bool INV>
struct Dft<V, 4, INV> {
  static __device__ __forceinline__ void run(V* z) {
    const V t0 = cx_add(z[0], z[2]), t1 = cx_sub(z[0], z[2]);
    const V t2 = cx_add(z[1], z[3]), t3 = cx_sub(z[1], z[3]);
    z[0] = cx_add(t0, t2);
    z[2] = cx_sub(t0, t2);
    // t1 +- (-i) t3 forward, t1 +- (+i) t3 inverse
    if constexpr (!INV) { z[1] = cx_add_mi(t1, t3); z[3] = cx_add_pi(t1, t3); }
    else { z[1] = cx_add_pi(t1, t3); z[3] = cx_add_mi(t1, t3); }
  }
};

// R = A*B Cooley-Tukey with A = 4:  n = B n1 + n2,  k = k1 + A k2
template <class V, int R, bool INV>
struct Dft {
  static constexpr int A = 4, B = R / 4;
  static_assert(R == 8 || R == 16, "radix must be 2, 4, 8 or 16");
  template <int N2, int K1>
  static constexpr int tw_idx() { return (N2 * K1 * 16) / R; }
  // the twiddle between the two stages; B == 2: a pure rotation is left to the second stage (bfly2_rot)
  template <int N2, int K1>
  static __device__ __forceinline__ V tw(V a) {
    if constexpr (B == 2 && W16<tw_idx<N2, K1>(), INV>::pure) return a;
    else return mul_w16<V, tw_idx<N2, K1>(), INV>(a);
  }
  template <int N2>
  static __device__ __forceinline__ void col(const V* z, V (*c)[A]) {
    V x[A];
#pragma unroll
    for (int n1 = 0; n1 < A; ++n1) x[n1] = z[B * n1 + N2];
    Dft<V, A, INV>::run(x);
    c[N2][0] = x[0];
    c[N2][1] = tw<N2, 1>(x[1]);
    c[N2][2] = tw<N2, 2>(x[2]);
    c[N2][3] = tw<N2, 3>(x[3]);
  }
  template <int K1>
  static __device__ __forceinline__ void row(V* z, V (*c)[A]) {
    if constexpr (B == 2) {
      V y0 = c[0][K1], y1 = c[1][K1];
      constexpr int rot = W16<tw_idx<1, K1>(), INV>::pure ? W16<tw_idx<1, K1>(), INV>::idx : 0;
      bfly2_rot<V, rot>(y0, y1);
      z[K1] = y0; z[K1 + A] = y1;
    } else {
      V y[B];
#pragma unroll
      for (int n2 = 0; n2 < B; ++n2) y[n2] = c[n2][K1];
      Dft<V, B, INV>::run(y);
#pragma unroll
      for (int k2 = 0; k2 < B; ++k2) z[K1 + A * k2] = y[k2];
    }
  }
  static __device__ __forceinline__ void run(V* z) {
    V c[B][A];
    col<0>(z, c);
    col<1>(z, c);
    if constexpr (B == 4) {
      col<2>(z, c);
      col<3>(z, c);
    }
    row<0>(z, c);
    row<1>(z, c);
    row<2>(z, c);
    row<3>(z, c);
  }
};

// twiddle tables hold complex values interleaved (re, im): one 16-byte (fp64) or 8-byte (fp32) load
template <typename T>
__device__ __forceinline__ Cx<T> ldc(const T* __restrict__ tab, int idx) {
  if constexpr (sizeof(T) == 8) {
    const double2 v = *reinterpret_cast<const double2*>(tab + 2 * (size_t)idx);
    return cx_make(v.x, v.y);
  } else {
    return *reinterpret_cast<const v2f*>(tab + 2 * (size_t)idx);
  }
}

// the same with a non-temporal hint (operands that are touched once per pass over a grid no cache holds)
typedef double chs_dv2 __attribute__((ext_vector_type(2)));
template <typename T, bool NT>
__device__ __forceinline__ Cx<T> ldc_hint(const T* __restrict__ tab, int idx) {
  if constexpr (!NT) {
    return ldc<T>(tab, idx);
  } else if constexpr (sizeof(T) == 8) {
    const chs_dv2 v = __builtin_nontemporal_load(reinterpret_cast<const chs_dv2*>(tab + 2 * (size_t)idx));
    return cx_make(v.x, v.y);
  } else {
    return __builtin_nontemporal_load(reinterpret_cast<const v2f*>(tab + 2 * (size_t)idx));
  }
}

// Group-level synchronisation of the LDS exchange.  A group inside one wavefront needs no
// s_barrier (the LDS executes one wave's accesses in issue order): wavefront-scope fences
// keep the COMPILER from moving LDS accesses across the hand-over point.  A group spanning
// wavefronts uses the workgroup barrier.
template <class C>
__device__ __forceinline__ void xsync() {
  if constexpr (C::WAVE_LOCAL) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// ownership helpers (runtime, per lane)
// ---------------------------------------------------------------------------
template <class C>
struct Own {
  // last-pass pair q of lane l: kappa1, kappa2 and whether it is the special pair (0, S2/2)
  static __device__ __forceinline__ void last_pair(int l, int q, int& k1, int& k2, bool& special) {
    const int kap = l + C::G * q;
    special = (kap == 0);
    k1 = special ? 0 : kap;
    k2 = special ? C::S2 / 2 : C::S2 - kap;
  }
};

template <typename T>
__device__ __forceinline__ void stc(T* __restrict__ tab, int idx, Cx<T> v) {
  if constexpr (sizeof(T) == 8) *reinterpret_cast<double2*>(tab + 2 * (size_t)idx) = make_double2(v.x, v.y);
  else *reinterpret_cast<v2f*>(tab + 2 * (size_t)idx) = v;
}
template <typename T, bool NT>
__device__ __forceinline__ void stc_hint(T* __restrict__ tab, int idx, Cx<T> v) {
  if constexpr (!NT) {
    stc<T>(tab, idx, v);
  } else if constexpr (sizeof(T) == 8) {
    chs_dv2 t; t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<chs_dv2*>(tab + 2 * (size_t)idx));
  } else {
    __builtin_nontemporal_store(v, reinterpret_cast<v2f*>(tab + 2 * (size_t)idx));
  }
}

// ---------------------------------------------------------------------------
// recombination slot and its adjoint (tools/dct_model.py: slot_fwd / slot_adj)
// ---------------------------------------------------------------------------
// the three recombination twiddles of one slot
template <typename T>
struct SlotTw {
  Cx<T> w, a, b;
};
template <typename T>
__device__ __forceinline__ SlotTw<T> slot_tw(const FTables<T>& tb, int kk) {
  SlotTw<T> w;
  w.w = ldc<T>(tb.wp, kk);
  w.a = ldc<T>(tb.t1, kk);
  w.b = ldc<T>(tb.t2, kk);
  return w;
}

// (A, Z) -> Ya = (y0, y1), Yb = (y2, y3): the four real coefficients of the slot
template <typename T>
__device__ __forceinline__ void slot_fwd(Cx<T> A, Cx<T> Z, const SlotTw<T>& w, Cx<T>& Ya, Cx<T>& Yb) {
  // B = conj(Z2); P = A + B; D = A - B
  const Cx<T> P = cx_addc(A, Z), D = cx_subc(A, Z);
  const Cx<T> Q = cx_mul(D, w.w);            // Q = w' D
  const Cx<T> S1 = cx_add(P, Q), S2 = cx_sub(P, Q);
  Ya = cx_mul_cj(S1, w.a);                   // (Re(T1 S1), -Im(T1 S1))
  Yb = cx_mul(S2, w.b);                      // (Re(T2 S2),  Im(T2 S2))
}

template <typename T>
__device__ __forceinline__ void slot_adj(Cx<T> Ya, Cx<T> Yb, const SlotTw<T>& w, Cx<T>& gA, Cx<T>& gZ) {
  // gS1 = conj(T1 * (y0 + i y1));  gS2 = conj(T2) * (y2 + i y3)
  const Cx<T> g1 = cx_mul_cj(Ya, w.a), g2 = cx_mulc(Yb, w.b);
  const Cx<T> gP = cx_add(g1, g2);
  const Cx<T> gD = cx_mulc(cx_sub(g1, g2), w.w);  // gD = conj(w') gQ
  gA = cx_add(gP, gD);
  gZ = cx_sub_cj(gP, gD);                    // gB = gP - gD ; gZ2 = conj(gB)
}

// ===========================================================================
// Radix passes.  Register index conventions (z[] = the lane's E complex values):
//   pass 0 operands / results   ((q*2+b)*R0 + j)   q < NP0 mirror pairs, b: m1 | m2 = L1-1-m1
//   pass A / pass B             (ib*R + j)          butterfly id = l + G*ib -> (kappa = id % S, m = id / S)
//   last pass                   ((q*2+b)*RL + j)    q < NP2 mirror pairs, b: kappa1 | kappa2
// Exchange buffers in the group's LDS scratch `scr` (C::SCR elements):
//   X1[kappa][m] at kappa*P1 + m,  X2[kappa][m] at kappa*P2 + m,  XL[m][kappa] at m*PL + kappa.
// fp32 (C::PAIR): a value travels as one 8-byte item (the register pair as it stands); fp64: the real parts of
// all values first, then the imaginary parts through the same scratch (half the LDS).
// ===========================================================================

// one middle pass of the forward transform, in registers
template <class C, int S_IN, int L_OUT, int R, int NB>
__device__ __forceinline__ void mid_fwd(typename C::V* z, const typename C::T* tw, int l) {
  using T = typename C::T;
  using V = typename C::V;
#pragma unroll
  for (int ib = 0; ib < NB; ++ib) {
    const int mm = (l + C::G * ib) / S_IN;
    V* r = z + ib * R;
    Dft<V, R, false>::run(r);
#pragma unroll
    for (int k = 1; k < R; ++k) r[k] = cx_mul(r[k], ldc<T>(tw, (k - 1) * L_OUT + mm));
  }
}
template <class C, int S_IN, int L_OUT, int R, int NB>
__device__ __forceinline__ void mid_inv(typename C::V* z, const typename C::T* tw, int l) {
  using T = typename C::T;
  using V = typename C::V;
#pragma unroll
  for (int ib = 0; ib < NB; ++ib) {
    const int mm = (l + C::G * ib) / S_IN;
    V* r = z + ib * R;
#pragma unroll
    for (int k = 1; k < R; ++k) r[k] = cx_mulc(r[k], ldc<T>(tw, (k - 1) * L_OUT + mm));
    Dft<V, R, true>::run(r);
  }
}

// ---- register <-> LDS movers; WR = true: registers -> LDS, false: LDS -> registers; PART (fp64 only): 0 = the
// real parts, 1 = the imaginary parts
template <class C, bool WR, int PART>
__device__ __forceinline__ void xfer(typename C::V& v, typename C::T* scr, int addr) {
  if constexpr (C::PAIR && sizeof(typename C::T) == 4) {
    v2f* s2 = reinterpret_cast<v2f*>(scr);
    if constexpr (WR) s2[addr] = v; else v = s2[addr];
  } else if constexpr (C::PAIR) {
    double2* s2 = reinterpret_cast<double2*>(scr);  // one 16-byte LDS access
    if constexpr (WR) s2[addr] = make_double2(v.x, v.y);
    else { const double2 t = s2[addr]; v.x = t.x; v.y = t.y; }
  } else {
    if constexpr (WR) scr[addr] = PART ? v.y : v.x;
    else if constexpr (PART) v.y = scr[addr];
    else v.x = scr[addr];
  }
}
// pass-0 results (q,b,k) <-> X1[k][m_b]   (or XL[m_b][k] when there is no middle pass)
template <class C, bool WR, int PART>
__device__ __forceinline__ void mv_pass0(typename C::V* z, typename C::T* scr, int l) {
#pragma unroll
  for (int q = 0; q < C::NP0; ++q) {
    const int m1 = l + C::G * q;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int m = b ? (C::L1 - 1 - m1) : m1;
#pragma unroll
      for (int k = 0; k < C::R0; ++k) {
        if constexpr (C::RA > 1) xfer<C, WR, PART>(z[(q * 2 + b) * C::R0 + k], scr, k * C::P1 + m);
        else xfer<C, WR, PART>(z[(q * 2 + b) * C::R0 + k], scr, m * C::PL + k);
      }
    }
  }
}
// pass-A operands (ib,j) <-> X1[kappa][mm + L2*j]
template <class C, bool WR, int PART>
__device__ __forceinline__ void mv_a_in(typename C::V* z, typename C::T* scr, int l) {
#pragma unroll
  for (int ib = 0; ib < C::NBA; ++ib) {
    const int id = l + C::G * ib;
    const int kap = id % C::S1, mm = id / C::S1;
#pragma unroll
    for (int j = 0; j < C::RA; ++j) xfer<C, WR, PART>(z[ib * C::RA + j], scr, kap * C::P1 + mm + C::L2 * j);
  }
}
// pass-A results (ib,k) <-> X2[kappa + S1*k][mm]   (or XL[mm][kappa + S1*k] when pass B is absent)
template <class C, bool WR, int PART>
__device__ __forceinline__ void mv_a_out(typename C::V* z, typename C::T* scr, int l) {
#pragma unroll
  for (int ib = 0; ib < C::NBA; ++ib) {
    const int id = l + C::G * ib;
    const int kap = id % C::S1, mm = id / C::S1;
#pragma unroll
    for (int k = 0; k < C::RA; ++k) {
      if constexpr (C::RB > 1) xfer<C, WR, PART>(z[ib * C::RA + k], scr, (kap + C::S1 * k) * C::P2 + mm);
      else xfer<C, WR, PART>(z[ib * C::RA + k], scr, mm * C::PL + kap + C::S1 * k);
    }
  }
}
// pass-B operands (ib,j) <-> X2[kappa][mm + L3*j]
template <class C, bool WR, int PART>
__device__ __forceinline__ void mv_b_in(typename C::V* z, typename C::T* scr, int l) {
#pragma unroll
  for (int ib = 0; ib < C::NBB; ++ib) {
    const int id = l + C::G * ib;
    const int kap = id % C::S2A, mm = id / C::S2A;
#pragma unroll
    for (int j = 0; j < C::RB; ++j) xfer<C, WR, PART>(z[ib * C::RB + j], scr, kap * C::P2 + mm + C::L3 * j);
  }
}
// pass-B results (ib,k) <-> XL[mm][kappa + S2A*k]
template <class C, bool WR, int PART>
__device__ __forceinline__ void mv_b_out(typename C::V* z, typename C::T* scr, int l) {
#pragma unroll
  for (int ib = 0; ib < C::NBB; ++ib) {
    const int id = l + C::G * ib;
    const int kap = id % C::S2A, mm = id / C::S2A;
#pragma unroll
    for (int k = 0; k < C::RB; ++k) xfer<C, WR, PART>(z[ib * C::RB + k], scr, mm * C::PL + kap + C::S2A * k);
  }
}
// last-pass operands (q,b,j) <-> XL[j][kappa_b]
template <class C, bool WR, int PART>
__device__ __forceinline__ void mv_last(typename C::V* z, typename C::T* scr, int l) {
#pragma unroll
  for (int q = 0; q < C::NP2; ++q) {
    int k1, k2; bool sp;
    Own<C>::last_pair(l, q, k1, k2, sp);
#pragma unroll
    for (int j = 0; j < C::RL; ++j) {
      xfer<C, WR, PART>(z[(q * 2 + 0) * C::RL + j], scr, j * C::PL + k1);
      xfer<C, WR, PART>(z[(q * 2 + 1) * C::RL + j], scr, j * C::PL + k2);
    }
  }
}

// one exchange: WRITER moves the registers out, READER brings the new ownership in
#define CHS_EXCHANGE(WRITER, READER)                                                   \
  do {                                                                                 \
    xsync<C>(); WRITER<C, true, 0>(z, scr, l); xsync<C>(); READER<C, false, 0>(z, scr, l);   \
    if constexpr (!C::PAIR) {                                                          \
      xsync<C>(); WRITER<C, true, 1>(z, scr, l); xsync<C>(); READER<C, false, 1>(z, scr, l); \
    }                                                                                  \
  } while (0)

// The pass-0 twiddles omega_M^(m k), k = 1..R0-1, of butterfly m.  POW: only the k = 1 entries are at hand
// (tw0[m], in LDS where the whole table does not fit beside two resident workgroups): the others are its powers,
// w_k = w_(k/2) w_(k - k/2) -- at most four products deep for R0 = 16.
template <class C, bool POW>
__device__ __forceinline__ void tw0_load(const typename C::T* tw0, int m, typename C::V* w /* [R0], w[0] unused */) {
  using T = typename C::T;
  if constexpr (!POW) {
#pragma unroll
    for (int k = 1; k < C::R0; ++k) w[k] = ldc<T>(tw0, (k - 1) * C::L1 + m);
  } else {
    w[1] = ldc<T>(tw0, m);
#pragma unroll
    for (int k = 2; k < C::R0; ++k) w[k] = cx_mul(w[k / 2], w[k - k / 2]);
  }
}

// Forward: pass-0 operands in -> last-pass outputs Z out (index ((q*2+b)*RL + k)).
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// `before_last()` runs right in front of the last-pass butterflies (behind the last exchange): the place to request what the
// stage behind the transform needs first, so that its round trip runs under those butterflies
template <class C, bool TW0POW = false, class HOOK = NoHook>
__device__ __forceinline__ void fwd_passes(typename C::V* z, typename C::T* scr, const FTables<typename C::T>& tb, int l,
                                           HOOK&& before_last = HOOK()) {
  using T = typename C::T;
  using V = typename C::V;
  // ---- pass 0: radix R0 on every owned butterfly, then twiddle by omega_M^(m k)
#pragma unroll
  for (int q = 0; q < C::NP0; ++q) {
    const int m1 = l + C::G * q;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int m = b ? (C::L1 - 1 - m1) : m1;
      V* r = z + (q * 2 + b) * C::R0;
      Dft<V, C::R0, false>::run(r);
      V w[C::R0];
      tw0_load<C, TW0POW>(tb.tw0, m, w);
#pragma unroll
      for (int k = 1; k < C::R0; ++k) r[k] = cx_mul(r[k], w[k]);
    }
  }
  if constexpr (C::RA > 1) {
    CHS_EXCHANGE(mv_pass0, mv_a_in);
    mid_fwd<C, C::S1, C::L2, C::RA, C::NBA>(z, tb.twa, l);
    if constexpr (C::RB > 1) {
      CHS_EXCHANGE(mv_a_out, mv_b_in);
      mid_fwd<C, C::S2A, C::L3, C::RB, C::NBB>(z, tb.twb, l);
      CHS_EXCHANGE(mv_b_out, mv_last);
    } else {
      CHS_EXCHANGE(mv_a_out, mv_last);
    }
  } else {
    CHS_EXCHANGE(mv_pass0, mv_last);
  }
  // ---- last pass
  before_last();
#pragma unroll
  for (int q = 0; q < 2 * C::NP2; ++q) Dft<V, C::RL, false>::run(z + q * C::RL);
}

// Inverse (exact transpose): last-pass output gradients in -> pass-0 operands out.
template <class C, bool TW0POW = false>
__device__ __forceinline__ void inv_passes(typename C::V* z, typename C::T* scr, const FTables<typename C::T>& tb, int l) {
  using T = typename C::T;
  using V = typename C::V;
#pragma unroll
  for (int q = 0; q < 2 * C::NP2; ++q) Dft<V, C::RL, true>::run(z + q * C::RL);
  if constexpr (C::RA > 1) {
    if constexpr (C::RB > 1) {
      CHS_EXCHANGE(mv_last, mv_b_out);
      mid_inv<C, C::S2A, C::L3, C::RB, C::NBB>(z, tb.twb, l);
      CHS_EXCHANGE(mv_b_in, mv_a_out);
    } else {
      CHS_EXCHANGE(mv_last, mv_a_out);
    }
    mid_inv<C, C::S1, C::L2, C::RA, C::NBA>(z, tb.twa, l);
    CHS_EXCHANGE(mv_a_in, mv_pass0);
  } else {
    CHS_EXCHANGE(mv_last, mv_pass0);
  }
  // ---- pass 0 transposed
#pragma unroll
  for (int q = 0; q < C::NP0; ++q) {
    const int m1 = l + C::G * q;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int m = b ? (C::L1 - 1 - m1) : m1;
      V* r = z + (q * 2 + b) * C::R0;
      V w[C::R0];
      tw0_load<C, TW0POW>(tb.tw0, m, w);
#pragma unroll
      for (int k = 1; k < C::R0; ++k) r[k] = cx_mulc(r[k], w[k]);
      Dft<V, C::R0, true>::run(r);
    }
  }
}

// ===========================================================================
// Recombination stage, in place on the last-pass registers.
//   FWD: (A, Z) -> Ya = (y0, y1), Yb = (y2, y3): the four real coefficients of the slot (slot_fwd)
//   f(pbase, idx, Ya, Yb, live, fetched): the caller consumes / replaces them.  pbase = (q*R2 + k)*4 is the
//        compile-time position of the slot, idx[t] the coefficient index of y_t, idx[4] the
//        slot's entry in per-slot tables: kk, or M + 1 for the special lane's own slot (indices 0, M/2, M, 3M/2).
//   ADJ: (Ya, Yb) -> (gA, gZ) written back over (A, Z) (slot_adj)
// With FWD only the registers are left untouched; with ADJ only Ya, Yb come from f.
// ===========================================================================
#ifndef CHS_RSTAMP
#define CHS_RSTAMP(I) do {} while (0)
#define CHS_RSTAMP1(I) do {} while (0)
#endif
struct NoFetch {};
// x through an opaque asm: address arithmetic derived from it is redone at the point of use instead
// of being computed for all slots up front and kept in registers
__device__ __forceinline__ int fc_opaque(int x) {
  asm volatile("" : "+v"(x));
  __builtin_assume(x >= 0 && x < 8192);  // lane and butterfly indices: keeps index arithmetic unsigned
  return x;
}

// Recombination slots of one lane.  PIPE: the twiddles of slot k+1 and whatever `pre(pbase, idx)` fetches for it
// (global loads only) are requested after slot k's `f(pbase, idx, y, live, fetched)` has computed its results
// and before `st(pbase, idx, y, live)` stores them.  A load requested after a store cannot be waited for without
// waiting for the store too (one in-order vmcnt counter), so without this every slot would sit out the full
// latency of the previous slot's stores.  !PIPE: everything of slot k is fetched in slot k (f may load and store
// as it likes, st is empty).
// PREALL (small grids, PreAll<C>): the twiddles and fetches of ALL slots are requested before the first slot is
// computed -- a small grid is a handful of workgroups with nothing to overlap with, and slot by slot every request is a
// round trip to L2 that nothing hides (N=512: 4-5 round trips in a row per kernel); the registers are there.
template <class C>
struct PreAll { static constexpr bool value = (C::E <= 8); };
// (recombine's last parameter: something the caller requested ahead of the stage for the first slot of pair 0; its
// `patch(fetched)` puts it in place of what the slot's own request would bring -- the compiler drops that part of the request)
struct NoSlot0 {};
template <class C, bool FWD, bool ADJ, bool PIPE_, class PRE, class F, class ST, class S0 = NoSlot0>
__device__ __forceinline__ void recombine(typename C::V* z, const FTables<typename C::T>& tb, int l, PRE&& pre, F&& f, ST&& st,
                                          const S0* ext0 = nullptr) {
  using T = typename C::T;
  using V = typename C::V;
  constexpr bool PREALL = PreAll<C>::value;
  constexpr bool PIPE = PIPE_ && !PREALL;
  constexpr bool EXT0 = PIPE && !std::is_same<S0, NoSlot0>::value;   // part of pair 0's first request was made by the caller
  constexpr int R2 = C::R2, N = C::N, M = C::M, H = R2 / 2;
  const V zero = cx_make(T(0), T(0));
#pragma unroll
  for (int q = 0; q < C::NP2; ++q) {
    V* r1 = z + (q * 2 + 0) * R2;
    V* r2 = z + (q * 2 + 1) * R2;
    const int kap = l + C::G * q;
    // Only butterfly pair 0 = (0, S2/2) (lane 0, q = 0) pairs its outputs inside its own butterflies.
    // A wavefront without that lane runs the plain slots; the wavefront holding it runs the same
    // slots with the special lane's operands selected in (v_cndmask) plus one short slot for the
    // lane itself -- instead of a second, divergent pass over all slots for a single lane.
    bool has_special = false;
    if (q == 0) has_special = (C::G < 64) || (__builtin_amdgcn_readfirstlane(l) == 0);  // (the wavefront that holds butterfly 0)
    if (!has_special) {
      SlotTw<T> wn;
      const int id0[5] = {kap, N - kap, M - kap, M + kap, kap};
      decltype(pre(0, id0)) pn;
      [[maybe_unused]] SlotTw<T> wv[PREALL ? R2 : 1];
      [[maybe_unused]] decltype(pre(0, id0)) pv[PREALL ? R2 : 1];
      if constexpr (PIPE) {
        wn = slot_tw<T>(tb, kap);
        pn = pre(q * R2 * 4, id0);
        if constexpr (EXT0) { if (q == 0) ext0->patch(pn); }
      }
      if constexpr (PREALL) {
#pragma unroll
        for (int k = 0; k < R2; ++k) {
          const int kk = kap + C::S2 * k;
          wv[k] = slot_tw<T>(tb, kk);
          const int idc[5] = {kk, N - kk, M - kk, M + kk, kk};
          pv[k] = pre((q * R2 + k) * 4, idc);
        }
      }
      CHS_RSTAMP1(0);
#pragma unroll
      for (int k = 0; k < R2; ++k) {
        if (k == 1) CHS_RSTAMP1(1);
        if (k == 4) CHS_RSTAMP1(2);
        const int kk = kap + C::S2 * k;
        if constexpr (PREALL) {
          wn = wv[k]; pn = pv[k];
        } else if (!PIPE) {
          wn = slot_tw<T>(tb, kk);
          const int idc[5] = {kk, N - kk, M - kk, M + kk, kk};
          pn = pre((q * R2 + k) * 4, idc);
        }
        const SlotTw<T> w = wn;  // shared by the forward and the adjoint half
        const auto pc = pn;
        V Ya = zero, Yb = zero;
        if constexpr (FWD) slot_fwd<T>(r1[k], r2[R2 - 1 - k], w, Ya, Yb);
        const int idx[5] = {kk, N - kk, M - kk, M + kk, kk};
        f((q * R2 + k) * 4, idx, Ya, Yb, true, pc);
        if (PIPE && k + 1 < R2) {  // requested before this slot's stores
          const int kn = fc_opaque(kap) + C::S2 * (k + 1);
          wn = slot_tw<T>(tb, kn);
          const int idn[5] = {kn, N - kn, M - kn, M + kn, kn};
          pn = pre((q * R2 + k + 1) * 4, idn);
        }
        st((q * R2 + k) * 4, idx, Ya, Yb, true);
        if constexpr (ADJ) slot_adj<T>(Ya, Yb, w, r1[k], r2[R2 - 1 - k]);
      }
      CHS_RSTAMP1(3);
    } else {
      const bool sp = (kap == 0);
      // The special lane: butterfly 0 pairs k <-> R2-k, butterfly S2/2 pairs k <-> R2-1-k.  Its slot
      // s = 1..H-1 works on (r1[s], r1[R2-s]), slot s = H..R2-1 on (r2[s-H], r2[R2-1-(s-H)]); slot 0
      // holds (X[0], X[M/2], X[M], X[3M/2]) from the two self-paired outputs r1[0], r1[H].
      auto sel = [&](V a, V b) { return cx_make(sp ? cx_re(a) : cx_re(b), sp ? cx_im(a) : cx_im(b)); };
      auto kk_of = [&](int k) {
        const int kk_s = (k == 0) ? C::S2 : ((k < H) ? C::S2 * k : C::S2 / 2 + C::S2 * (k - H));  // k = 0: unused
        return sp ? kk_s : kap + C::S2 * k;
      };
      const int kk0 = kk_of(0);
      SlotTw<T> wn;
      // PIPE: the request for slot 0 of the loop below also serves the special lane's own slot.  That lane's
      // slot 0 is dead, so ITS share of the request is made for the own slot's indices (0, M/2, M, 3M/2 and the
      // twiddles of kk = 0) instead: one round trip for the wavefront instead of two in a row (the own slot
      // used to fetch on its own, with nothing to overlap the wait).
      const int o1 = PIPE ? (sp ? 0 : kk0) : kk0, o2 = PIPE ? (sp ? M / 2 : N - kk0) : N - kk0;
      const int o3 = PIPE ? (sp ? M : M - kk0) : M - kk0, o4 = PIPE ? (sp ? 3 * (M / 2) : M + kk0) : M + kk0;
      const int id0[5] = {o1, o2, o3, o4, (PIPE && sp) ? M + 1 : kk0};
      decltype(pre(0, id0)) pn;
      SlotTw<T> whp = SlotTw<T>();
      [[maybe_unused]] SlotTw<T> wv[PREALL ? R2 : 1];
      [[maybe_unused]] decltype(pre(0, id0)) pv[PREALL ? R2 : 1];
      [[maybe_unused]] SlotTw<T> w0v = SlotTw<T>();
      [[maybe_unused]] decltype(pre(0, id0)) p0v;
      if constexpr (PIPE) {  // slot 0 of the loop below, requested ahead of the special lane's own slot
        wn = slot_tw<T>(tb, o1);
        pn = pre(q * R2 * 4, id0);
        if constexpr (EXT0) ext0->patch(pn);   // (the special pair is pair 0: q == 0 here)
        if (sp) whp = slot_tw<T>(tb, M / 2);  // the special lane's second self-paired butterfly
      }
      if constexpr (PREALL) {
        // every lane requests its R2 slots; the special lane's dead slot 0 asks for a valid index (kk_of(0)) and its
        // own slot's operands ride in the same burst
        if (sp) {
          const int ido[5] = {0, M / 2, M, 3 * (M / 2), M + 1};
          w0v = slot_tw<T>(tb, 0); whp = slot_tw<T>(tb, M / 2);
          p0v = pre(q * R2 * 4, ido);
        }
#pragma unroll
        for (int k = 0; k < R2; ++k) {
          const int kk = kk_of(k);
          wv[k] = slot_tw<T>(tb, kk);
          const int idc[5] = {kk, N - kk, M - kk, M + kk, kk};
          pv[k] = pre((q * R2 + k) * 4, idc);
        }
      }
      CHS_RSTAMP(0);
      if (sp) {
        const int idx[5] = {0, M / 2, M, 3 * (M / 2), M + 1};
        SlotTw<T> w0, wh;
        decltype(pre(0, id0)) p0;
        if constexpr (PREALL) {
          w0 = w0v; wh = whp; p0 = p0v;
        } else if constexpr (PIPE) {
          w0 = wn; wh = whp; p0 = pn;
        } else {
          w0 = slot_tw<T>(tb, 0); wh = slot_tw<T>(tb, M / 2);
          p0 = pre(q * R2 * 4, idx);
        }
        V Ya = zero, Yb = zero;
        if constexpr (FWD) {
          V a01, a23, b01, b23;
          slot_fwd<T>(r1[0], r1[0], w0, a01, a23);
          slot_fwd<T>(r1[H], r1[H], wh, b01, b23);
          Ya = cx_make(cx_re(a01), cx_re(b01));
          Yb = cx_make(cx_re(a23), cx_im(b01));
        }
        f(q * R2 * 4, idx, Ya, Yb, true, p0);
        st(q * R2 * 4, idx, Ya, Yb, true);
        if constexpr (ADJ) {
          V ga, gz;
          slot_adj<T>(cx_make(cx_re(Ya), T(0)), cx_make(cx_re(Yb), T(0)), w0, ga, gz);
          r1[0] = cx_add(ga, gz);
          slot_adj<T>(cx_make(cx_im(Ya), cx_im(Yb)), zero, wh, ga, gz);
          r1[H] = cx_add(ga, gz);
        }
      }
      CHS_RSTAMP(1);
#pragma unroll
      for (int k = 0; k < R2; ++k) {
        if (k == 1) CHS_RSTAMP(2);
        if (k == 4) CHS_RSTAMP(3);
        const int kk = kk_of(k);
        if constexpr (PREALL) {
          wn = wv[k]; pn = pv[k];
        } else if (!PIPE) {
          wn = slot_tw<T>(tb, kk);
          const int idc[5] = {kk, N - kk, M - kk, M + kk, kk};
          pn = pre((q * R2 + k) * 4, idc);
        }
        const SlotTw<T> w = wn;
        const auto pc = pn;
        // operand registers: *_n for every other lane, *_s for the special lane
        V* a_n = &r1[k]; V* b_n = &r2[R2 - 1 - k];
        V* a_s = (k < H) ? &r1[k] : &r2[k - H];
        V* b_s = (k == 0) ? &r2[R2 - 1] : ((k < H) ? &r1[R2 - k] : &r2[R2 - 1 - (k - H)]);
        V Ya = zero, Yb = zero;
        if constexpr (FWD) slot_fwd<T>(sel(*a_s, *a_n), sel(*b_s, *b_n), w, Ya, Yb);
        const int idx[5] = {kk, N - kk, M - kk, M + kk, kk};
        const bool live = (k > 0) || !sp;
        f((q * R2 + k) * 4, idx, Ya, Yb, live, pc);
        if (PIPE && k + 1 < R2) {  // requested before this slot's stores
          const int kap_o = fc_opaque(kap);  // (unconditional: an asm in one arm of a select becomes a branch)
          const int kn = sp ? kk_of(k + 1) : kap_o + C::S2 * (k + 1);
          wn = slot_tw<T>(tb, kn);
          const int idn[5] = {kn, N - kn, M - kn, M + kn, kn};
          pn = pre((q * R2 + k + 1) * 4, idn);
        }
        st((q * R2 + k) * 4, idx, Ya, Yb, live);
        if constexpr (ADJ) {
          V na, nb;
          slot_adj<T>(Ya, Yb, w, na, nb);
          if (k == 0) {
            *a_n = sel(*a_n, na);
            *b_n = sel(*b_n, nb);
          } else {
            if (k < H) {
              *a_n = na;
            } else {
              *a_s = sel(na, *a_s);
              *a_n = sel(*a_n, na);
            }
            *b_s = sel(nb, *b_s);
            *b_n = sel(*b_n, nb);
          }
        }
      }
      CHS_RSTAMP(4);
    }
  }
}

// ---------------------------------------------------------------------------
// quad <-> pass-0 operand packing (tools/dct_model.py: forward()/inverse()).
// Lane l, pair q, half-index j < R0/2:
//   quad Q1 = x[4*(m1 + L1 j) ..+3],  Q2 = x[4*(m2 + L1 j) ..+3],  m2 = L1-1-m1
//   a[j] = (Q1.0, Q1.2)   a[R0-1-j] = (Q2.3, Q2.1)      (butterfly m1)
//   b[j] = (Q2.0, Q2.2)   b[R0-1-j] = (Q1.3, Q1.1)      (butterfly m2)
// ---------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void pack_quads(const typename C::T q1[4], const typename C::T q2[4], int q, int j, typename C::V* z) {
  constexpr int R0 = C::R0;
  z[(q * 2 + 0) * R0 + j] = cx_make(q1[0], q1[2]);
  z[(q * 2 + 1) * R0 + j] = cx_make(q2[0], q2[2]);
  z[(q * 2 + 1) * R0 + (R0 - 1 - j)] = cx_make(q1[3], q1[1]);
  z[(q * 2 + 0) * R0 + (R0 - 1 - j)] = cx_make(q2[3], q2[1]);
}
template <class C>
__device__ __forceinline__ void unpack_quads(const typename C::V* z, int q, int j, typename C::T q1[4], typename C::T q2[4]) {
  constexpr int R0 = C::R0;
  q1[0] = cx_re(z[(q * 2 + 0) * R0 + j]); q1[2] = cx_im(z[(q * 2 + 0) * R0 + j]);
  q2[0] = cx_re(z[(q * 2 + 1) * R0 + j]); q2[2] = cx_im(z[(q * 2 + 1) * R0 + j]);
  q1[3] = cx_re(z[(q * 2 + 1) * R0 + (R0 - 1 - j)]); q1[1] = cx_im(z[(q * 2 + 1) * R0 + (R0 - 1 - j)]);
  q2[3] = cx_re(z[(q * 2 + 0) * R0 + (R0 - 1 - j)]); q2[1] = cx_im(z[(q * 2 + 0) * R0 + (R0 - 1 - j)]);
}
