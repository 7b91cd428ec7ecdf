// chs_fast_kernels.h -- fast transform engine: the kernels (templates over the configuration) and their
// launchers.  Included by chs_fast.hip (plans, tables, step sequence) and by the two translation units that
// instantiate the configurations (chs_fast_f64.hip, chs_fast_f32.hip: compiled in parallel).
//
// Three kernels per timestep:
//
//   k_row_fwd  reads U rows, evaluates EnergieEut (chsimpy/solver.py:166-175) on the fly,
//              DCT-II along the row, writes T1                      [1 read + 1 write of N^2]
//   k_col      reads a column tile of T1, DCT-II along the columns, semi-implicit spectral
//              update of the carried hat_U (solver.py:201-206, utils.py:39-49 formed from
//              the 1-D lambda table), DCT-III along the columns, writes T2
//                                                                   [2 reads + 2 writes]
//   k_row_inv  reads T2 rows, DCT-III along the row, writes U (solver.py:208)
//                                                                   [1 read + 1 write]
// = the 8 full-array transfers per timestep of SURVEY.md section 8(d).
//
// HBM layouts.  U is row-major (it is the boundary's array).  T1/T2 are "column-tile
// major": element (r, c) of PHYSICAL column c lives at ((c / CT) * N + r) * CT + (c % CT) with CT
// columns per tile, so a column tile is one contiguous slab (k_col streams it with 16-byte
// coalesced accesses through an LDS stage) while the row kernels touch it in CT*sizeof(T)-byte
// pieces, consecutive rows completing each 128-byte line.
// fp64: physical column = coefficient index.  fp32 (FCfg::SLOT): the physical columns are in SLOT ORDER: column
// 4 s + t holds coefficient t of recombination slot s -- the four coefficients {kk, N-kk, M-kk, M+kk} a lane of a
// row kernel produces (or consumes) together are one contiguous 16-byte piece, ONE access instead of four scattered
// 4-byte ones (row_access; col_coef maps a physical column back to its coefficient for k_col's spectral constants).
// hat_U lives in k_col's native order: column kc at kc*N, inside it the lane's positions in PAIRS -- positions
// (2j, 2j+1) of lane l are the two components of the value pair number j*G + l (hat_pair_index): a recombination
// slot's four coefficients are two 16-byte (fp64) / 8-byte (fp32) accesses per lane, contiguous across the
// lanes of a group -- nobody else reads it.
#pragma once
#include <cmath>
#include <vector>
#include <cstdlib>

#include "chs_common.h"
#ifdef CHS_STAMPS
#define CHS_NSTAMP 12
#ifdef CHS_FAST_MAIN_TU
__device__ unsigned long long g_stamps[2][8192 * CHS_NSTAMP];
#endif
#define STAMP(K, I)                                                                         \
  do {                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    unsigned long long t__;                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps[K][blockIdx.x * CHS_NSTAMP + (I)] = t__; \
  } while (0)
// recombination-internal stamps of k_col<MODE_STEP> (the only PIPE user), slots 6..11
#define CHS_RSTAMP(I) do { if constexpr (PIPE) STAMP(1, 6 + (I)); } while (0)
// the same from wave 1 (the plain path), slots 10..11 + reuse: diagnostic builds only
#define CHS_RSTAMP1(I)                                                                      \
  do {                                                                                      \
    if constexpr (PIPE) {                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                    \
      unsigned long long t__;                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");          \
      __builtin_amdgcn_sched_barrier(0);                                                    \
      if (threadIdx.x == 64 && blockIdx.x < 8192) g_stamps[0][blockIdx.x * CHS_NSTAMP + 7 + (I)] = t__; \
    }                                                                                       \
  } while (0)
#ifdef CHS_FAST_MAIN_TU
extern "C" int chs_debug_stamps(int which, unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n,
                                  sizeof(unsigned long long) * 8192 * CHS_NSTAMP * which);
}
#endif
#else
#define STAMP(K, I) do {} while (0)
#define CHS_RSTAMP(I) do {} while (0)
#define CHS_RSTAMP1(I) do {} while (0)
#endif
#include "chs_fast_core.h"
#include "chs_tail.h"
#include "chs_math.h"

enum { MODE_STEP = 0, MODE_FWD_NATIVE = 1, MODE_FWD_NATURAL = 2, MODE_INV_NATURAL = 3, MODE_INV_NATIVE = 4 };

// Accesses to the tile-major arrays go through a 32-bit BYTE offset from the (uniform) array base:
// `global_load/store v, voffset, s[base]` -- one or two integer instructions per address instead of a 64-bit
// multiply-add chain.  N*N*sizeof(T) < 2^32 for every configuration (<= 512 MB).
// Byte offset of the 4-element piece of recombination slot (q, k) of lane l in row r: the slot's number is
// s = k * S2/2 + kap with kap = l + G q in [0, S2/2) (the special lane's own slot is s = 0: kap = 0, k = 0),
// CT/4 slots share a tile.  pbase = (q*R2 + k)*4 is what recombine() hands its callbacks.
template <class C>
__device__ __forceinline__ unsigned slot_boff(unsigned r, int pbase, int l) {
  static_assert(C::CT % 4 == 0, "a tile holds whole slots");
  constexpr unsigned SPT = C::CT / 4;
  const unsigned s = (unsigned)(((pbase / 4) % C::R2) * (C::S2 / 2) + C::G * ((pbase / 4) / C::R2)) + (unsigned)l;
  return (((s / SPT) * C::N + r) * C::CT + (s % SPT) * 4) * (unsigned)sizeof(typename C::T);
}
// element (r, k) of a tile-major array in the natural column order
template <class C>
__device__ __forceinline__ unsigned tile_boff(unsigned r, unsigned k) {
  return (((k / C::CT) * C::N + r) * C::CT + (k % C::CT)) * (unsigned)sizeof(typename C::T);
}
// coefficient index held by physical column c (SLOT: the inverse of the slot order of slot_boff)
template <class C>
__device__ __forceinline__ int col_coef(int c) {
  if constexpr (!C::SLOT) return c;
  constexpr int H = C::R2 / 2, HS = C::S2 / 2;
  const int s = c >> 2, t = c & 3;
  const int k = s / HS, kap = s % HS;
  if (kap == 0 && k == 0) return t * (C::M / 2);                      // (X[0], X[M/2], X[M], X[3M/2])
  const int kk = (kap == 0) ? ((k < H) ? C::S2 * k : HS + C::S2 * (k - H)) : kap + C::S2 * k;
  return t == 0 ? kk : (t == 1 ? C::N - kk : (t == 2 ? C::M - kk : C::M + kk));
}
template <typename T>
__device__ __forceinline__ const T* at_boff(const T* base, unsigned boff);
template <typename T>
__device__ __forceinline__ T* at_boff(T* base, unsigned boff);
template <typename T>
__device__ __forceinline__ void load4(const T* p, T q[4]);
template <typename T>
__device__ __forceinline__ void store4(T* p, const T q[4]);
// the four coefficients y[t] <-> columns idx[t] of row r of a tile-major array: slot (pbase, l) of a row kernel
template <typename T>
__device__ __forceinline__ void store4_nt(T* p, const T q[4]);
template <class C, bool NT = false>
__device__ __forceinline__ void row_store(typename C::T* base, unsigned r, int pbase, int l, const int* idx, const typename C::T y[4]) {
  if constexpr (C::SLOT) {
    if constexpr (NT) store4_nt<typename C::T>(at_boff(base, slot_boff<C>(r, pbase, l)), y);
    else store4<typename C::T>(at_boff(base, slot_boff<C>(r, pbase, l)), y);
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if constexpr (NT) __builtin_nontemporal_store(y[t], at_boff(base, tile_boff<C>(r, idx[t])));
      else *at_boff(base, tile_boff<C>(r, idx[t])) = y[t];
    }
  }
}
template <class C>
__device__ __forceinline__ void row_load(const typename C::T* base, unsigned r, int pbase, int l, const int* idx, typename C::T y[4]) {
  if constexpr (C::SLOT) {
    load4<typename C::T>(at_boff(base, slot_boff<C>(r, pbase, l)), y);
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) y[t] = *at_boff(base, tile_boff<C>(r, idx[t]));
  }
}
template <typename T>
__device__ __forceinline__ const T* at_boff(const T* base, unsigned boff) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + boff);
}
template <typename T>
__device__ __forceinline__ T* at_boff(T* base, unsigned boff) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + boff);
}

// First row of a row-kernel workgroup.  CT consecutive rows share 128-byte lines of T1/T2; when
// a workgroup holds fewer rows (C < CT) the CT/C workgroups of one line group are given block
// numbers b, b+8, b+16, ... : workgroups are dealt round-robin over the 8 XCDs, so these land on
// one XCD (one L2) and are dispatched back to back -- a speed matter only, never correctness.
template <class C>
__device__ __forceinline__ int row_of_block(int b) {
  constexpr int Q = C::CT / C::C;
  if constexpr (Q == 1) {
    return b * C::C;
  } else {
    const int xcd = b & 7, j = b >> 3;
    const int g = xcd + 8 * (j / Q), m = j % Q;
    return g * C::CT + m * C::C;
  }
}

// Lanes of a row workgroup.  Two rows per workgroup, transforms of whole wavefronts (N >= 4096): every wavefront
// takes 32 lanes of EACH row (lane i the first row, lane i+32 the second, same butterfly index), so that a
// wave-load of the tile-major operand touches the two rows' neighbouring pieces -- 64 contiguous bytes per tile
// instead of 32 (tools/ubench/strided_rows.hip: the access pattern alone 84 -> 40 us at N=4096 fp64).  Measured on
// the kernels: N=8192 fp64 row 481 -> 435 us (+3 % steps/s), fp32 332 -> 322 us; N=4096 fp64 row -2 % but k_col
// behind it +4 % (net -0.8 %), so it is on from CHS_ROW_INTERLEAVE_MIN_N upwards only.
#ifndef CHS_ROW_INTERLEAVE_MIN_N
#define CHS_ROW_INTERLEAVE_MIN_N 8192
#endif
// `wave` = chs_wave_id() (an SGPR), the lane number is taken afresh (chs_lane_id): neither threadIdx.x nor l / sub need to
// stay in registers between the phases of a row kernel -- they are recomputed by calling this again.  Groups of whole
// wavefronts without the interleave: sub is wave-uniform (an SGPR, and with it the row number and the scratch base).
template <class C>
__device__ __forceinline__ void row_lane_map(int wave, int& l, int& sub) {
  const int ln = chs_lane_id();
  if constexpr (C::C == 2 && (C::G % 64 == 0) && C::N >= CHS_ROW_INTERLEAVE_MIN_N) {
    sub = ln >> 5;
    l = wave * 32 + (ln & 31);
  } else if constexpr (C::G % 64 == 0) {
    constexpr int WG = C::G / 64;   // wavefronts per transform
    sub = wave / WG;
    l = (wave % WG) * 64 + ln;
  } else {
    const int tid = wave * 64 + ln;
    l = tid % C::G;
    sub = tid / C::G;
  }
}
__device__ __forceinline__ int launder(int x);
// FRESH (fp64, whose row kernels sit at their 128 registers and spilled these indices): lane index and row within the
// workgroup are recomputed per phase.  Otherwise (fp32: room to spare, the recomputation only adds instructions -- N=8192
// fp32 measured 1.1 % slower with it) the values taken at the top of the kernel are passed through an opaque copy.
#ifndef CHS_FRESH_MIN_E
#define CHS_FRESH_MIN_E 16   // (8 values per lane, N <= 1024: registers to spare; N=1024 measured 2.9 % slower with it)
#endif
template <class C>
struct FreshLane { static constexpr bool value = (sizeof(typename C::T) == 8) && (C::E >= CHS_FRESH_MIN_E); };
template <class C, bool FORCE = false>
__device__ __forceinline__ int row_l(int wave, int l_top) {
  if constexpr (FORCE || FreshLane<C>::value) { int l, sub; row_lane_map<C>(wave, l, sub); return l; }
  else return launder(l_top);
}
template <class C, bool FORCE = false>
__device__ __forceinline__ int row_sub(int wave, int sub_top) {
  if constexpr (FORCE || FreshLane<C>::value) { int l, sub; row_lane_map<C>(wave, l, sub); return sub; }
  else return launder(sub_top);
}

template <typename T>
__device__ __forceinline__ void load4(const T* p, T q[4]) {
  if constexpr (sizeof(T) == 8) {
    const double2 a = *reinterpret_cast<const double2*>(p);
    const double2 b = *reinterpret_cast<const double2*>(p + 2);
    q[0] = a.x; q[1] = a.y; q[2] = b.x; q[3] = b.y;
  } else {
    const float4 a = *reinterpret_cast<const float4*>(p);
    q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w;
  }
}
// U at the borders of a call is streamed once (read by the entry kernel, written by the last row kernel): non-temporal
// accesses, so that it does not displace T and hat_U from the caches the step loop lives in (the driver's 20-step
// literal call: 4368 -> 4450 steps/s, profiles/r03_ab_nt.txt)
typedef double chs_d2v __attribute__((ext_vector_type(2)));
typedef float chs_f4v __attribute__((ext_vector_type(4)));
typedef float chs_f2v __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ void load4_nt(const T* p, T q[4]) {
  if constexpr (sizeof(T) == 8) {
    const chs_d2v a = __builtin_nontemporal_load(reinterpret_cast<const chs_d2v*>(p));
    const chs_d2v b = __builtin_nontemporal_load(reinterpret_cast<const chs_d2v*>(p + 2));
    q[0] = a.x; q[1] = a.y; q[2] = b.x; q[3] = b.y;
  } else {
    const chs_f4v a = __builtin_nontemporal_load(reinterpret_cast<const chs_f4v*>(p));
    q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w;
  }
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const T q[4]);
template <typename T>
__device__ __forceinline__ void store4_nt(T* p, const T q[4]) {
  if constexpr (sizeof(T) == 8) {
    chs_d2v a, b; a.x = q[0]; a.y = q[1]; b.x = q[2]; b.y = q[3];
    __builtin_nontemporal_store(a, reinterpret_cast<chs_d2v*>(p));
    __builtin_nontemporal_store(b, reinterpret_cast<chs_d2v*>(p + 2));
  } else {
    chs_f4v a; a.x = q[0]; a.y = q[1]; a.z = q[2]; a.w = q[3];
    __builtin_nontemporal_store(a, reinterpret_cast<chs_f4v*>(p));
  }
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const T q[4]) {
  if constexpr (sizeof(T) == 8) {
    *reinterpret_cast<double2*>(p) = make_double2(q[0], q[1]);
    *reinterpret_cast<double2*>(p + 2) = make_double2(q[2], q[3]);
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(q[0], q[1], q[2], q[3]);
  }
}

// index (in value pairs, from the column's start) of positions (pbase, pbase+1) of lane l; (pbase+2, pbase+3)
// follow G pairs later
template <class C>
__device__ __forceinline__ int hat_pair_index(int pbase, int l) {
  return (pbase / 2) * C::G + l;
}

extern __shared__ __attribute__((aligned(16))) unsigned char chs_dyn_lds[];

// Build knobs that remain (everything else that was tried is recorded with its measurement in DESIGN.md section 7):
#ifndef CHS_ALIAS_T
#define CHS_ALIAS_T 1     // T2 overwrites T1 in place (one array less in the per-step working set)
#endif
#ifndef CHS_COL_ZIGZAG
#define CHS_COL_ZIGZAG 1  // k_col walks the tiles in alternating direction from step to step (Infinity Cache)
#endif

// The fused row kernel takes its pass twiddles from L2 (its four workgroups per CU leave no LDS) unless a configuration
// has room for the compact form: the k = 1 entries of the pass-0 table + the middle-pass tables, the other pass-0
// twiddles as powers (tw0_load<POW>)
// (value 2), for the middle-pass tables alone (value 3; the pass-0 table stays in L2), or for all of them (value 1)
template <class C>
struct RowTwLds { static constexpr int value = 0; };
template <class C>
constexpr int row_tw_lds_elems() {
  constexpr int mid = 2 * ((C::RA > 1 ? (C::RA - 1) * C::L2 : 0) + (C::RB > 1 ? (C::RB - 1) * C::L3 : 0));
  constexpr int v = RowTwLds<C>::value;
  return v == 1 ? 2 * (C::R0 - 1) * C::L1 + mid : (v == 2 ? 2 * C::L1 + mid : (v == 3 ? mid : 0));
}

// The row kernels' LDS copy of their pass twiddles (RowTwLds), behind the exchange scratch and the log table; returns the
// table set to use.  The caller provides the barrier between the copy and the first use.
template <class C>
__device__ __forceinline__ FTables<typename C::T> row_twiddles_to_lds(const FTables<typename C::T>& tb, int tid) {
  using T = typename C::T;
  constexpr int RTWM = RowTwLds<C>::value;
  FTables<T> tbp = tb;
  if constexpr (RTWM != 0) {
    T* ltw = reinterpret_cast<T*>(chs_dyn_lds + (size_t)C::C * C::SCR * sizeof(T) + CHS_LOGTAB_N * 16);
    constexpr int N0 = (RTWM == 2) ? 2 * C::L1 : (RTWM == 1 ? 2 * (C::R0 - 1) * C::L1 : 0), NM = row_tw_lds_elems<C>() - N0;
    auto cp = [&](T* dst, const T* src, int n) {
      for (int i = 2 * tid; i < n; i += 2 * C::THREADS) {
        if constexpr (sizeof(T) == 8) *reinterpret_cast<double2*>(dst + i) = *reinterpret_cast<const double2*>(src + i);
        else *reinterpret_cast<v2f*>(dst + i) = *reinterpret_cast<const v2f*>(src + i);
      }
    };
    cp(ltw, tb.tw0, N0);
    cp(ltw + N0, tb.twa, NM);
    if constexpr (N0 != 0) tbp.tw0 = ltw;
    tbp.twa = ltw + N0;
    tbp.twb = tbp.twa + (C::RA > 1 ? 2 * (C::RA - 1) * C::L2 : 0);
  }
  return tbp;
}

// The same copy in two halves -- `request` at the very top of a kernel (loads into registers), `store` behind the first
// loads of the kernel's own first phase: as one load/store loop the copy was an L2 round trip during which nothing else of
// the workgroup was in flight (the log table of the fused row kernel likewise: two round trips in a row, ~1.5 K cycles).
// One iteration per thread wherever the tables fit 2 * THREADS elements per part (N >= 4096); the kernels fall back to the
// loop otherwise (RowTwSplit).
template <class C>
struct RowTwSplit {
  static constexpr int RTWM = RowTwLds<C>::value;
  static constexpr int N0 = (RTWM == 2) ? 2 * C::L1 : (RTWM == 1 ? 2 * (C::R0 - 1) * C::L1 : 0);
  static constexpr int NM = row_tw_lds_elems<C>() - N0;
  static constexpr int I0 = (N0 + 2 * C::THREADS - 1) / (2 * C::THREADS), IM = (NM + 2 * C::THREADS - 1) / (2 * C::THREADS);
  static constexpr bool value = (RTWM != 0) && (I0 + IM <= 2) && (C::THREADS >= CHS_LOGTAB_N);
  using Pair = typename std::conditional<sizeof(typename C::T) == 8, double2, v2f>::type;
  Pair a[I0 > 0 ? I0 : 1], b[IM > 0 ? IM : 1];
  double2 lt;   // the log-table entry of this thread (fp64 kernels with a pointwise part)
};
template <class C, bool LOGTAB>
__device__ __forceinline__ RowTwSplit<C> row_tables_request(const FTables<typename C::T>& tb, int tid) {
  using S = RowTwSplit<C>;
  using P = typename S::Pair;
  RowTwSplit<C> r;   // (every member written on every path: the structure stays in registers)
  r.a[0] = P{}; r.b[0] = P{}; r.lt = make_double2(0.0, 0.0);
#pragma unroll
  for (int i = 0; i < S::I0; ++i) {
    const int e = 2 * (tid + i * C::THREADS);
    P v = P{};
    if (e < S::N0) v = *reinterpret_cast<const P*>(tb.tw0 + e);
    r.a[i] = v;
  }
#pragma unroll
  for (int i = 0; i < S::IM; ++i) {
    const int e = 2 * (tid + i * C::THREADS);
    P v = P{};
    if (e < S::NM) v = *reinterpret_cast<const P*>(tb.twa + e);
    r.b[i] = v;
  }
  if constexpr (LOGTAB) {
    if (tid < CHS_LOGTAB_N) r.lt = reinterpret_cast<const double2*>(chs_log_table)[tid];
  }
  return r;
}
template <class C, bool LOGTAB>
__device__ __forceinline__ void row_tables_store(int tid, const RowTwSplit<C>& r) {
  using S = RowTwSplit<C>;
  using T = typename C::T;
  T* ltw = reinterpret_cast<T*>(chs_dyn_lds + (size_t)C::C * C::SCR * sizeof(T) + CHS_LOGTAB_N * 16);
#pragma unroll
  for (int i = 0; i < S::I0; ++i) {
    const int e = 2 * (tid + i * C::THREADS);
    if (e < S::N0) *reinterpret_cast<typename S::Pair*>(ltw + e) = r.a[i];
  }
#pragma unroll
  for (int i = 0; i < S::IM; ++i) {
    const int e = 2 * (tid + i * C::THREADS);
    if (e < S::NM) *reinterpret_cast<typename S::Pair*>(ltw + S::N0 + e) = r.b[i];
  }
  if constexpr (LOGTAB) {
    double2* ltab = reinterpret_cast<double2*>(chs_dyn_lds + (size_t)C::C * C::SCR * sizeof(T));
    if (tid < CHS_LOGTAB_N) ltab[tid] = r.lt;
  }
}
template <class C>
__device__ __forceinline__ FTables<typename C::T> row_tables_in_lds(const FTables<typename C::T>& tb) {
  using S = RowTwSplit<C>;
  using T = typename C::T;
  FTables<T> tbp = tb;
  T* ltw = reinterpret_cast<T*>(chs_dyn_lds + (size_t)C::C * C::SCR * sizeof(T) + CHS_LOGTAB_N * 16);
  if constexpr (S::N0 != 0) tbp.tw0 = ltw;
  tbp.twa = ltw + S::N0;
  tbp.twb = tbp.twa + (C::RA > 1 ? 2 * (C::RA - 1) * C::L2 : 0);
  return tbp;
}

// Returns x through an opaque asm so that index arithmetic derived from it is not CSE'd
// with (and kept alive since) an earlier phase of the kernel: recomputing a few integer
// offsets is far cheaper than holding dozens of address registers across a phase.
__device__ __forceinline__ int launder(int x) {
  asm volatile("" : "+v"(x));
  // lane / thread indices only: without the range the divisions and remainders by powers of two that
  // the index maps are made of compile to signed sequences (4-5 instructions instead of one shift)
  __builtin_assume(x >= 0 && x < 1024);
  return x;
}

// ---------------------------------------------------------------------------
// k_row_fwd: one group per row.  POINTWISE: the operand is EnergieEut(U) and the
// block's sum(mu^2) is recorded (solver.py:225); otherwise a plain transform.
// (Prologue of a solve_or_resume call, and the unfused/jitter path.)
// ---------------------------------------------------------------------------
// STREAM (the row half of hat_U = dctn(U) at the entry of a call that finds the first step's operand already on the
// device, chs_fast_enter_hat): U is read for the last time and the result is read once, by k_col<FWD_NATIVE> right behind
// this kernel -- non-temporal stores, as k_row_fwd2's.
template <class C, bool POINTWISE, bool STREAM = false>
__global__ __launch_bounds__(C::THREADS, C::WPS) void k_row_fwd(const typename C::T* __restrict__ U, typename C::T* __restrict__ T1,
                                                    FTables<typename C::T> tb, DevConsts dc,
                                                    const DevState* __restrict__ st, double* __restrict__ partMu) {
  using T = typename C::T;
  __shared__ double red[32];
  if (st->halt) return;
  T* lds = reinterpret_cast<T*>(chs_dyn_lds);
  const int wv = chs_wave_id();   // (threadIdx.x is not kept: chs_common.h)
  int l, sub;
  row_lane_map<C>(wv, l, sub);
  const int row = row_of_block<C>(blockIdx.x) + sub;
  T* scr = lds + (size_t)sub * C::SCR;
  constexpr int RTWM = RowTwLds<C>::value;
  const FTables<T> tbp = row_twiddles_to_lds<C>(tb, wv * 64 + chs_lane_id());
  if constexpr (RTWM != 0) __syncthreads();
  typename C::V z[C::E];
  double s2 = 0.0;
  const T RT = (T)dc.RT, BRT = (T)dc.BRT, A0 = (T)dc.A0, A1 = (T)dc.A1;
  const unsigned urow = (unsigned)row * C::N;  // (32-bit offsets from the uniform base, see slot_boff)
#pragma unroll
  for (int q = 0; q < C::NP0; ++q) {
    const int m1 = l + C::G * q, m2 = C::L1 - 1 - m1;
#pragma unroll
    for (int j = 0; j < C::R0 / 2; ++j) {
      T q1[4], q2[4];
      // (ordinary loads also with STREAM: non-temporal loads that miss every cache took this kernel from ~75 to 188 us at
      // N=4096 fp64 -- k_row_fwd2's non-temporal second read of U hits L2)
      load4<T>(at_boff(U, (urow + 4u * (unsigned)(m1 + C::L1 * j)) * (unsigned)sizeof(T)), q1);
      load4<T>(at_boff(U, (urow + 4u * (unsigned)(m2 + C::L1 * j)) * (unsigned)sizeof(T)), q2);
      pack_quads<C>(q1, q2, q, j, z);
    }
  }
  if constexpr (POINTWISE) {
    // EnergieEut in the shared-log form of the fused row kernel (log U - log(1-U) from the table-driven log:
    // no division, no spills at four waves per SIMD; the division-based chs_mu needed 204 bytes of scratch here)
    double2* ltab = reinterpret_cast<double2*>(chs_dyn_lds + (size_t)C::C * C::SCR * sizeof(T));
    if constexpr (sizeof(T) == 8) {
      for (int t = wv * 64 + chs_lane_id(); t < CHS_LOGTAB_N; t += C::THREADS) ltab[t] = reinterpret_cast<const double2*>(chs_log_table)[t];
      __syncthreads();
    }
    unsigned dom = 0;
    auto mu = [&](T& u) {
      const T uinv = T(1) - u;
      const T lU = chs_log_unit_tab<T>(u, ltab, dom), lV = chs_log_unit_tab<T>(uinv, ltab, dom);
      const T m = chs_mu_from_logs_fast<T>(u, uinv, lU, lV, RT, BRT, A0, A1);
      s2 += (double)m * (double)m;
      u = m;
      asm volatile("" : "+v"(u), "+v"(s2), "+v"(dom));  // one grid point at a time (register pressure)
    };
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
      T a = cx_re(z[e]), b = cx_im(z[e]);
      mu(a); mu(b);
      z[e] = cx_make(a, b);
      __builtin_amdgcn_sched_barrier(0);   // (as in the fused row kernel: one value pair at a time)
    }
    if (dom > (unsigned)(CHS_LOGTAB_N - 1)) s2 = __builtin_nan("");  // U left (0,1): L2 of this step becomes NaN
    // (the wavefront sums go to LDS here: no register carries the sum across the transform)
    const double acc1[1] = {s2};
    block_reduce_begin<1, FreshLane<C>::value>(acc1, red, wv);
  }
  fwd_passes<C, (RTWM == 2)>(z, scr, tbp, row_l<C>(wv, l));
  // (lane index and row of the last phase: taken afresh here, laundered per slot -- address arithmetic is redone at the
  // point of use instead of being computed for all slots up front and kept in registers)
  const int lr = row_l<C>(wv, l), sr = row_sub<C>(wv, sub);
  const int rw = row_of_block<C>(blockIdx.x) + sr;
  recombine<C, true, false, false>(z, tb, lr, [](int, const int*) { return NoFetch{}; },
                            [&](int pbase, const int* idx, Cx<T>& Ya, Cx<T>& Yb, bool live, NoFetch) {
    if (live) {
      const T y[4] = {cx_re(Ya), cx_im(Ya), cx_re(Yb), cx_im(Yb)};
      row_store<C, STREAM>(T1, rw, pbase, launder(lr), idx, y);
    }
  }, [](int, const int*, Cx<T>&, Cx<T>&, bool) {});
  if constexpr (POINTWISE) {
    double tot[1];
    if (block_reduce_end<1, C::THREADS / 64>(red, tot, wv)) partMu[blockIdx.x] = tot[0];
  }
}

// ---------------------------------------------------------------------------
// k_row_fwd2: the entry of a solve_or_resume call on the fused pipeline in ONE launch:
//   Ta <- row DCT-II of U            (the row half of hat_U = dctn(U), solver.py:159)
//   T1 <- row DCT-II of EnergieEut(U) (what the fused row kernel of a previous step would have left)
// and the block's sum(mu^2) (solver.py:225).  The two transforms run one after the other on the same
// registers (four waves per SIMD like the other row kernels); the second pass re-reads the row of U
// the workgroup has just read (L2), so HBM sees U once.  EnergieEut uses the shared-log form of the
// fused row kernel (log U - log(1-U) from the table-driven log).
// ---------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(C::THREADS, C::WPS) void k_row_fwd2(const typename C::T* __restrict__ U, typename C::T* __restrict__ Ta,
                                                                 typename C::T* __restrict__ T1, FTables<typename C::T> tb,
                                                                 DevConsts dc, const DevState* __restrict__ st,
                                                                 double* __restrict__ partMu) {
  using T = typename C::T;
  __shared__ double red[32];
  if (st->halt) return;
  T* lds = reinterpret_cast<T*>(chs_dyn_lds);
  const int wv = chs_wave_id();   // (threadIdx.x is not kept: chs_common.h; lane index, row and scratch base are taken
                                  // afresh in every phase instead of living -- or being spilled -- across the passes)
  int l, sub;
  row_lane_map<C>(wv, l, sub);
  double2* ltab = reinterpret_cast<double2*>(chs_dyn_lds + (size_t)C::C * C::SCR * sizeof(T));
  if constexpr (sizeof(T) == 8) {
    for (int t = wv * 64 + chs_lane_id(); t < CHS_LOGTAB_N; t += C::THREADS) ltab[t] = reinterpret_cast<const double2*>(chs_log_table)[t];
  }
  // pass twiddles from LDS where the configuration has room (RowTwLds, as in the fused row kernel: at N=8192 fp32,
  // 30 twiddle loads per radix-16 butterfly, this kernel took 511 us with them in L2)
  constexpr int RTWM = RowTwLds<C>::value;
  constexpr bool RTW = (RTWM == 2);
  const FTables<T> tbp = row_twiddles_to_lds<C>(tb, wv * 64 + chs_lane_id());
  if constexpr (sizeof(T) == 8 || RTWM != 0) __syncthreads();  // the log table and the twiddles are visible
  typename C::V z[C::E];
  const T RT = (T)dc.RT, BRT = (T)dc.BRT, A0 = (T)dc.A0, A1 = (T)dc.A1;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int lp = row_l<C, true>(wv, l);
    const unsigned urow = (unsigned)(row_of_block<C>(blockIdx.x) + row_sub<C, true>(wv, sub)) * C::N;
#pragma unroll
    for (int q = 0; q < C::NP0; ++q) {
      const int m1 = lp + C::G * q, m2 = C::L1 - 1 - m1;
#pragma unroll
      for (int j = 0; j < C::R0 / 2; ++j) {
        T q1[4], q2[4];
        if (pass == 1) {  // the last read of this row
          load4_nt<T>(at_boff(U, (urow + 4u * (unsigned)(m1 + C::L1 * j)) * (unsigned)sizeof(T)), q1);
          load4_nt<T>(at_boff(U, (urow + 4u * (unsigned)(m2 + C::L1 * j)) * (unsigned)sizeof(T)), q2);
        } else {
          load4<T>(at_boff(U, (urow + 4u * (unsigned)(m1 + C::L1 * j)) * (unsigned)sizeof(T)), q1);
          load4<T>(at_boff(U, (urow + 4u * (unsigned)(m2 + C::L1 * j)) * (unsigned)sizeof(T)), q2);
        }
        pack_quads<C>(q1, q2, q, j, z);
      }
    }
    if (pass == 1) {
      // (the sum lives in this branch only: its wavefront sums go to LDS in front of the transform, so that no register
      // carries it through the passes -- or, where the two passes stay a rolled loop, around the loop)
      double s2 = 0.0;
      unsigned dom = 0;
      auto mu = [&](T& u) {
        const T uinv = T(1) - u;
        const T lU = chs_log_unit_tab<T>(u, ltab, dom), lV = chs_log_unit_tab<T>(uinv, ltab, dom);
        const T m = chs_mu_from_logs_fast<T>(u, uinv, lU, lV, RT, BRT, A0, A1);
        s2 += (double)m * (double)m;
        u = m;
        asm volatile("" : "+v"(u), "+v"(s2), "+v"(dom));  // one grid point at a time (register pressure)
      };
#pragma unroll
      for (int e = 0; e < C::E; ++e) {
        T a = cx_re(z[e]), b = cx_im(z[e]);
        mu(a); mu(b);
        z[e] = cx_make(a, b);
      }
      if (dom > (unsigned)(CHS_LOGTAB_N - 1)) s2 = __builtin_nan("");  // U left (0,1): L2 of the first step becomes NaN
      const double acc1[1] = {s2};
      block_reduce_begin<1, true>(acc1, red, wv);
    }
    T* dst = pass ? T1 : Ta;
    {
      const int lf = row_l<C, true>(wv, l), sf = row_sub<C, true>(wv, sub);
      fwd_passes<C, RTW>(z, lds + (size_t)sf * C::SCR, tbp, lf);
    }
    const int lr = row_l<C, true>(wv, l), sr = row_sub<C, true>(wv, sub);
    const int rw = row_of_block<C>(blockIdx.x) + sr;
    recombine<C, true, false, false>(z, tb, lr, [](int, const int*) { return NoFetch{}; },
                              [&](int pbase, const int* idx, Cx<T>& Ya, Cx<T>& Yb, bool live, NoFetch) {
      if (live) {
        const T y[4] = {cx_re(Ya), cx_im(Ya), cx_re(Yb), cx_im(Yb)};
        // Ta is read once, by k_col<FWD_NATIVE> right behind this kernel: streamed like U (driver protocol 4471 -> 4533
        // steps/s, profiles/r03_ab_nt.txt); T1 is the first step's operand and stays cached
#ifndef CHS_FWD2_TA_NT
#define CHS_FWD2_TA_NT 1
#endif
        if (pass == 0) row_store<C, (CHS_FWD2_TA_NT != 0)>(dst, rw, pbase, launder(lr), idx, y);
        else row_store<C>(dst, rw, pbase, launder(lr), idx, y);
      }
    }, [](int, const int*, Cx<T>&, Cx<T>&, bool) {});
    __builtin_amdgcn_sched_barrier(0);
  }
  double tot[1];
  if (block_reduce_end<1, C::THREADS / 64>(red, tot, wv)) partMu[blockIdx.x] = tot[0];
}

// ---------------------------------------------------------------------------
// k_row_inv: one group per row: T2 (tile-major) -> DCT-III -> U (row-major, solver.py:208).
//   DIAG: while U is in registers, the pointwise part of the record of this step
//         (solver.py:218-228): bulk energy density, |U - mean| and the U < threshold
//         count, plus the column-edge terms of the gradient energy (the rest of E2
//         comes from the spectrum, see k_col).  partDiag[block] = {sE, sEdge, sPS, cSA}.
//   FUSE: the row then goes straight on to the next timestep: EnergieEut (solver.py:
//         166-175, sharing log U and log(1-U) with the energy density), sum(mu^2), and
//         the forward row DCT-II into T1 -- U is never re-read from HBM.
// ---------------------------------------------------------------------------
template <class C, bool DIAG, bool FUSE, bool ADAPT = false>
__global__ __launch_bounds__(C::THREADS, C::WPS) void k_row_inv(const typename C::T* __restrict__ T2, typename C::T* __restrict__ U,
                                                    typename C::T* __restrict__ T1, FTables<typename C::T> tb,
                                                    DevConsts dc, const DevState* __restrict__ st,
                                                    double* __restrict__ partDiag, double* __restrict__ partMu,
                                                    double* __restrict__ partRa, int store_u,
                                                    typename C::T* __restrict__ partColRows = nullptr) {
  using T = typename C::T;
  __shared__ double red[64];
  if (st->halt) return;
  T* lds = reinterpret_cast<T*>(chs_dyn_lds);
  const int wv = chs_wave_id();   // (threadIdx.x is not kept across the kernel: chs_common.h)
  int l, sub;
  row_lane_map<C>(wv, l, sub);
  const double mean_u = st->meanU;  // requested at entry (k_col of this step wrote it), used in the pointwise part
  // reduction table of the table-driven log, behind the exchange scratch (visible after the first barrier)
  double2* ltab = reinterpret_cast<double2*>(chs_dyn_lds + (size_t)C::C * C::SCR * sizeof(T));
#ifndef CHS_ROW_TABLES_SPLIT
#define CHS_ROW_TABLES_SPLIT 1
#endif
  // (SPLIT: the table copies are requested here and stored behind the first phase's loads, below)
  // (measured: N=8192 fp64 +0.9 %, N=4096 fp64 equal, N=4096 fp32 -0.8 %, N=8192 fp32 -5 %: four row workgroups of a CU
  // cover each other's round trips, and the registers held across the first phase cost the fp32 kernels more)
  constexpr bool SPLIT = (CHS_ROW_TABLES_SPLIT != 0) && RowTwSplit<C>::value && !C::WAVE_LOCAL && sizeof(T) == 8 && C::N >= 8192;
  constexpr bool LOGT = DIAG && sizeof(T) == 8;
  [[maybe_unused]] RowTwSplit<C> tabs;
  if constexpr (SPLIT) {
    tabs = row_tables_request<C, LOGT>(tb, wv * 64 + chs_lane_id());
  } else if constexpr (LOGT) {
    for (int t = wv * 64 + chs_lane_id(); t < CHS_LOGTAB_N; t += C::THREADS) ltab[t] = reinterpret_cast<const double2*>(chs_log_table)[t];
  }
  // pass twiddles from LDS where the configuration has room (RowTwLds): visible behind the first exchange barrier of
  // the inverse passes, whose last-pass butterflies come first and need none
  constexpr int RTWM = RowTwLds<C>::value;
  constexpr bool RTW = (RTWM == 2);   // pass-0 twiddles by powers
  FTables<T> tbp_ = tb;
  if constexpr (SPLIT) tbp_ = row_tables_in_lds<C>(tb);
  else tbp_ = row_twiddles_to_lds<C>(tb, wv * 64 + chs_lane_id());
  const FTables<T> tbp = tbp_;
  // (groups inside one wavefront exchange behind wavefront fences only: no block barrier would make the log table and
  // the twiddles visible before their first use)
  if constexpr (C::WAVE_LOCAL && ((DIAG && sizeof(T) == 8) || RTWM != 0)) __syncthreads();
  const int row0 = row_of_block<C>(blockIdx.x);
  const int row = row0 + sub;
  T* scr = lds + (size_t)sub * C::SCR;
  typename C::V z[C::E];
  if constexpr (DIAG && FUSE) STAMP(0, 0);
  // (the loads are the slot's fetch: issued in the slot on the large grids, all up front on the small ones -- PreAll)
  struct RowQuad { T y[4]; };
  // (!FUSE: the lane index goes in through an opaque copy -- the once-per-call kernels otherwise keep a multiple of it
  // alive from the log-table copy at the top and spill it; the fused kernel's code is left as it is)
#ifndef CHS_ROW_IN_PIPE
#define CHS_ROW_IN_PIPE 0
#endif
  recombine<C, false, true, (CHS_ROW_IN_PIPE != 0)>(z, tb, FUSE ? l : fc_opaque(l), [&](int pbase, const int* idx) {
    RowQuad p;
    row_load<C>(T2, row, pbase, l, idx, p.y);
    return p;
  }, [&](int, const int*, Cx<T>& Ya, Cx<T>& Yb, bool, const RowQuad& p) {
    Ya = cx_make(p.y[0], p.y[1]); Yb = cx_make(p.y[2], p.y[3]);
  }, [](int, const int*, Cx<T>&, Cx<T>&, bool) {});
  if constexpr (DIAG && FUSE) STAMP(0, 1);
  if constexpr (SPLIT) row_tables_store<C, LOGT>(wv * 64 + chs_lane_id(), tabs);   // (visible behind the first exchange barrier)
  inv_passes<C, RTW>(z, scr, tbp, row_l<C>(wv, l));
  if constexpr (DIAG && FUSE) STAMP(0, 2);
  __builtin_amdgcn_sched_barrier(0);  // phase fence: nothing of the next phase is hoisted up here
  const unsigned urow = (unsigned)row * C::N;
  double sEdge = 0.0;
  const int ls = row_l<C>(wv, l);
  // FUSE: between the steps of one call nothing reads U from HBM (the next step continues from the
  // registers) except the tail's np.gradient row-edge terms, which look at rows 0, 1, N-2, N-1: with
  // store_u == 0 only the workgroups owning those rows write them (chs_fast_step decides).
  // The whole field at the end of a call (FUSE with store_u) is streamed out: nothing on the device reads it before
  // the next call's entry.  The edge rows of every step and the field of the unfused pipeline are read right away
  // (tail, k_diag, k_row_fwd): ordinary stores.
  const bool write_u = !FUSE || store_u || row0 < 2 || row0 + C::C > C::N - 2;
  auto put_u = [&](auto nt) {
#pragma unroll
    for (int q = 0; q < C::NP0; ++q) {
      const int m1 = ls + C::G * q, m2 = C::L1 - 1 - m1;
#pragma unroll
      for (int j = 0; j < C::R0 / 2; ++j) {
        T q1[4], q2[4];
        unpack_quads<C>(z, q, j, q1, q2);
        T* p1 = at_boff(U, (urow + 4u * (unsigned)(m1 + C::L1 * j)) * (unsigned)sizeof(T));
        T* p2 = at_boff(U, (urow + 4u * (unsigned)(m2 + C::L1 * j)) * (unsigned)sizeof(T));
        if constexpr (decltype(nt)::value) { store4_nt<T>(p1, q1); store4_nt<T>(p2, q2); }
        else { store4<T>(p1, q1); store4<T>(p2, q2); }
      }
    }
  };
#ifndef CHS_FUSED_U_NT
#define CHS_FUSED_U_NT 1
#endif
  if (FUSE && store_u && CHS_FUSED_U_NT) put_u(std::true_type{});
  else if (write_u) put_u(std::false_type{});
  if constexpr (DIAG) {
    // np.gradient edge columns: (U[r,1]-U[r,0]) and (U[r,N-1]-U[r,N-2]) live in lane 0
    if (ls == 0) {
      T q1[4], q2[4];
      unpack_quads<C>(z, 0, 0, q1, q2);
      const double d0 = (double)q1[1] - (double)q1[0];
      unpack_quads<C>(z, 0, C::R0 / 2 - 1, q1, q2);
      const double d1 = (double)q2[3] - (double)q2[2];
      sEdge += d0 * d0 + d1 * d1;
    }
  }
  if constexpr (DIAG && FUSE) STAMP(0, 3);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (DIAG) {
    // Ra of row int(N/2)+1 (solver.py:226-227): only the workgroup that owns that row takes this
    // block-uniform branch.  (The row-edge terms of np.gradient -- rows 0/1 and N-2/N-1 -- are
    // added by the tail, which reads those four rows back from HBM: chs_tail.h.)
    constexpr int RR = C::N / 2 + 1;
    const bool ra_blk = (row0 == (RR / C::C) * C::C);
    if (ra_blk) {
      // Ra: mean absolute deviation of one row from its own mean (two passes over registers)
      const bool mine = ra_blk && row_sub<C>(wv, sub) == RR % C::C;
      double rs = 0.0;
      if (mine) {
#pragma unroll
        for (int e = 0; e < C::E; ++e) rs += (double)cx_re(z[e]) + (double)cx_im(z[e]);
      }
      const double rmean = block_sum_w(rs, red, wv, C::THREADS / 64) / (double)C::N;
      double ad = 0.0;
      if (mine) {
#pragma unroll
        for (int e = 0; e < C::E; ++e) ad += fabs((double)cx_re(z[e]) - rmean) + fabs((double)cx_im(z[e]) - rmean);
      }
      const double ra = block_sum_w(ad, red, wv, C::THREADS / 64) / (double)C::N;
      if (ra_blk && wv == 0 && chs_lane_id() == 0) partRa[0] = ra;
      __syncthreads();
    }
    const T RT = (T)dc.RT, BRT = (T)dc.BRT, B = (T)dc.B, A0 = (T)dc.A0, A1 = (T)dc.A1;
    const double mean = mean_u, thr = dc.threshold;
    double sE = 0.0, sPS = 0.0, s2 = 0.0;
    int cSA = 0;
    // Domain (numpy: log of a non-positive number is NaN / -inf, which the reference turns into its NaN
    // assertion, timedata.py:10): the table index of the logs doubles as the check (chs_log_unit_tab),
    // one integer maximum per log; the sums are poisoned at the end.
    unsigned dom = 0;
    auto point = [&](T& u) {
      const T uinv = T(1) - u;
      const T lU = chs_log_unit_tab<T>(u, ltab, dom), lV = chs_log_unit_tab<T>(uinv, ltab, dom);
      sE += (double)chs_energy_from_logs_fast<T>(u, uinv, lU, lV, RT, B, A0, A1);
      sPS += fabs((double)u - mean);
      cSA += ((double)u < thr) ? 1 : 0;
      if constexpr (FUSE) {
        const T m = chs_mu_from_logs_fast<T>(u, uinv, lU, lV, RT, BRT, A0, A1);
        s2 += (double)m * (double)m;
        u = m;
        // opaque use: finishes this grid point before the next one starts, so the
        // intermediates (logs, 1-U, ...) of 128 points are never alive together
        asm volatile("" : "+v"(u), "+v"(s2));
      }
      // ... and the running sums: otherwise the compiler postpones all 2E energy terms
      // (keeping log U, log(1-U), 1-U of every point alive) to add them up at the end
      asm volatile("" : "+v"(sE), "+v"(sPS), "+v"(cSA), "+v"(dom));
    };
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
      T a = cx_re(z[e]), b = cx_im(z[e]);
      point(a);
      point(b);
      z[e] = cx_make(a, b);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (dom > (unsigned)(CHS_LOGTAB_N - 1)) sE = __builtin_nan("");  // U left (0,1): the record of this step becomes NaN
    // The five sums leave the registers HERE (wavefront sums into `red`), not at the end of the kernel: carried across the
    // forward transform they were what the register allocator spilled (and a spill reload waits behind every T1 store).
    const double acc[5] = {sE, sEdge, sPS, (double)cSA, s2};
    block_reduce_begin<5, FreshLane<C>::value>(acc, red, wv);
  }
  if constexpr (ADAPT) {
    // Adaptive step (solver.py:177-183): on the steps whose successor re-evaluates delt, the column
    // sums of delt_max/sqrt(1 + alpha*mu^2) are needed.  mu of the next step is in the registers
    // right now: every workgroup adds up its rows per column (through the idle exchange scratch) and
    // writes one partial row; k_colsum_slices / k_colmin_slices add them up and take the minimum.  No sweep of U.
    // The partial sums of a workgroup's few rows travel in the engine's element type (fp32 engine: a sum of two to
    // four fp32 terms per column, added up in fp64 by the reduction kernels -- half the bytes of the N/2 partial rows).
    // NH passes over the columns when a row of them does not fit the scratch
    constexpr int LDSB = C::C * C::SCR * (int)sizeof(T);
    constexpr int NH = (C::C == 1 || C::N * (int)sizeof(T) <= LDSB) ? 1 : 2;
    constexpr int JH = (C::R0 / 2) / NH;  // quads j*L1 .. (j+1)*L1 cover a quarter (R0 = 8) of the columns each
    static_assert(!ADAPT || (FUSE && C::C <= 4 && (C::C == 1 || C::N * (int)sizeof(T) / NH <= LDSB) && (C::R0 / 2) % NH == 0),
                  "partial column sums need the scratch");
    const long long cs = st->computed_steps + 1;  // the record of this step has not advanced it yet
    if (cs > 500 && (cs % 2) == 0) {              // (uniform) cf. k_mu's want_col
      const int lg = row_l<C>(wv, l);
      const int sub_a = row_sub<C>(wv, sub);
      T* prow = partColRows + (size_t)blockIdx.x * C::N;
      T* gl = reinterpret_cast<T*>(chs_dyn_lds);
      const T dmax = (T)dc.delt_max;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int s = C::C - 1; s >= 0; --s) {
          __syncthreads();
          if (sub_a == s) {
#pragma unroll
            for (int q = 0; q < C::NP0; ++q) {
              const int m1 = lg + C::G * q, m2 = C::L1 - 1 - m1;
#pragma unroll
              for (int j = h * JH; j < (h + 1) * JH; ++j) {
                T q1[4], q2[4];
                unpack_quads<C>(z, q, j, q1, q2);  // mu of row `row`, columns 4*(m + L1*j) .. +3
                const int c1 = 4 * (m1 + C::L1 * j), c2 = 4 * (m2 + C::L1 * j);
                const int o1 = c1 - h * (C::N / NH), o2 = c2 - h * (C::N / NH);
                T g1[4], g2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  g1[e] = chs_dt_integrand_fast(q1[e], dmax);
                  g2[e] = chs_dt_integrand_fast(q2[e], dmax);
                  if (s < C::C - 1) { g1[e] += gl[o1 + e]; g2[e] += gl[o2 + e]; }
                  if (s > 0) { gl[o1 + e] = g1[e]; gl[o2 + e] = g2[e]; }
                }
                if (s == 0) {
                  // the quad's four columns are contiguous in the partial row: ONE 16-byte store per element pair/quad (element
                  // by element the non-temporal stores stayed 4-byte instructions: 64 of them per thread at N=8192 fp32).
                  // Written once, read once by the reduction kernels: streamed where the grid does not fit the cache
                  // (N=8192 fp32 adaptive 1893 -> 1925 steps/s; N=4096 fp64, where it fits, 4119 -> 3974 if streamed)
                  if constexpr (chs_grid_exceeds_cache(C::N, sizeof(T))) { store4_nt<T>(&prow[c1], g1); store4_nt<T>(&prow[c2], g2); }
                  else { store4<T>(&prow[c1], g1); store4<T>(&prow[c2], g2); }
                }
                __builtin_amdgcn_sched_barrier(0);  // one quad pair at a time: no pile-up of addresses and operands
              }
            }
          }
        }
      }
      __syncthreads();  // the scratch goes back to the forward passes
    }
  }
  if constexpr (DIAG && FUSE) STAMP(0, 4);
  if constexpr (FUSE) {
    __builtin_amdgcn_sched_barrier(0);
    T* scr_f = reinterpret_cast<T*>(chs_dyn_lds) + (size_t)row_sub<C>(wv, sub) * C::SCR;
    fwd_passes<C, RTW>(z, scr_f, tbp, row_l<C>(wv, l));
    if constexpr (DIAG && FUSE) STAMP(0, 5);
    __builtin_amdgcn_sched_barrier(0);
    const int lr = row_l<C>(wv, l), sr = row_sub<C>(wv, sub);
    const int rw = row_of_block<C>(blockIdx.x) + sr;
#ifndef CHS_ROW_OUT_PIPE
#define CHS_ROW_OUT_PIPE 0
#endif
    // (CHS_ROW_OUT_PIPE: the twiddles of slot k+1 requested in front of slot k's T1 stores, as k_col's spectral stage does)
    recombine<C, true, false, (CHS_ROW_OUT_PIPE != 0)>(z, tb, lr, [](int, const int*) { return NoFetch{}; },
                              [](int, const int*, Cx<T>&, Cx<T>&, bool, NoFetch) {},
                              [&](int pbase, const int* idx, Cx<T>& Ya, Cx<T>& Yb, bool live) {
      if (live) {
        const T y[4] = {cx_re(Ya), cx_im(Ya), cx_re(Yb), cx_im(Yb)};
        row_store<C>(T1, rw, pbase, launder(lr), idx, y);
      }
    });
  }
  if constexpr (DIAG && FUSE) STAMP(0, 6);
  if constexpr (DIAG) {
    // the wavefront sums have been in `red` since the pointwise part: one barrier, one thread adds them up
    double out5[5];
    if (block_reduce_end<5, C::THREADS / 64>(red, out5, wv)) {
      double* p = partDiag + (size_t)blockIdx.x * 4;
      p[0] = out5[0]; p[1] = out5[1]; p[2] = out5[2]; p[3] = out5[3];
      if (FUSE) partMu[blockIdx.x] = out5[4];
    }
  }
}

// ---------------------------------------------------------------------------
// k_col: one workgroup per column tile (C columns, one group each).
// The tile (N rows x C columns, contiguous) is staged through LDS in two rounds of
// N/2 rows so that HBM sees only 16-byte coalesced accesses.
// MODE_STEP also accumulates sum(hat_U^2 (sin^2(pi kr/N) + sin^2(pi kc/N))) per tile:
// by Parseval this is the interior part of np.gradient's sum of squares (solver.py:
// 213-217) -- see DESIGN.md section "E2 from the spectrum".
// ---------------------------------------------------------------------------
// ---- LDS-DMA stage-in (ColDma): the tile rows go from L2 straight into LDS (global_load_lds_dwordx4: no staging
// registers, no ds_write), every piece of both halves requested at kernel entry, ONE wait.
// byte address of an LDS object inside the workgroup's allocation (what DS instructions and M0 take)
__device__ __forceinline__ unsigned lds_byte_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// one LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to 1 KiB of LDS at `lds_dst` (wave-uniform).
// M0 carries the destination; it is compiler-reserved, hence saved and restored inside the statement
// (cdna_hip_programming.md, inline-assembly rules).
template <bool NT>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// which configurations stage in by LDS-DMA (chs_fast_f64.hip)
template <class C>
struct ColDma { static constexpr bool value = false; };
// Landing-zone image of one half of the tile (N/2 rows): one 16-byte unit per tile row (the workgroup's 2 columns).  A DMA
// piece writes 1 KiB linearly -- lane i's 16 bytes at base + 16 i -- so the bank spread that the padded image of the
// register path gets from its pitch comes from permuting the rows INSIDE each quad on the SOURCE side: unit 4m + (e ^
// ((m >> 2) & 3)) holds row 4m + e.  The lanes of a piece still fetch the same 16 lines (coalescing unchanged); the quad
// reads are 2-way conflicted at worst (lanes i and i+16: 32 lanes x 8 bytes out of 16-byte units reach half the banks).
template <class C>
struct DmaStage {
  static constexpr int ROWS = C::N / 2;
  static constexpr int NW = C::THREADS / 64;
  static constexpr int NDMA = ROWS / (64 * NW);   // pieces per wavefront and half
  static constexpr int ZONE = ROWS * C::C;         // elements of a landing zone
  static constexpr bool OK = (C::C * sizeof(typename C::T) == 16) && (C::G % 64 == 0) && (ROWS % (64 * NW) == 0) && (C::R0 >= 4);
  static __device__ __forceinline__ int row_of_unit(int p) { const int mq = p >> 2; return 4 * mq + ((p & 3) ^ ((mq >> 2) & 3)); }
  static __device__ __forceinline__ int unit_of(int mm, int e) { return 4 * mm + (e ^ ((mm >> 2) & 3)); }
};

template <class C>
struct ColStage {
  // A workgroup stages C of the CT columns of a tile: per row a piece of C elements at offset
  // h*C inside the row's CT elements (C == CT: the whole, contiguous tile).
  static constexpr int LINE = 4 * C::C;        // elements of one staged 4-row line in LDS
  static constexpr int LP = LINE + 1;          // padded line pitch in LDS (conflict-free quad reads)
  static constexpr int LINES = C::M / 4;       // lines per round (N/2 rows)
  static constexpr int ROWS = C::N / 2;        // rows per round
  static constexpr int ELEMS = LINES * LP;     // staging elements
  static constexpr int JR = C::R0 / 4;         // pass-0 half-indices j per round
  // elements per piece: a 16-byte piece when it can be (fp64: 2 columns; fp32: 4 columns), else 8 bytes, else one element
  static constexpr int PW = (sizeof(typename C::T) == 4 && C::C % 4 == 0) ? 4 : ((C::C % 2 == 0) ? 2 : 1);
  static constexpr int PER = ROWS * C::C / (PW * C::THREADS);   // pieces per thread per round
  static constexpr int Q = C::CT / C::C;       // workgroups per tile
  static_assert(C::R0 >= 4, "pass-0 radix must be >= 4");
  static_assert((ROWS * C::C) % (PW * C::THREADS) == 0, "the tile must split evenly over the threads");
  // global element offset (inside the tile) and LDS offset of piece f = PW*(tid + i*THREADS) of round rho
  static __device__ __forceinline__ size_t goff(int rho, int f, int h) {
    const int row = f / C::C, c = f % C::C;
    return ((size_t)rho * ROWS + row) * C::CT + h * C::C + c;
  }
  static __device__ __forceinline__ int loff(int f) {
    const int row = f / C::C, c = f % C::C;
    return (row / 4) * LP + (row % 4) * C::C + c;
  }
};

// k_col<MODE_STEP> keeps the pass twiddles in LDS (value 1) unless a configuration says otherwise: 0 = all from L2,
// 2 = the middle-pass tables and only the k = 1 entries of the pass-0 table in LDS, the other pass-0 twiddles formed as
// their powers (tw0_load<POW>) -- where the whole table would cost the second workgroup of a CU (chs_fast_f32.hip)
template <class C>
struct ColTwLds { static constexpr int value = 1; };
// k_col<MODE_STEP> sits at its 256 registers; whether the step's two coefficients are forced back into scalar registers
// in front of the spectral stage (4 VGPRs less) is chosen per configuration by what the register allocator makes of it
// (tools/scratch_check.py: no spill in any k_col from N = 1024 upwards)
template <class C>
struct ColLamSgpr { static constexpr bool value = true; };
// ... and whether the tile workgroups start with their loads (stop flag taken along with the first staging barrier, twiddle
// copy requested in front of the first tile request and stored behind it) instead of two L2 round trips and a barrier in
// front of them: where the extra registers held across the request do not spill (fp64 below N = 4096: 48-80 bytes)
#ifndef CHS_COL_PIPE
#define CHS_COL_PIPE 1   // 0: the spectral stage fetches slot k in slot k (fewer live registers; measured)
#endif
#ifndef CHS_COL_PRE0
#define CHS_COL_PRE0 0   // (measured: fp32 within +-0.5 %, fp64 -- with the coefficients left in VGPRs to avoid scratch -- 2.4 % slower)
#endif
#ifndef CHS_COL_PRE0_F64
#define CHS_COL_PRE0_F64 0   // (fp64 at its 256 registers: 20-28 bytes of scratch with it, unless the coefficients stay in VGPRs)
#endif
// ... and whether the spectral stage's first request goes out in front of the last forward pass (k_col: PRE0)
template <class C>
struct ColPre0 { static constexpr bool value = (CHS_COL_PRE0 != 0) && (sizeof(typename C::T) == 4 || (CHS_COL_PRE0_F64 != 0 && C::N >= 4096)); };
template <class C>
struct ColLateStart { static constexpr bool value = (sizeof(typename C::T) == 4) || (C::N >= 4096); };
template <class C>
constexpr int col_tw_lds_elems() {
  if (ColTwLds<C>::value == 1) return 2 * ((C::R0 - 1) * C::L1 + (C::RA > 1 ? (C::RA - 1) * C::L2 : 0) + (C::RB > 1 ? (C::RB - 1) * C::L3 : 0));
  if (ColTwLds<C>::value == 2) return 2 * (C::L1 + (C::RA > 1 ? (C::RA - 1) * C::L2 : 0) + (C::RB > 1 ? (C::RB - 1) * C::L3 : 0));
  return 0;
}

template <class C>
constexpr int col_lds_elems() {
  constexpr int base = (C::C * C::SCR > ColStage<C>::ELEMS) ? C::C * C::SCR : ColStage<C>::ELEMS;
  // (LDS-DMA stage-in: the two landing zones, which the exchange scratch and the stage-out image then reuse)
  return (ColDma<C>::value && 2 * DmaStage<C>::ZONE > base) ? 2 * DmaStage<C>::ZONE : base;
}

template <class C, int MODE>
__global__ __launch_bounds__(C::THREADS, C::WPS) void k_col(const typename C::T* __restrict__ Tin, typename C::T* __restrict__ Tout,
                                                typename C::T* __restrict__ hat, typename C::T* __restrict__ nat,
                                                FTables<typename C::T> tb, const double* __restrict__ lam,
                                                const double* __restrict__ sinsq, DevState* __restrict__ st,
                                                double* __restrict__ partE2, TailArgs ta) {
  using T = typename C::T;
  using CS = ColStage<C>;
  __shared__ double red[32];
  // MODE_STEP: block 0 of this very launch may raise the stop flag (the riding tail, a gate that timed out): one thread
  // reads it for the whole workgroup, so that all its wavefronts leave or stay together.  EARLY_HALT: at the very top, in
  // front of everything (a load, a barrier and an LDS round trip before the first tile load is even requested); otherwise
  // the tile workgroups take the flag along with their first staging barrier -- nothing but loads has been issued by then.
  __shared__ int halted;
#ifndef CHS_COL_LATE_HALT
#define CHS_COL_LATE_HALT 1
#endif
  constexpr bool LATE_HALT = (MODE == MODE_STEP) && (CHS_COL_LATE_HALT != 0) && ColLateStart<C>::value && !(ColDma<C>::value);
  if constexpr (MODE == MODE_STEP && !LATE_HALT) {
    if (threadIdx.x == 0) halted = st->halt;
    __syncthreads();
    if (halted) return;
  } else if constexpr (MODE != MODE_STEP) {
    if (st->halt) return;
  }
  if constexpr (MODE == MODE_STEP) STAMP(1, 0);
  T* lds = reinterpret_cast<T*>(chs_dyn_lds);
  int bid = blockIdx.x;
  if constexpr (MODE == MODE_STEP) {
    // One extra workgroup carries the record of the PREVIOUS step and this step's time bookkeeping
    // (chs_fast_step) instead of a launch of its own.  Gated (somebody waits for its decision): block 0,
    // dispatched first, so that its short chain of dependent loads runs under the first wave of tiles.
    // Deferred (nobody waits, ta.gate == 0): the LAST block -- the tiles are a whole number of rounds of
    // workgroups; an extra workgroup at the front pushes its slot's tiles back by its own duration, at the back
    // it fills the gap the first slot to finish leaves.
    if (ta.enabled) {
      const bool at_end = !ta.gate;
      if (bid == (at_end ? (int)gridDim.x - 1 : 0)) {
        if constexpr (LATE_HALT) {   // (the bookkeeping workgroup keeps its check up front)
          if (threadIdx.x == 0) halted = st->halt;
          __syncthreads();
          if (halted) return;
        }
        step_tail_body<C::THREADS>(ta, st, reinterpret_cast<double*>(chs_dyn_lds));
        return;
      }
      if (!at_end) bid -= 1;
    }
  }
  const int l = threadIdx.x % C::G, sub = threadIdx.x / C::G;
  // tile and the part of it this workgroup owns; the Q workgroups of a tile get block numbers
  // b, b+8, ...: same XCD under round-robin dispatch (speed only, see row_of_block)
  int ct, hh;
  if constexpr (CS::Q == 1) {
    ct = bid; hh = 0;
  } else {
    const int xcd = bid & 7, j = bid >> 3;
    ct = xcd + 8 * (j / CS::Q); hh = j % CS::Q;
  }
  if constexpr (MODE == MODE_STEP) {
    // every other step walks the tiles in the opposite direction: what the previous step touched
    // last (hat_U of its last tiles, still in the 256 MB Infinity Cache) is touched first
    if (ta.reverse) ct = C::N / C::CT - 1 - ct;
  }
  const int kcp = ct * C::CT + hh * C::C + sub;  // this group's (physical) column of T and hat_U ...
  const int kc = col_coef<C>(kcp);               // ... and the coefficient index it holds (slot order, slot_boff)
  T* scr = lds + (size_t)sub * C::SCR;
  typename C::V z[C::E];
  // MODE_STEP: the twiddles of the radix passes come from LDS (copied once per workgroup; visible
  // behind the barriers of the stage-in): no L2 round trip per pass, and no load that would have to
  // wait behind the hat_U stores at the start of the inverse passes
  FTables<T> tbp = tb;
  constexpr bool TW0POW = (MODE == MODE_STEP) && (ColTwLds<C>::value == 2);
  constexpr bool DMA = ColDma<C>::value && (MODE == MODE_STEP || MODE == MODE_FWD_NATIVE);
  static_assert(!DMA || (DmaStage<C>::OK && sizeof(T) == 8), "LDS-DMA stage-in: 16-byte row pieces of whole wavefronts");
  // (DMA: the twiddle copy is requested in front of the DMA pieces and stored behind their wait, below)
  // LATE_HALT (the register-staged path): the copy's loads are requested HERE and stored to LDS behind the request of
  // the first half of the tile -- as a load/store loop in front of it, the copy was two L2 round trips in a row during
  // which no tile load was in flight (~1.5 K cycles of a ~47 K-cycle workgroup life)
  constexpr bool TW_SPLIT = LATE_HALT && (ColTwLds<C>::value != 0);
  constexpr int TW_N0 = TW0POW ? 2 * C::L1 : 2 * (C::R0 - 1) * C::L1;
  constexpr int TW_NM = col_tw_lds_elems<C>() - TW_N0;
  constexpr int TW_I0 = (TW_N0 + 2 * C::THREADS - 1) / (2 * C::THREADS), TW_IM = (TW_NM + 2 * C::THREADS - 1) / (2 * C::THREADS);
  using TwPair = typename std::conditional<sizeof(T) == 8, double2, float2>::type;
  [[maybe_unused]] TwPair tws_a[TW_SPLIT ? TW_I0 : 1], tws_b[TW_SPLIT ? (TW_IM > 0 ? TW_IM : 1) : 1];
  if constexpr (TW_SPLIT) {
#pragma unroll
    for (int i = 0; i < TW_I0; ++i) {
      const int e = 2 * ((int)threadIdx.x + i * C::THREADS);
      if (e < TW_N0) tws_a[i] = *reinterpret_cast<const TwPair*>(tb.tw0 + e);
    }
#pragma unroll
    for (int i = 0; i < TW_IM; ++i) {
      const int e = 2 * ((int)threadIdx.x + i * C::THREADS);
      if (e < TW_NM) tws_b[i] = *reinterpret_cast<const TwPair*>(tb.twa + e);
    }
    T* ltw = lds + col_lds_elems<C>();
    tbp.tw0 = ltw;
    tbp.twa = ltw + TW_N0;
    tbp.twb = tbp.twa + (C::RA > 1 ? 2 * (C::RA - 1) * C::L2 : 0);
  }
  if constexpr (MODE == MODE_STEP && ColTwLds<C>::value != 0 && !DMA && !TW_SPLIT) {
    T* ltw = lds + col_lds_elems<C>();
    // pass-0 part (the whole table, or its k = 1 entries), then twa | twb (contiguous behind tw0 in the table buffer)
    constexpr int N0 = TW0POW ? 2 * C::L1 : 2 * (C::R0 - 1) * C::L1;
    constexpr int NM = col_tw_lds_elems<C>() - N0;
    auto cp = [&](T* dst, const T* src, int n) {
      for (int i = 2 * threadIdx.x; i < n; i += 2 * C::THREADS) {
        if constexpr (sizeof(T) == 8) *reinterpret_cast<double2*>(dst + i) = *reinterpret_cast<const double2*>(src + i);
        else *reinterpret_cast<float2*>(dst + i) = *reinterpret_cast<const float2*>(src + i);
      }
    };
    cp(ltw, tb.tw0, N0);
    cp(ltw + N0, tb.twa, NM);
    tbp.tw0 = ltw;
    tbp.twa = ltw + N0;
    tbp.twb = tbp.twa + (C::RA > 1 ? 2 * (C::RA - 1) * C::L2 : 0);
  }
  T* hcol = hat + (size_t)kcp * C::N;
  // MODE_STEP with a second hat_U buffer (`nat`, chs_fast_step): the updated coefficients go there and the ones
  // read stay what they were -- the state of the last completed step if the riding tail stops the run
  T* hout = (MODE == MODE_STEP && nat != nullptr) ? nat + (size_t)kcp * C::N : hcol;
  // constants of the spectral stage, requested here: their latency disappears behind the stage-in
  // (loaded where they are used they cost every workgroup ~4 K cycles of waiting)
  double lam1 = st->lam1, lam2 = st->lam2;  // (gated launches read them again behind the gate)
  // (the column index is wave-uniform when a group fills whole wavefronts: scalar loads, no VGPRs)
  const int kc_u = (C::G >= 64) ? __builtin_amdgcn_readfirstlane(kc) : kc;
  const double lc = lam[kc_u];
  const double sqc = (MODE == MODE_STEP) ? sinsq[2 * kc_u + 1] : 0.0;
  // What the spectral stage reads per recombination slot (4 positions of this lane), fetched one slot
  // ahead: hat_U (two value pairs) and the eigenvalues / gradient weights of its four coefficients -- fp64:
  // {lambda_kr, sin^2(pi kr/N)} per coefficient (L2); fp32: one 16-byte entry of each per slot (FTables::lam4, sin4).
  // Where T and hat_U together do not fit the 256 MiB Infinity Cache, hat_U -- touched once per step, here -- is
  // streamed (non-temporal loads and stores) and T, which both kernels read and write, keeps the cache: N=8192 fp32
  // (T = 256 MiB) fused row kernel 205 -> 174 us, k_col 313 -> 302 us, 1913 -> 2096 steps/s; at N=8192 fp64 nothing
  // fits either way (no change); at N=4096 fp64 the two arrays are exactly the cache's size and streaming hat_U costs
  // 4-6 % (profiles/r03_ab_nt.txt).  Loads or stores alone change nothing.
  constexpr bool HAT_STREAM = chs_grid_exceeds_cache(C::N, sizeof(T));
  constexpr bool HAT_NT_LD = HAT_STREAM, HAT_NT_ST = HAT_STREAM;
  struct Fetched64 { double2 ls[4]; Cx<T> h01, h23; };
  struct Fetched32 { float4 la, sa; Cx<T> h01, h23; };
  using Fetched = typename std::conditional<sizeof(T) == 8, Fetched64, Fetched32>::type;
  auto fetch = [&](int pbase, const int* idx) {
    Fetched p;
    if constexpr (sizeof(T) == 8) {
#pragma unroll
      for (int t = 0; t < 4; ++t) p.ls[t] = reinterpret_cast<const double2*>(sinsq)[idx[t]];
    } else {
      p.la = reinterpret_cast<const float4*>(tb.lam4)[idx[4]];
      p.sa = reinterpret_cast<const float4*>(tb.sin4)[idx[4]];
    }
    const int hp = hat_pair_index<C>(pbase, fc_opaque(l));
    p.h01 = ldc_hint<T, HAT_NT_LD>(hcol, hp);
    p.h23 = ldc_hint<T, HAT_NT_LD>(hcol, hp + C::G);
    return p;
  };
  // The spectral stage's first request (slot 0 of pair 0: twiddles, hat_U, eigenvalues) goes out in front of the LAST forward
  // pass instead of behind it: its round trip (hat_U comes from the Infinity Cache) runs under that pass's butterflies
  // (ColPre0; where a lane owns one pair, E = 2 RL, and the stage is pipelined)
  constexpr bool PRE0 = (MODE == MODE_STEP) && ColPre0<C>::value && !DMA && !PreAll<C>::value && (CHS_COL_PIPE != 0) && (C::NP2 == 1);
  // (hat_U alone: it is what comes from furthest away; the whole request held across the pass spilled 20-40 bytes in fp64)
  struct HatPre {
    Cx<T> h01, h23;
    __device__ __forceinline__ void patch(Fetched& p) const { p.h01 = h01; p.h23 = h23; }
  };
  [[maybe_unused]] HatPre s0;
  if constexpr (DMA) {
    // ---- stage in by LDS-DMA: all pieces of both halves of the tile (NDMA per wavefront and half) are requested here,
    // in front of them the (compiler-visible) loads of the pass twiddles; one wait covers everything (vector-memory
    // operations complete in order), then the twiddles go to LDS, one barrier, and every lane reads its quads.
    using DS = DmaStage<C>;
    constexpr bool TWL = (MODE == MODE_STEP && ColTwLds<C>::value != 0);
    constexpr int N0 = TW0POW ? 2 * C::L1 : 2 * (C::R0 - 1) * C::L1;
    constexpr int NM = col_tw_lds_elems<C>() - N0;
    constexpr int TWI0 = (N0 + 2 * C::THREADS - 1) / (2 * C::THREADS), TWIM = (NM + 2 * C::THREADS - 1) / (2 * C::THREADS);
    [[maybe_unused]] double2 tw_a[TWL ? TWI0 : 1], tw_b[TWL ? TWIM : 1];
    T* ltw = lds + col_lds_elems<C>();
    if constexpr (TWL) {
#pragma unroll
      for (int i = 0; i < TWI0; ++i) {
        const int e = 2 * ((int)threadIdx.x + i * C::THREADS);
        if (e < N0) tw_a[i] = *reinterpret_cast<const double2*>(tb.tw0 + e);
      }
#pragma unroll
      for (int i = 0; i < TWIM; ++i) {
        const int e = 2 * ((int)threadIdx.x + i * C::THREADS);
        if (e < NM) tw_b[i] = *reinterpret_cast<const double2*>(tb.twa + e);
      }
      tbp.tw0 = ltw;
      tbp.twa = ltw + N0;
      tbp.twb = tbp.twa + (C::RA > 1 ? 2 * (C::RA - 1) * C::L2 : 0);
    }
    {
      const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
      const T* tile = Tin + (size_t)ct * C::N * C::CT + hh * C::C;
      const unsigned base = lds_byte_addr(lds);
      constexpr bool NT = (MODE == MODE_FWD_NATIVE);  // the entry's intermediate (k_row_fwd2's Ta): its only read
#pragma unroll
      for (int rho = 0; rho < 2; ++rho) {
#pragma unroll
        for (int i = 0; i < DS::NDMA; ++i) {
          const int piece = i * DS::NW + wave;
          const int row = rho * DS::ROWS + DS::row_of_unit(piece * 64 + lane);
          glds16<NT>(tile + (size_t)row * C::CT,
                     __builtin_amdgcn_readfirstlane(base + (unsigned)((rho * DS::ZONE) * sizeof(T)) + piece * 1024));
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (TWL) {
#pragma unroll
      for (int i = 0; i < TWI0; ++i) {
        const int e = 2 * ((int)threadIdx.x + i * C::THREADS);
        if (e < N0) *reinterpret_cast<double2*>(ltw + e) = tw_a[i];
      }
#pragma unroll
      for (int i = 0; i < TWIM; ++i) {
        const int e = 2 * ((int)threadIdx.x + i * C::THREADS);
        if (e < NM) *reinterpret_cast<double2*>(ltw + N0 + e) = tw_b[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int rho = 0; rho < 2; ++rho) {
      const T* zone = lds + rho * DS::ZONE;
#pragma unroll
      for (int q = 0; q < C::NP0; ++q) {
        const int m1 = l + C::G * q, m2 = C::L1 - 1 - m1;
#pragma unroll
        for (int jj = 0; jj < CS::JR; ++jj) {
          const int j = rho * CS::JR + jj;
          T q1[4], q2[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            q1[e] = zone[DS::unit_of(m1 + C::L1 * jj, e) * C::C + sub];
            q2[e] = zone[DS::unit_of(m2 + C::L1 * jj, e) * C::C + sub];
          }
          pack_quads<C>(q1, q2, q, j, z);
        }
      }
    }
    __syncthreads();   // the landing zones become the exchange scratch
    if constexpr (MODE == MODE_STEP) STAMP(1, 1);
    fwd_passes<C, TW0POW>(z, scr, tbp, l);
    if constexpr (MODE == MODE_STEP) STAMP(1, 2);
    if constexpr (MODE == MODE_STEP) {
      if (ta.gate) {  // (the gated tail: see the register-staged path below)
        if (gate_wait(st, ta.seq, ta.gate_spins, red, lam1, lam2)) return;
      }
    }
  } else if constexpr (MODE != MODE_INV_NATURAL && MODE != MODE_INV_NATIVE) {
    // ---- stage in: tile rows -> quads of this group's column.  The first half is requested at once, the
    // second as soon as the first has left its registers (its latency runs under the first half's barrier
    // and quad reads); both pass through LDS half by half.
    const T* tile = Tin + (size_t)ct * C::N * C::CT;
    constexpr int PER = CS::PER;
    constexpr int PW = CS::PW;
    T stage[2][PW * PER];
    auto request = [&](int rho) {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int f = PW * (threadIdx.x + i * C::THREADS);
        const T* src = tile + CS::goff(rho, f, hh);
        constexpr bool NT = (MODE == MODE_FWD_NATIVE);  // the entry's intermediate (k_row_fwd2's Ta): its only read
        if constexpr (PW == 1) {
          stage[rho][i] = NT ? __builtin_nontemporal_load(src) : *src;
        } else if constexpr (PW == 4) {
          chs_f4v v;
          if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const chs_f4v*>(src));
          else v = *reinterpret_cast<const chs_f4v*>(src);
          stage[rho][4 * i] = v.x; stage[rho][4 * i + 1] = v.y; stage[rho][4 * i + 2] = v.z; stage[rho][4 * i + 3] = v.w;
        } else if constexpr (sizeof(T) == 8) {
          if constexpr (NT) {
            const chs_d2v v = __builtin_nontemporal_load(reinterpret_cast<const chs_d2v*>(src));
            stage[rho][2 * i] = v.x; stage[rho][2 * i + 1] = v.y;
          } else {
            const double2 v = *reinterpret_cast<const double2*>(src);
            stage[rho][2 * i] = v.x; stage[rho][2 * i + 1] = v.y;
          }
        } else if constexpr (NT) {
          const chs_f2v v = __builtin_nontemporal_load(reinterpret_cast<const chs_f2v*>(src));
          stage[rho][2 * i] = v.x; stage[rho][2 * i + 1] = v.y;
        } else {
          const float2 v = *reinterpret_cast<const float2*>(src);
          stage[rho][2 * i] = v.x; stage[rho][2 * i + 1] = v.y;
        }
      }
    };
    request(0);
    if constexpr (TW_SPLIT) {
      T* ltw = lds + col_lds_elems<C>();
#pragma unroll
      for (int i = 0; i < TW_I0; ++i) {
        const int e = 2 * ((int)threadIdx.x + i * C::THREADS);
        if (e < TW_N0) *reinterpret_cast<TwPair*>(ltw + e) = tws_a[i];
      }
#pragma unroll
      for (int i = 0; i < TW_IM; ++i) {
        const int e = 2 * ((int)threadIdx.x + i * C::THREADS);
        if (e < TW_NM) *reinterpret_cast<TwPair*>(ltw + TW_N0 + e) = tws_b[i];
      }
    }
    if constexpr (LATE_HALT) {
      if (threadIdx.x == 0) halted = st->halt;
    }
#pragma unroll
    for (int rho = 0; rho < 2; ++rho) {
      __syncthreads();
      if constexpr (LATE_HALT) {
        if (rho == 0 && halted) return;   // (uniform: every thread reads the same LDS word behind the barrier)
      }
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int f = PW * (threadIdx.x + i * C::THREADS);
        const int lo = CS::loff(f);
#pragma unroll
        for (int e = 0; e < PW; ++e) lds[lo + e] = stage[rho][PW * i + e];
      }
      if (rho == 0) request(1);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < C::NP0; ++q) {
        const int m1 = l + C::G * q, m2 = C::L1 - 1 - m1;
#pragma unroll
        for (int jj = 0; jj < CS::JR; ++jj) {
          const int j = rho * CS::JR + jj;
          T q1[4], q2[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            q1[e] = lds[(m1 + C::L1 * jj) * CS::LP + e * C::C + sub];
            q2[e] = lds[(m2 + C::L1 * jj) * CS::LP + e * C::C + sub];
          }
          pack_quads<C>(q1, q2, q, j, z);
        }
      }
    }
    __syncthreads();
    if constexpr (MODE == MODE_STEP) STAMP(1, 1);
#ifndef CHS_GATE_EARLY_POLL
#define CHS_GATE_EARLY_POLL 0   // (measured equal in energy-stop mode: 0.9932 / 0.9954 with it off; experiment switch)
#endif
    // (gated launches: the polling lane asks for the sequence word HERE; the answer is looked at behind the passes)
    [[maybe_unused]] unsigned long long early_seq = ~0ull;
    if constexpr (MODE == MODE_STEP && CHS_GATE_EARLY_POLL != 0) {
      if (ta.gate && threadIdx.x == 0) early_seq = __hip_atomic_load(&st->decided, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (PRE0) fwd_passes<C, TW0POW>(z, scr, tbp, l, [&]() {
      const int hp = hat_pair_index<C>(0, fc_opaque(l));
      s0.h01 = ldc_hint<T, HAT_NT_LD>(hcol, hp);
      s0.h23 = ldc_hint<T, HAT_NT_LD>(hcol, hp + C::G);
    });
    else fwd_passes<C, TW0POW>(z, scr, tbp, l);
    if constexpr (MODE == MODE_STEP) STAMP(1, 2);
    if constexpr (MODE == MODE_STEP) {
      // Gated tail: the bookkeeping of the previous step -- stop rules, adaptive time step -- runs as block 0
      // of THIS launch; nothing has been written yet (staging and forward passes only read), so wait for its
      // decision here: stopped -> leave hat_U, T and the partial sums as the previous step left them
      // (run_steps rebuilds U from hat_U); otherwise take this step's coefficients from it.
      if (ta.gate) {
        if (gate_wait(st, ta.seq, ta.gate_spins, red, lam1, lam2, early_seq)) return;
      }
    }
  }

  // ---- recombination / spectral stage / adjoint recombination, in place per slot
  double e2 = 0.0;
#ifndef CHS_MEAN_NOW_ALL
#define CHS_MEAN_NOW_ALL 0
#endif
  constexpr bool MEAN_NOW = !PreAll<C>::value || (CHS_MEAN_NOW_ALL != 0);
  [[maybe_unused]] T h00 = T(0);
  constexpr bool FWD = (MODE != MODE_INV_NATURAL && MODE != MODE_INV_NATIVE);
  constexpr bool ADJ = (MODE == MODE_STEP || MODE == MODE_INV_NATURAL || MODE == MODE_INV_NATIVE);
  [[maybe_unused]] v2f e2v = {0.0f, 0.0f};  // fp32: the lane's share of the gradient sum, in two packed halves
  if constexpr (MODE == MODE_STEP) {
    if constexpr (ColLamSgpr<C>::value) {
      // the step's coefficients are uniform but reach this point through a select (entry load | the gate's LDS box) and
      // would sit in four VGPRs all through the stage that needs every one of them: back into scalar registers
      auto uni = [](double x) {
        const long long b = __double_as_longlong(x);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
      };
      lam1 = uni(lam1); lam2 = uni(lam2);
    }
    // fp32: the coefficients of this launch as floats (after the gate: lam1/lam2 are this step's)
    [[maybe_unused]] const float lam1f = (float)lam1, lam2f = (float)lam2, lcf = (float)lc, sqcf = (float)sqc;
    auto spec_f =
      [&](int pbase, const int*, Cx<T>& Ya, Cx<T>& Yb, bool live, const Fetched& p) {
        if constexpr (sizeof(T) == 8) {
          T y[4] = {cx_re(Ya), cx_im(Ya), cx_re(Yb), cx_im(Yb)};
          const T hold[4] = {cx_re(p.h01), cx_im(p.h01), cx_re(p.h23), cx_im(p.h23)};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const T h = chs_spectral<T>(hold[t], y[t], p.ls[t].x, lc, lam1, lam2);
            y[t] = h;
            const double term = (double)h * (double)h * (p.ls[t].y + sqc);
            e2 += live ? term : 0.0;
            // lane 0 of column 0 holds the DC coefficient at position 0 (its own slot): ortho DC term = sum(U)/N
            // (solver.py:223), stored right here -- carried to the end of the stage it is two registers too many where
            // the kernel sits at its register limit (a spilled value's reload waits behind the hat_U stores)
            // (the small grids, which request everything of the stage up front, keep it in a register to the end: MEAN_NOW)
            if constexpr (MEAN_NOW) { if (pbase + t == 0 && live && l == 0 && kc == 0) st->meanU = (double)h / (double)C::N; }
            else if (pbase + t == 0 && live) h00 = h;
          }
          asm volatile("" : "+v"(e2));  // do not postpone the 2E energy terms
          Ya = cx_make(y[0], y[1]); Yb = cx_make(y[2], y[3]);
        } else {
          // the semi-implicit update (solver.py:201-206) on value pairs, in single precision like its operands
          // and its result (chs_spectral_f32 is the scalar form): reciprocal + residual correction
          auto upd = [&](v2f hold, v2f y, v2f lamr, v2f sr) {
            const v2f leig = lamr + lcf;
            const v2f CHeig = __builtin_elementwise_fma(lam2f * leig, leig, v2f{1.0f, 1.0f});
            const v2f rhs = __builtin_elementwise_fma(lam1f * leig, y, hold);
            v2f r;
            r.x = __builtin_amdgcn_rcpf(CHeig.x); r.y = __builtin_amdgcn_rcpf(CHeig.y);
            const v2f q = rhs * r;
            const v2f rho = __builtin_elementwise_fma(-CHeig, q, rhs);
            const v2f h = __builtin_elementwise_fma(rho, r, q);
            v2f term = (h * h) * (sr + sqcf);
            if (!live) term = v2f{0.0f, 0.0f};
            e2v += term;
            return h;
          };
          Ya = upd(p.h01, Ya, v2f{p.la.x, p.la.y}, v2f{p.sa.x, p.sa.y});
          Yb = upd(p.h23, Yb, v2f{p.la.z, p.la.w}, v2f{p.sa.z, p.sa.w});
          if constexpr (MEAN_NOW) { if (pbase == 0 && live && l == 0 && kc == 0) st->meanU = (double)Ya.x / (double)C::N; }  // (as in fp64 above)
          else if (pbase == 0 && live) h00 = Ya.x;
        }
      };
    auto spec_st =
      [&](int pbase, const int*, Cx<T>& Ya, Cx<T>& Yb, bool live) {
        if (live) {
          const int hp = hat_pair_index<C>(pbase, fc_opaque(l));
          stc_hint<T, HAT_NT_ST>(hout, hp, Ya);
          stc_hint<T, HAT_NT_ST>(hout, hp + C::G, Yb);
        }
      };
    if constexpr (PRE0) recombine<C, true, true, (CHS_COL_PIPE != 0)>(z, tb, l, fetch, spec_f, spec_st, &s0);
    else recombine<C, true, true, (CHS_COL_PIPE != 0)>(z, tb, l, fetch, spec_f, spec_st);
    if constexpr (sizeof(T) == 4) e2 = (double)e2v.x + (double)e2v.y;
  } else {
    recombine<C, FWD, ADJ, false>(z, tb, l, [](int, const int*) { return NoFetch{}; },
                           [&](int pbase, const int* idx, Cx<T>& Ya, Cx<T>& Yb, bool live, NoFetch) {
      [[maybe_unused]] const int hp = hat_pair_index<C>(pbase, l);
      if constexpr (MODE == MODE_FWD_NATIVE) {
        if (live) { stc<T>(hcol, hp, Ya); stc<T>(hcol, hp + C::G, Yb); }
      } else if constexpr (MODE == MODE_INV_NATIVE) {
        Ya = ldc<T>(hcol, hp); Yb = ldc<T>(hcol, hp + C::G);  // hat_U as MODE_STEP left it: the column half of U = idctn(hat_U)
      } else {
        T y[4] = {cx_re(Ya), cx_im(Ya), cx_re(Yb), cx_im(Yb)};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int kr = idx[t];
          if constexpr (MODE == MODE_FWD_NATURAL) {
            if (live) nat[(size_t)kr * C::N + kc] = y[t];
          } else {
            y[t] = nat[(size_t)kr * C::N + kc];
          }
        }
        if constexpr (MODE != MODE_FWD_NATURAL) { Ya = cx_make(y[0], y[1]); Yb = cx_make(y[2], y[3]); }
      }
    }, [](int, const int*, Cx<T>&, Cx<T>&, bool) {});
  }
  if constexpr (MODE == MODE_STEP) STAMP(1, 5);
  if constexpr (MODE == MODE_STEP && !MEAN_NOW) {
    if (l == 0 && kc == 0) st->meanU = (double)h00 / (double)C::N;  // ortho DC term = sum(U)/N (solver.py:223)
  }
  if constexpr (MODE == MODE_STEP) STAMP(1, 3);
  if constexpr (ADJ) {
    inv_passes<C, TW0POW>(z, scr, tbp, l);
    if constexpr (MODE == MODE_STEP) STAMP(1, 4);
    // ---- stage out: quads -> tile rows
    T* tile = Tout + (size_t)ct * C::N * C::CT;
#pragma unroll
    for (int rho = 0; rho < 2; ++rho) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < C::NP0; ++q) {
        const int m1 = l + C::G * q, m2 = C::L1 - 1 - m1;
#pragma unroll
        for (int jj = 0; jj < CS::JR; ++jj) {
          const int j = rho * CS::JR + jj;
          T q1[4], q2[4];
          unpack_quads<C>(z, q, j, q1, q2);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            lds[(m1 + C::L1 * jj) * CS::LP + e * C::C + sub] = q1[e];
            lds[(m2 + C::L1 * jj) * CS::LP + e * C::C + sub] = q2[e];
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < CS::PER; ++i) {
        // laundered: the staging addresses are recomputed here instead of being kept alive in
        // registers since the stage-in (a spill reload here would wait on every hat_U store)
        const int f = CS::PW * (launder((int)threadIdx.x) + i * C::THREADS);
        const int lo = CS::loff(f);
        T* dst = tile + CS::goff(rho, f, hh);
        if constexpr (CS::PW == 1) {
          *dst = lds[lo];
        } else if constexpr (CS::PW == 4) {
          chs_f4v v; v.x = lds[lo]; v.y = lds[lo + 1]; v.z = lds[lo + 2]; v.w = lds[lo + 3];
          *reinterpret_cast<chs_f4v*>(dst) = v;
        } else {
          const T a = lds[lo], b = lds[lo + 1];
#ifndef CHS_COL_OUT_POLICY
#define CHS_COL_OUT_POLICY 0   // experiment: cache policy of k_col's T stores: 0 plain, 1 sc1 (write-through), 2 nt, 3 sc0 sc1
#endif
          if constexpr (sizeof(T) == 8 && CHS_COL_OUT_POLICY != 0 && MODE == MODE_STEP) {
            chs_d2v v; v.x = a; v.y = b;
            if constexpr (CHS_COL_OUT_POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
            else if constexpr (CHS_COL_OUT_POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(dst), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(v) : "memory");
          } else if constexpr (sizeof(T) == 8) *reinterpret_cast<double2*>(dst) = make_double2(a, b);
          else *reinterpret_cast<float2*>(dst) = make_float2(a, b);
        }
      }
    }
  }
  if constexpr (MODE == MODE_STEP) {
    // the tile's share of the spectral gradient sum, last (registers are free, one barrier)
    const double acc1[1] = {e2};
    double tot1[1];
    block_sum_store<1, C::THREADS / 64>(acc1, red, tot1);
    if (threadIdx.x == 0) partE2[bid] = tot1[0];
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
enum { ROW_INV_PLAIN = 0, ROW_INV_DIAG = 1, ROW_INV_FUSED = 2, ROW_INV_FUSED_ADAPT = 3 };
enum { ROW_FWD_PLAIN = 0, ROW_FWD_POINTWISE = 1, ROW_FWD_STREAM = 2 };

struct FastPlan {
  int N, G, R0, RA, RB, RL, threads, col_tiles;
  void* tables = nullptr;  // one device allocation
  size_t off_tw0, off_twa, off_twb, off_wp, off_t1, off_t2, off_lam4 = 0, off_sin4 = 0;  // element offsets
  int (*row_fwd)(Engine*, const void*, void*, int) = nullptr;   // mode: ROW_FWD_PLAIN / _POINTWISE / _STREAM
  int (*row_fwd2)(Engine*, const void*, void*, void*) = nullptr;
  int (*row_inv)(Engine*, int, const void*, void*, void*) = nullptr;
  int (*col)(Engine*, int, const void*, void*, void*, void*) = nullptr;
  int (*init)(Engine*) = nullptr;
};

template <typename T>
static FTables<T> get_tables(Engine* E) {
  FastPlan* P = (FastPlan*)E->dTw;
  const T* base = (const T*)P->tables;
  FTables<T> tb;
  tb.tw0 = base + P->off_tw0; tb.twa = base + P->off_twa; tb.twb = base + P->off_twb; tb.wp = base + P->off_wp;
  tb.t1 = base + P->off_t1; tb.t2 = base + P->off_t2;
  tb.lam4 = base + P->off_lam4; tb.sin4 = base + P->off_sin4;
  return tb;
}

template <class C, class CC = C>
struct Launch {
  using T = typename C::T;
  static constexpr size_t row_lds = (size_t)C::C * C::SCR * sizeof(T) + CHS_LOGTAB_N * 16 + (size_t)row_tw_lds_elems<C>() * sizeof(T);
  // the fused row kernel can add up the adaptive-step integrand per column itself (chs_fast_step)
  static constexpr bool ADAPT_OK = (C::C <= 4) && (C::C == 1 || (size_t)C::N * sizeof(T) / 2 <= row_lds) && (C::R0 % 4 == 0);
  // staging / exchange scratch + the pass twiddles (k_col<MODE_STEP>)
  static constexpr size_t col_lds = ((size_t)col_lds_elems<CC>() + (size_t)col_tw_lds_elems<CC>()) * sizeof(T);
  static_assert(C::N == CC::N && C::CT == CC::CT, "row/column configs must agree on the tile layout");

  template <class K>
  static int set_lds(K kernel, size_t bytes) {
    CHS_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return CHS_OK;
  }
  static int init(Engine* E) {
    (void)E;
    int rc;
    if ((rc = set_lds(k_row_fwd<C, true>, row_lds))) return rc;
    if ((rc = set_lds(k_row_fwd<C, false>, row_lds))) return rc;
    if ((rc = set_lds(k_row_fwd<C, false, true>, row_lds))) return rc;
    if ((rc = set_lds(k_row_fwd2<C>, row_lds))) return rc;
    if ((rc = set_lds(k_row_inv<C, false, false>, row_lds))) return rc;
    if ((rc = set_lds(k_row_inv<C, true, false>, row_lds))) return rc;
    if ((rc = set_lds(k_row_inv<C, true, true>, row_lds))) return rc;
    if constexpr (ADAPT_OK) {
      if ((rc = set_lds(k_row_inv<C, true, true, true>, row_lds))) return rc;
    }
    E->adaptOk = ADAPT_OK;
    E->fusedAdapt = ADAPT_OK && getenv("CHS_ADAPT_SWEEP") == nullptr;  // CHS_ADAPT_SWEEP=1: keep the separate sweep of U
    if ((rc = set_lds(k_col<CC, MODE_STEP>, col_lds))) return rc;
    if ((rc = set_lds(k_col<CC, MODE_FWD_NATIVE>, col_lds))) return rc;
    if ((rc = set_lds(k_col<CC, MODE_FWD_NATURAL>, col_lds))) return rc;
    if ((rc = set_lds(k_col<CC, MODE_INV_NATURAL>, col_lds))) return rc;
    if ((rc = set_lds(k_col<CC, MODE_INV_NATIVE>, col_lds))) return rc;
    return CHS_OK;
  }
  static int row_fwd(Engine* E, const void* in, void* out, int mode) {
    const int grid = C::N / C::C;
    if (mode == ROW_FWD_POINTWISE)
      k_row_fwd<C, true><<<grid, C::THREADS, row_lds, E->stream>>>((const T*)in, (T*)out, get_tables<T>(E), E->dc, E->dState,
                                                            E->dPartMu);
    else if (mode == ROW_FWD_STREAM)
      k_row_fwd<C, false, true><<<grid, C::THREADS, row_lds, E->stream>>>((const T*)in, (T*)out, get_tables<T>(E), E->dc, E->dState,
                                                                   E->dPartMu);
    else
      k_row_fwd<C, false><<<grid, C::THREADS, row_lds, E->stream>>>((const T*)in, (T*)out, get_tables<T>(E), E->dc, E->dState,
                                                             E->dPartMu);
    CHS_HIP(hipGetLastError());
    return CHS_OK;
  }
  static int row_fwd2(Engine* E, const void* in, void* ta, void* t1) {
    k_row_fwd2<C><<<C::N / C::C, C::THREADS, row_lds, E->stream>>>((const T*)in, (T*)ta, (T*)t1, get_tables<T>(E), E->dc,
                                                                E->dState, E->dPartMu);
    CHS_HIP(hipGetLastError());
    return CHS_OK;
  }
  static int row_inv(Engine* E, int mode, const void* t2, void* u, void* t1) {
    const int grid = C::N / C::C;
    const FTables<T> tb = get_tables<T>(E);
    if (mode == ROW_INV_PLAIN)
      k_row_inv<C, false, false><<<grid, C::THREADS, row_lds, E->stream>>>((const T*)t2, (T*)u, (T*)t1, tb, E->dc, E->dState,
                                                                    E->dPartDiag, E->dPartMu, E->dPartRa, 1);
    else if (mode == ROW_INV_DIAG)
      k_row_inv<C, true, false><<<grid, C::THREADS, row_lds, E->stream>>>((const T*)t2, (T*)u, (T*)t1, tb, E->dc, E->dState,
                                                                   E->dPartDiag, E->dPartMu, E->dPartRa, 1);
    else if (mode == ROW_INV_FUSED)
      k_row_inv<C, true, true><<<grid, C::THREADS, row_lds, E->stream>>>((const T*)t2, (T*)u, (T*)t1, tb, E->dc, E->dState,
                                                                  E->dPartDiag, E->dPartMu, E->dPartRa, E->storeU ? 1 : 0);
    else {
      if constexpr (ADAPT_OK)
        k_row_inv<C, true, true, true><<<grid, C::THREADS, row_lds, E->stream>>>((const T*)t2, (T*)u, (T*)t1, tb, E->dc, E->dState,
                                                                          E->dPartDiag, E->dPartMu, E->dPartRa, E->storeU ? 1 : 0, (T*)E->dPartColRows);
      else { chs_set_error("fused adaptive row kernel is not built for this configuration"); return CHS_EINVAL; }
    }
    CHS_HIP(hipGetLastError());
    return CHS_OK;
  }
  static int col(Engine* E, int mode, const void* tin, void* tout, void* hat, void* nat) {
    const int grid = CC::N / CC::C;
    const FTables<T> tb = get_tables<T>(E);
    TailArgs ta;
    switch (mode) {
      case MODE_STEP: {
        int g = grid;
        if (E->tailDeferred) {
          ta = chs_tail_args(E, E->tailSet, 1);
          g = grid + 1;
          if (E->tailGated) {
            ta.gate = 1; ta.seq = ++E->gateSeq;
            if (E->testGateWithhold) { ta.withhold = 1; ta.gate_spins = 1 << 10; }
            ta.early = E->tailEarly ? 1 : 0;
          }
        } else if (E->preRider) {
          ta = chs_tail_args(E, -1, 1);
          ta.pre_only = 1;
          g = grid + 1;
        }
        E->preRider = false;
        ta.reverse = (CHS_COL_ZIGZAG && (E->stepCount & 1)) ? 1 : 0;
        ++E->stepCount;
        k_col<CC, MODE_STEP><<<g, CC::THREADS, col_lds, E->stream>>>((const T*)tin, (T*)tout, (T*)hat, (T*)nat, tb, E->dLambda, E->dSinSq, E->dState, E->dPartE2, ta);
        break;
      }
      case MODE_FWD_NATIVE:
        k_col<CC, MODE_FWD_NATIVE><<<grid, CC::THREADS, col_lds, E->stream>>>((const T*)tin, (T*)tout, (T*)hat, (T*)nat, tb, E->dLambda, E->dSinSq, E->dState, E->dPartE2, ta);
        break;
      case MODE_FWD_NATURAL:
        k_col<CC, MODE_FWD_NATURAL><<<grid, CC::THREADS, col_lds, E->stream>>>((const T*)tin, (T*)tout, (T*)hat, (T*)nat, tb, E->dLambda, E->dSinSq, E->dState, E->dPartE2, ta);
        break;
      case MODE_INV_NATIVE:
        k_col<CC, MODE_INV_NATIVE><<<grid, CC::THREADS, col_lds, E->stream>>>((const T*)tin, (T*)tout, (T*)hat, (T*)nat, tb, E->dLambda, E->dSinSq, E->dState, E->dPartE2, ta);
        break;
      default:
        k_col<CC, MODE_INV_NATURAL><<<grid, CC::THREADS, col_lds, E->stream>>>((const T*)tin, (T*)tout, (T*)hat, (T*)nat, tb, E->dLambda, E->dSinSq, E->dState, E->dPartE2, ta);
        break;
    }
    CHS_HIP(hipGetLastError());
    return CHS_OK;
  }
};

template <class C, class CC = C>
static void bind(FastPlan* P) {
  P->N = C::N; P->G = C::G; P->R0 = C::R0; P->RA = C::RA; P->RB = C::RB; P->RL = C::RL; P->threads = C::THREADS;
  P->col_tiles = CC::N / CC::C;
  P->row_fwd = &Launch<C, CC>::row_fwd;
  P->row_fwd2 = &Launch<C, CC>::row_fwd2;
  P->row_inv = &Launch<C, CC>::row_inv;
  P->col = &Launch<C, CC>::col;
  P->init = &Launch<C, CC>::init;
}


// the configurations of one element type, bound to a plan (chs_fast_f64.hip / chs_fast_f32.hip)
bool chs_fast_bind_f64(int N, FastPlan* P);
bool chs_fast_bind_f32(int N, FastPlan* P);
