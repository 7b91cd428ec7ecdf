// chs_pointwise.hip -- pointwise sweeps, reductions and the device-side
// bookkeeping kernels (time step control, timedata rows, stop rule).
//
// Reference lines implemented here:
//   k_mu        chsimpy/solver.py:166-175 (+ the operands of 183 and 225)
//   k_colsum_slices / k_colmin_slices    chsimpy/solver.py:183  (np.linalg.norm(.., ord=-1) = min column abs-sum)
//   k_pre       chsimpy/solver.py:177-199,225
//   k_spectral  chsimpy/solver.py:201-206 with chsimpy/utils.py:39-49
//   k_diag      chsimpy/solver.py:213-228 (= 100-116 for prepare)
//   k_fin       chsimpy/solver.py:118-134,230-249 ; chsimpy/timedata.py:8-10,51-63
#include "chs_common.h"
#include "chs_math.h"
#include "chs_tail.h"

#define PW_THREADS 256
#define DISPATCH_T(E, expr_d, expr_f) \
  do { if ((E)->dtype == CHS_F64) { expr_d; } else { expr_f; } } while (0)
#define PW_BAND 8  // rows per block in the banded sweeps
#define DIAG_BAND 32  // rows per block in k_diag (blocks of DIAG_BAND x PW_THREADS points)

// ---------------------------------------------------------------------------
// k_mu: MU = EnergieEut(U); per-band sum(mu^2); per-band column sums of the
// adaptive-step integrand (only when `want_col`).
// grid.x = ceil(N/PW_BAND) row bands, threads sweep the columns.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(PW_THREADS) void k_mu(const T* __restrict__ U, T* __restrict__ MU, DevConsts dc,
                                                   const DevState* __restrict__ st, double* __restrict__ partMu,
                                                   double* __restrict__ partCol, int only_col, int cs_offset) {
  __shared__ double scratch[32];
  if (st->halt) return;
  const int N = dc.N;
  const int r0 = blockIdx.x * PW_BAND;
  const int r1 = min(r0 + PW_BAND, N);
  // cs_offset = 1: launched before the record of the running step has advanced computed_steps
  const long long cs = st->computed_steps + cs_offset;
  const bool want_col = (dc.adaptive_time && cs > 500 && (cs % 2) == 0);
  if (only_col && !want_col) return;
  const T RT = (T)dc.RT, BRT = (T)dc.BRT, A0 = (T)dc.A0, A1 = (T)dc.A1;
  double s2 = 0.0;
  for (int c = threadIdx.x; c < N; c += PW_THREADS) {
    double cs = 0.0;
    for (int r = r0; r < r1; ++r) {
      const T u = U[(size_t)r * N + c];
      const T m = chs_mu<T>(u, RT, BRT, A0, A1);
      if (!only_col) MU[(size_t)r * N + c] = m;
      const double md = (double)m;
      s2 += md * md;
      if (want_col) cs += chs_dt_integrand(md, dc.delt_max);
    }
    if (want_col) partCol[(size_t)blockIdx.x * N + c] = cs;
  }
  const double tot = block_sum(s2, scratch);
  if (threadIdx.x == 0) partMu[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------
// k_pre (one block): L2 of the running step, adaptive time step, time
// bookkeeping and the time-limit stop.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(PW_THREADS) void k_pre(DevConsts dc, DevState* __restrict__ st,
                                                    const double* __restrict__ partMu, int nPartMu,
                                                    const double* __restrict__ partColMin, int nColMin) {
  __shared__ double scratch[32];
  if (st->halt) return;
  double s = 0.0;
  for (int i = threadIdx.x; i < nPartMu; i += PW_THREADS) s += partMu[i];
  const double musq = block_sum(s, scratch);
  double m = 1.0e300;
  const bool adapt = dc.adaptive_time && st->computed_steps > 500 && (st->computed_steps % 2) == 0;
  if (adapt) {
    for (int i = threadIdx.x; i < nColMin; i += PW_THREADS) m = fmin(m, partColMin[i]);
  }
  const double delt_dyn = block_min(m, scratch);
  if (threadIdx.x == 0) pre_update(dc, st, musq, adapt, delt_dyn);
}

// ---------------------------------------------------------------------------
// k_call_begin (one thread): entry of a solve_or_resume call.  Re-arms the loop (`halt` only lives
// inside one call) and reloads the coefficients of params.delt: solver.py:154-155 takes
// solution.Seig/CHeig, which solution.py:52-55 built once from params.delt, whatever self.delt has
// become -- an adaptive run that is resumed works with them until its step is re-evaluated (189-193).
// A kernel instead of a host round trip: nothing waits for it.
// ---------------------------------------------------------------------------
__global__ void k_call_begin(DevConsts dc, DevState* __restrict__ st) {
#pragma clang fp contract(off)
  st->halt = 0; st->nan_flag = 0; st->rows_written = 0; st->gate_timeout = 0; st->colmin_ticket = 0;
  st->delt_coef = dc.delt0;
  const double lam1 = dc.delt0 / dc.delx2;
  st->lam1 = lam1;
  st->lam2 = dc.kappa_tilde * lam1 / dc.delx2;
}

int chs_launch_call_begin(Engine* E) {
  k_call_begin<<<1, 1, 0, E->stream>>>(E->dc, E->dState);
  CHS_HIP(hipGetLastError());
  return CHS_OK;
}

// ---------------------------------------------------------------------------
// k_spectral (direct engine, natural order): hat_U <- (hat_U + Seig*hat_mu)/CHeig
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(PW_THREADS) void k_spectral(T* __restrict__ hatU, const T* __restrict__ hatMu,
                                                         const double* __restrict__ lam, int N,
                                                         DevState* __restrict__ st) {
  if (st->halt) return;
  const double lam1 = st->lam1, lam2 = st->lam2;
  const size_t total = (size_t)N * N;
  for (size_t idx = (size_t)blockIdx.x * PW_THREADS + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * PW_THREADS) {
    const int i = (int)(idx / N), j = (int)(idx - (size_t)i * N);
    const T h = chs_spectral<T>(hatU[idx], hatMu[idx], lam[i], lam[j], lam1, lam2);
    hatU[idx] = h;
    if (idx == 0) st->meanU = (double)h / (double)N;  // ortho DCT-II DC term = sum(U)/N
  }
}

// ---------------------------------------------------------------------------
// k_sum / k_sum_fin: meanU <- mean(U) (prepare, and after jitter).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(PW_THREADS) void k_sum(const T* __restrict__ U, int N, const DevState* __restrict__ st,
                                                    double* __restrict__ partSum, int ignore_halt) {
  __shared__ double scratch[32];
  if (!ignore_halt && st->halt) return;
  const int r0 = blockIdx.x * PW_BAND, r1 = min(r0 + PW_BAND, N);
  double s = 0.0;
  for (int c = threadIdx.x; c < N; c += PW_THREADS)
    for (int r = r0; r < r1; ++r) s += (double)U[(size_t)r * N + c];
  const double tot = block_sum(s, scratch);
  if (threadIdx.x == 0) partSum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(PW_THREADS) void k_sum_fin(const double* __restrict__ partSum, int n, int N,
                                                        DevState* __restrict__ st, int ignore_halt) {
  __shared__ double scratch[32];
  if (!ignore_halt && st->halt) return;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += PW_THREADS) s += partSum[i];
  const double tot = block_sum(s, scratch);
  if (threadIdx.x == 0) st->meanU = tot / ((double)N * (double)N);
}

// ---------------------------------------------------------------------------
// k_diag: one sweep over U for E (bulk density), E2 (np.gradient stencil),
// PS and SA.  grid.x = row bands; partDiag[block][4] = {sE, sG, sPS, cSA}.
// np.gradient(U, delx, axis=[0,1], edge_order=1): interior (f[i+1]-f[i-1])/(2 dx),
// edges (f[1]-f[0])/dx and (f[N-1]-f[N-2])/dx.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(PW_THREADS) void k_diag(const T* __restrict__ U, DevConsts dc,
                                                     const DevState* __restrict__ st,
                                                     double* __restrict__ partDiag, int ignore_halt) {
  __shared__ double scratch[32];
  __shared__ double2 ltab[CHS_LOGTAB_N];
  if (!ignore_halt && st->halt) return;
  if constexpr (sizeof(T) == 8) {
    for (int t = threadIdx.x; t < CHS_LOGTAB_N; t += PW_THREADS) ltab[t] = reinterpret_cast<const double2*>(chs_log_table)[t];
    __syncthreads();
  }
  const int N = dc.N;
  // a block = DIAG_BAND rows x PW_THREADS columns; a thread walks down its column with the three vertical
  // neighbours in registers
  const int r0 = blockIdx.x * DIAG_BAND, r1 = min(r0 + DIAG_BAND, N);
  const int c = blockIdx.y * PW_THREADS + threadIdx.x;
  const double mean = st->meanU;
  const double inv2dx = 1.0 / (2.0 * dc.delx), invdx = 1.0 / dc.delx;
  const T RT = (T)dc.RT, B = (T)dc.B, A0 = (T)dc.A0, A1 = (T)dc.A1;
  double sE = 0.0, sG = 0.0, sPS = 0.0, cSA = 0.0;
  unsigned dom = 0;
  if (c < N) {
    const T* col = U + c;
    T up = r0 > 0 ? col[(size_t)(r0 - 1) * N] : T(0), cur = col[(size_t)r0 * N];
    for (int r = r0; r < r1; ++r) {
      const size_t o = (size_t)r * N;
      const T nxt = r < N - 1 ? col[o + N] : T(0);
      const T lft = c > 0 ? col[o - 1] : T(0), rgt = c < N - 1 ? col[o + 1] : T(0);
      const double ud = (double)cur;
      double gx, gy;
      if (r == 0)
        gx = ((double)nxt - ud) * invdx;
      else if (r == N - 1)
        gx = (ud - (double)up) * invdx;
      else
        gx = ((double)nxt - (double)up) * inv2dx;
      if (c == 0)
        gy = ((double)rgt - ud) * invdx;
      else if (c == N - 1)
        gy = (ud - (double)lft) * invdx;
      else
        gy = ((double)rgt - (double)lft) * inv2dx;
      sG += gx * gx + gy * gy;
      const T uinv = T(1) - cur;
      const T lU = chs_log_unit_tab<T>(cur, ltab, dom), lV = chs_log_unit_tab<T>(uinv, ltab, dom);
      sE += (double)chs_energy_from_logs<T>(cur, uinv, lU, lV, RT, B, A0, A1);
      sPS += fabs(ud - mean);
      cSA += (ud < dc.threshold) ? 1.0 : 0.0;
      up = cur; cur = nxt;
    }
  }
  if (dom > (unsigned)(CHS_LOGTAB_N - 1)) sE = __builtin_nan("");  // U left (0,1): the logarithms are undefined
  const double tE = block_sum(sE, scratch);
  const double tG = block_sum(sG, scratch);
  const double tP = block_sum(sPS, scratch);
  const double tS = block_sum(cSA, scratch);
  if (threadIdx.x == 0) {
    double* p = partDiag + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
    p[0] = tE; p[1] = tG; p[2] = tP; p[3] = tS;
  }
}


// ---------------------------------------------------------------------------
// k_fin (one block): reduce the sweep partials, Ra of row int(N/2)+1, write the
// timedata row, advance the counters and apply the energy stop rule.
// Row layout (timedata.py:9): [it, E, E2, SA, domtime, Ra, L2, PS, delt].
// Slot 4 carries time_passed; the host applies ** (1/3) (solver.py:230) with
// the same libm pow the reference's Python float uses.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(PW_THREADS) void k_fin(const T* __restrict__ U, DevConsts dc, DevState* __restrict__ st,
                                                    const double* __restrict__ partDiag, int nPart,
                                                    double* __restrict__ rows, long long rowsCap, int prepare_mode,
                                                    const double* __restrict__ partE2, int nE2, int fused) {
  __shared__ double scratch[32];
  if (!prepare_mode && st->halt) return;
  const int N = dc.N;
  double sE = 0.0, sG = 0.0, sP = 0.0, sS = 0.0;
  for (int i = threadIdx.x; i < nPart; i += PW_THREADS) {
    sE += partDiag[(size_t)i * 4 + 0];
    sG += partDiag[(size_t)i * 4 + 1];
    sP += partDiag[(size_t)i * 4 + 2];
    sS += partDiag[(size_t)i * 4 + 3];
  }
  sE = block_sum(sE, scratch);
  sG = block_sum(sG, scratch);
  sP = block_sum(sP, scratch);
  sS = block_sum(sS, scratch);
  if (fused) {
    // Fused fast-engine record: sG so far holds the column-edge terms.  np.gradient's sum of
    // squares = [4 * sum(hat_U^2 (sin^2(pi kr/N) + sin^2(pi kc/N))) + 3 * (edge terms)] / (4 delx^2):
    // central differences of the even-extended field are diagonal in the DCT-II basis, and the
    // one-sided edge rows/columns are twice the extended central ones (4x^2 = x^2 + 3x^2).
    double sp = 0.0;
    for (int i = threadIdx.x; i < nE2; i += PW_THREADS) sp += partE2[i];
    sp = block_sum(sp, scratch);
    double er = 0.0;
    for (int c = threadIdx.x; c < N; c += PW_THREADS) {
      const double d0 = (double)U[(size_t)N + c] - (double)U[c];
      const double d1 = (double)U[(size_t)(N - 1) * N + c] - (double)U[(size_t)(N - 2) * N + c];
      er += d0 * d0 + d1 * d1;
    }
    er = block_sum(er, scratch);
    sG = (4.0 * sp + 3.0 * (sG + er)) / (4.0 * dc.delx * dc.delx);
  }
  // Ra: solver.py:226-227 (row int(N/2)+1; two passes over one row)
  const int rr = N / 2 + 1;
  double s = 0.0;
  if (rr < N)
    for (int c = threadIdx.x; c < N; c += PW_THREADS) s += (double)U[(size_t)rr * N + c];
  const double rmean = block_sum(s, scratch) / (double)N;
  s = 0.0;
  if (rr < N)
    for (int c = threadIdx.x; c < N; c += PW_THREADS) s += fabs((double)U[(size_t)rr * N + c] - rmean);
  const double Ra = block_sum(s, scratch) / (double)N;

  if (threadIdx.x == 0) {
#pragma clang fp contract(off)
    const double N2 = (double)N * (double)N;
    const double L2sq = dc.L * dc.L;
    const double E2 = 0.5 * dc.Amr * dc.kappa_tilde * L2sq * (sG / N2);
    const double E = dc.Amr * L2sq * (sE / N2) + E2;
    const double PS = sP / N2;
    if (prepare_mode) {
      double* row = rows;  // row 0 of the prepare buffer
      row[0] = 0.0; row[1] = E; row[2] = E2; row[3] = 0.0; row[4] = 0.0;
      row[5] = Ra; row[6] = 0.0; row[7] = PS; row[8] = st->delt;
      st->E2_0 = E2;
      st->E2_prev = E2;
      st->tau0 = 0.0; st->t0 = 0.0;
      st->stop_reason = CHS_STOP_NONE;
      st->computed_steps = 1;
      st->halt = 0;
      st->nan_flag = (E != E || E2 != E2 || Ra != Ra || PS != PS) ? 1 : 0;
      st->rows_written = 0;
    } else {
      fin_update(dc, st, E, E2, PS, sS / N2, Ra, rows, rowsCap);
    }
  }
}

// ---------------------------------------------------------------------------
// k_step_tail (fast engine, fused pipeline; one block of 1024 threads): the record of step s
// (k_fin) and, with do_pre, the time-step control of step s+1 (k_pre) in one launch.  Every
// input is requested up front, so the block pays one memory latency instead of a dozen.
//   partDiag[nRow][4] = {sE, edge-row/column terms, sPS, cSA} and partRa from k_row_inv
//   partE2[nE2]       = spectral gradient sums from k_col
//   partMu[nMu]       = sum(mu^2) of the NEXT step's EnergieEut from k_row_inv (fused)
// ---------------------------------------------------------------------------
#define TAIL_THREADS 1024
__global__ __launch_bounds__(TAIL_THREADS) void k_step_tail(TailArgs ta, DevState* __restrict__ st) {
  __shared__ double red[TAIL_RED_DOUBLES(TAIL_THREADS)];
  if (st->halt) return;
  step_tail_body<TAIL_THREADS>(ta, st, red);
}

TailArgs chs_tail_args(const Engine* E, int set, int do_pre) {
  TailArgs ta;
  ta.enabled = 1; ta.do_pre = do_pre;
  ta.dc = E->dc;
  const bool pp = E->partSet[0][0] != nullptr && set >= 0;
  ta.partDiag = pp ? E->partSet[set][0] : E->dPartDiag;
  ta.partMu = pp ? E->partSet[set][1] : E->dPartMu;
  ta.partE2 = pp ? E->partSet[set][2] : E->dPartE2;
  ta.partRa = pp ? E->partSet[set][3] : E->dPartRa;
  ta.partColMin = E->dPartColMin;
  ta.nRow = E->nRowBlocks; ta.nE2 = E->nPartE2; ta.nMu = E->nPartMu; ta.nColMin = E->nColMinCur ? E->nColMinCur : E->nColMinBlocks;
  ta.rows = E->dRows; ta.rowsCap = E->rowsCap;
  ta.U = E->dU; ta.f32 = (E->dtype == CHS_F32) ? 1 : 0;
  return ta;
}

int chs_launch_step_tail(Engine* E, int do_pre) {
  chs_slot_begin(E, SLOT_FIN);
  k_step_tail<<<1, TAIL_THREADS, 0, E->stream>>>(chs_tail_args(E, -1, do_pre), E->dState);
  chs_slot_end(E, SLOT_FIN);
  CHS_HIP(hipGetLastError());
  return CHS_OK;
}

// ---------------------------------------------------------------------------
// k_jitter: U += jitter*(2*noise-1)   (solver.py:210-211)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(PW_THREADS) void k_jitter(T* __restrict__ U, const T* __restrict__ noise, double jitter,
                                                       size_t total, const DevState* __restrict__ st,
                                                       double* __restrict__ partSum) {
  __shared__ double scratch[32];
  if (st->halt) return;
  double sum = 0.0;  // of the perturbed field, for meanU: same grid and element order as k_jitter_pcg
  for (size_t i = (size_t)blockIdx.x * PW_THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * PW_THREADS) {
#pragma clang fp contract(off)
    const T u = (T)((double)U[i] + jitter * (2.0 * (double)noise[i] - 1.0));
    U[i] = u;
    sum += (double)u;
  }
  sum = block_sum(sum, scratch);
  if (threadIdx.x == 0) partSum[blockIdx.x] = sum;
}

// ---------------------------------------------------------------------------
// k_jitter_pcg: the same with r drawn on the device from numpy's PCG64 stream (XSL-RR 128/64:
// state <- state*MULT + inc, output = rotr64(hi ^ lo, hi >> 58), double = (out >> 11) * 2^-53;
// element i of the C-ordered array takes draw i).  Thread t handles elements t, t+TT, t+2TT, ...
// (coalesced): it jumps to draw t once (LCG jump-ahead, O(log t)) and then strides TT draws at a time
// with the precomputed jump (multTT, plusTT).
// ---------------------------------------------------------------------------
typedef unsigned __int128 chs_u128;
#define CHS_JITTER_BLOCKS_MAX 4096
#define CHS_PCG_MULT ((((chs_u128)0x2360ED051FC65DA4ULL) << 64) | (chs_u128)0x4385DF649FCCF645ULL)
__host__ __device__ inline void chs_pcg_jump(chs_u128 delta, chs_u128 inc, chs_u128& acc_mult, chs_u128& acc_plus) {
  chs_u128 cur_mult = CHS_PCG_MULT, cur_plus = inc;
  acc_mult = 1; acc_plus = 0;
  while (delta > 0) {
    if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
    cur_plus = (cur_mult + 1) * cur_plus;
    cur_mult *= cur_mult;
    delta >>= 1;
  }
}
template <typename T, bool INIT>
__global__ __launch_bounds__(PW_THREADS) void k_jitter_pcg(T* __restrict__ U, double jitter, double base, size_t total,
                                                           unsigned long long s_hi, unsigned long long s_lo,
                                                           unsigned long long i_hi, unsigned long long i_lo,
                                                           unsigned long long m_hi, unsigned long long m_lo,
                                                           unsigned long long p_hi, unsigned long long p_lo,
                                                           const DevState* __restrict__ st,
                                                           double* __restrict__ partSum) {
  __shared__ double scratch[32];
  if (!INIT && st->halt) return;
  double sum = 0.0;
  const size_t tt = (size_t)gridDim.x * PW_THREADS, t = (size_t)blockIdx.x * PW_THREADS + threadIdx.x;
  const chs_u128 inc = ((chs_u128)i_hi << 64) | i_lo;
  const chs_u128 multTT = ((chs_u128)m_hi << 64) | m_lo, plusTT = ((chs_u128)p_hi << 64) | p_lo;
  chs_u128 am, ap;
  chs_pcg_jump((chs_u128)t + 1, inc, am, ap);  // the state behind draw t (a draw steps first, then outputs)
  chs_u128 s = (((chs_u128)s_hi << 64) | s_lo) * am + ap;
  for (size_t i = t; i < total; i += tt) {
    const unsigned long long hi = (unsigned long long)(s >> 64), lo = (unsigned long long)s;
    const unsigned long long x = hi ^ lo;
    const unsigned rot = (unsigned)(hi >> 58);
    const unsigned long long o = (x >> rot) | (x << ((64u - rot) & 63u));
    const double r = (double)(o >> 11) * (1.0 / 9007199254740992.0);
    {
#pragma clang fp contract(off)
      if constexpr (INIT) U[i] = (T)(base + jitter * (r - 0.5));  // solver.py:82 (jitter = the scale here)
      else {
        const T u = (T)((double)U[i] + jitter * (2.0 * r - 1.0));
        U[i] = u;
        sum += (double)u;
      }
    }
    s = s * multTT + plusTT;
  }
  if constexpr (!INIT) {  // the mean of the perturbed field rides along (k_sum_fin finishes it)
    sum = block_sum(sum, scratch);
    if (threadIdx.x == 0) partSum[blockIdx.x] = sum;
  }
}

// One launch shape for both noise sources, so that their partial sums (and with them meanU and PS) agree bit for bit.
static int jitter_blocks(size_t total) {
  int blocks = (int)((total + PW_THREADS * 64 - 1) / ((size_t)PW_THREADS * 64));  // 64 elements per thread
  return blocks < 1 ? 1 : blocks > CHS_JITTER_BLOCKS_MAX ? CHS_JITTER_BLOCKS_MAX : blocks;
}

static int launch_pcg(Engine* E, bool init, double a, double base, const unsigned long long st[2], const unsigned long long ic[2]) {
  const size_t total = (size_t)E->N * E->N;
  const int blocks = jitter_blocks(total);
  const chs_u128 inc = ((chs_u128)ic[0] << 64) | ic[1];
  chs_u128 mtt, ptt;
  chs_pcg_jump((chs_u128)blocks * PW_THREADS, inc, mtt, ptt);
  const unsigned long long mh = (unsigned long long)(mtt >> 64), ml = (unsigned long long)mtt;
  const unsigned long long ph = (unsigned long long)(ptt >> 64), pl = (unsigned long long)ptt;
  chs_slot_begin(E, SLOT_MISC);
  if (init) {
    DISPATCH_T(E,
      (k_jitter_pcg<double, true><<<blocks, PW_THREADS, 0, E->stream>>>((double*)E->dU, a, base, total, st[0], st[1], ic[0], ic[1], mh, ml, ph, pl, E->dState, E->dPartSum)),
      (k_jitter_pcg<float, true><<<blocks, PW_THREADS, 0, E->stream>>>((float*)E->dU, a, base, total, st[0], st[1], ic[0], ic[1], mh, ml, ph, pl, E->dState, E->dPartSum)));
  } else {
    DISPATCH_T(E,
      (k_jitter_pcg<double, false><<<blocks, PW_THREADS, 0, E->stream>>>((double*)E->dU, a, base, total, st[0], st[1], ic[0], ic[1], mh, ml, ph, pl, E->dState, E->dPartSum)),
      (k_jitter_pcg<float, false><<<blocks, PW_THREADS, 0, E->stream>>>((float*)E->dU, a, base, total, st[0], st[1], ic[0], ic[1], mh, ml, ph, pl, E->dState, E->dPartSum)));
  }
  if (!init) k_sum_fin<<<1, PW_THREADS, 0, E->stream>>>(E->dPartSum, blocks, E->N, E->dState, 0);
  chs_slot_end(E, SLOT_MISC);
  CHS_HIP(hipGetLastError());
  return CHS_OK;
}

int chs_launch_init_pcg(Engine* E, double base, double scale, const unsigned long long state[2], const unsigned long long inc[2]) {
  return launch_pcg(E, true, scale, base, state, inc);
}

int chs_launch_jitter_pcg(Engine* E) {
  const int rc = launch_pcg(E, false, E->jitter, 0.0, E->pcgState, E->pcgInc);
  if (rc) return rc;
  // the generator moves on by one field per step, whether or not a stop has turned the kernel into a
  // no-op: the caller re-seeds from its own generator at the next call
  const chs_u128 state = ((chs_u128)E->pcgState[0] << 64) | E->pcgState[1];
  const chs_u128 inc = ((chs_u128)E->pcgInc[0] << 64) | E->pcgInc[1];
  chs_u128 mall, pall;
  chs_pcg_jump((chs_u128)E->N * E->N, inc, mall, pall);
  const chs_u128 next = state * mall + pall;
  E->pcgState[0] = (unsigned long long)(next >> 64); E->pcgState[1] = (unsigned long long)next;
  return CHS_OK;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
int chs_pointwise_alloc(Engine* E) {
  const int N = E->N;
  E->nBands = (N + PW_BAND - 1) / PW_BAND;
  E->nPartMu = E->nBands;
  E->nDiagBlocks = ((N + DIAG_BAND - 1) / DIAG_BAND) * ((N + PW_THREADS - 1) / PW_THREADS);
  E->nColMinBlocks = (N + PW_THREADS - 1) / PW_THREADS;
  CHS_HIP(hipMalloc(&E->dPartMu, sizeof(double) * (size_t)(E->nBands > N ? E->nBands : N)));
  CHS_HIP(hipMalloc(&E->dPartMuAux, sizeof(double) * (size_t)E->nBands));
  CHS_HIP(hipMalloc(&E->dPartDiag, sizeof(double) * 4 * (size_t)(E->nDiagBlocks > N ? E->nDiagBlocks : N)));
  CHS_HIP(hipMalloc(&E->dPartSum, sizeof(double) * (size_t)(E->nBands > CHS_JITTER_BLOCKS_MAX ? E->nBands : CHS_JITTER_BLOCKS_MAX)));
  CHS_HIP(hipMalloc(&E->dPartColMin, sizeof(double) * (size_t)((N + 31) / 32)));  // (k_colmin_slices writes N/256 entries)
  CHS_HIP(hipMalloc(&E->dPartCol, sizeof(double) * (size_t)E->nBands * N));
  return CHS_OK;
}
void chs_pointwise_free(Engine* E) {
  hipFree(E->dPartMu); hipFree(E->dPartMuAux); hipFree(E->dPartDiag); hipFree(E->dPartSum);
  hipFree(E->dPartColMin); hipFree(E->dPartCol);
  if (E->dColSlices) { hipFree(E->dColSlices); E->dColSlices = nullptr; }
}


int chs_launch_mu(Engine* E) {
  chs_slot_begin(E, SLOT_MU);
  DISPATCH_T(E,
    (k_mu<double><<<E->nBands, PW_THREADS, 0, E->stream>>>((const double*)E->dU, (double*)E->dMU, E->dc, E->dState,
                                                            E->dPartMu, E->dPartCol, 0, 0)),
    (k_mu<float><<<E->nBands, PW_THREADS, 0, E->stream>>>((const float*)E->dU, (float*)E->dMU, E->dc, E->dState,
                                                           E->dPartMu, E->dPartCol, 0, 0)));
  chs_slot_end(E, SLOT_MU);
  CHS_HIP(hipGetLastError());
  return CHS_OK;
}

static int launch_colmin(Engine* E, const void* rows, bool rows_f32, int nRows, int cs_offset, bool decide = false);

int chs_launch_mu_colsums(Engine* E, int cs_offset) {
  chs_slot_begin(E, SLOT_MISC);
  DISPATCH_T(E,
    (k_mu<double><<<E->nBands, PW_THREADS, 0, E->stream>>>((const double*)E->dU, (double*)E->dMU, E->dc, E->dState,
                                                            E->dPartMuAux, E->dPartCol, 1, cs_offset)),
    (k_mu<float><<<E->nBands, PW_THREADS, 0, E->stream>>>((const float*)E->dU, (float*)E->dMU, E->dc, E->dState,
                                                           E->dPartMuAux, E->dPartCol, 1, cs_offset)));
  const int rcm = launch_colmin(E, E->dPartCol, false, E->nBands, cs_offset);
  chs_slot_end(E, SLOT_MISC);
  if (rcm) return rcm;
  CHS_HIP(hipGetLastError());
  return CHS_OK;
}

// k_colsum_slices / k_colmin_slices: the same for the partial rows the fused row kernel writes (one per workgroup:
// up to N/2 of them, 67 MB at N=4096).  Two stages so that the array streams: CS_SLICES x N/64 blocks add up
// 64 columns over one slice of the rows each (a wavefront reads 512 contiguous bytes per row, 8 independent
// requests per thread in flight), then one thread per column adds the slices and the block takes the minimum.
// (One block per 32 columns with a rolled loop over all rows -- 128 blocks, one request in flight per thread --
// took longer than the step's two transform kernels together.)
#define CS_COLS 64
#define CS_SLICES 32
#define CS_UN 8
// STREAM: the rows are read past the caches (grids whose T and hat_U do not fit the Infinity Cache together: the
// partial rows would displace T; where everything fits, the cached read is the faster one -- profiles/r03_ab_nt.txt)
template <typename PT, bool STREAM>
__global__ __launch_bounds__(PW_THREADS) void k_colsum_slices(const PT* __restrict__ partRows, int nRows, int N,
                                                              const DevState* __restrict__ st, int adaptive,
                                                              double* __restrict__ slices, int cs_offset) {
  __shared__ double acc[PW_THREADS];
  if (st->halt) return;
  const long long cs = st->computed_steps + cs_offset;
  if (!(adaptive && cs > 500 && (cs % 2) == 0)) return;
  constexpr int NWV = PW_THREADS / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * CS_COLS + lane;
  const int per = (nRows + CS_SLICES - 1) / CS_SLICES;
  const int r0 = blockIdx.y * per, r1 = min(r0 + per, nRows);
  double s = 0.0;
  if (c < N) {
    const PT* col = partRows + c;
    for (int rb = r0 + w; rb < r1; rb += NWV * CS_UN) {
      PT x[CS_UN];
#pragma unroll
      for (int u = 0; u < CS_UN; ++u) {
        const int r = rb + u * NWV;
        const PT* src = &col[(size_t)(r < r1 ? r : r0) * N];
        x[u] = STREAM ? __builtin_nontemporal_load(src) : *src;
      }
#pragma unroll
      for (int u = 0; u < CS_UN; ++u) s += (rb + u * NWV < r1) ? (double)x[u] : 0.0;
    }
  }
  acc[threadIdx.x] = s;
  __syncthreads();
  if (w == 0 && c < N) {
    double t = acc[lane];
#pragma unroll
    for (int g = 1; g < NWV; ++g) t += acc[g * 64 + lane];
    slices[(size_t)blockIdx.y * N + c] = t;
  }
}
// `decide` (the fused pipeline with nothing armed that a tail would have to decide: chs_fast_step): the block that finishes
// last also takes the minimum over the blocks and works out the coming step's coefficients lam1 / lam2 -- the step-size rule of
// solver.py:184-193 is a function of the current delt and of this minimum alone -- so that the next k_col finds them in the
// state when it starts and needs NO gate; the riding bookkeeping arrives at the same values (same function, same inputs) and
// writes them again with the rest of the state.  Arrival ticket in the state, release / acquire at agent scope.
__global__ __launch_bounds__(PW_THREADS) void k_colmin_slices(const double* __restrict__ slices, int N,
                                                              DevState* __restrict__ st, int adaptive,
                                                              double* __restrict__ partColMin, int cs_offset, DevConsts dc, int decide) {
  __shared__ double scratch[32];
  if (st->halt) return;
  const long long cs = st->computed_steps + cs_offset;
  if (!(adaptive && cs > 500 && (cs % 2) == 0)) return;
  const int c = blockIdx.x * PW_THREADS + threadIdx.x;
  double s = 1.0e300;
  if (c < N) {
    double x[CS_SLICES];
#pragma unroll
    for (int b = 0; b < CS_SLICES; ++b) x[b] = slices[(size_t)b * N + c];
    s = 0.0;
#pragma unroll
    for (int b = 0; b < CS_SLICES; ++b) s += x[b];
  }
  const double m = block_min(s, scratch);
  if (threadIdx.x == 0) {
    partColMin[blockIdx.x] = m;
    if (decide) {
      // release: this block's minimum is visible at agent scope before its arrival is (explicit waits on both sides of the
      // write-back, as in step_tail_body: hipcc may drop the fence's own wait, cdna_hip_programming.md Guideline 16)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int arrived = atomicAdd(&st->colmin_ticket, 1);
      if (arrived == (int)gridDim.x - 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the other blocks' minima (read with agent-scope loads below)
        double mm = 1.0e300;
        for (int b = 0; b < (int)gridDim.x; ++b)
          mm = fmin(mm, __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&partColMin[b]),
                                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
        DevState e = *st;
        pre_update(dc, &e, 0.0, true, mm);
        st->lam1 = e.lam1; st->lam2 = e.lam2;
        st->colmin_ticket = 0;   // (for the next launch: kernel boundaries order it)
      }
    }
  }
}

// partColMin[0 .. nColMinBlocks) <- block minima of the column sums of rows[nRows][N] (doubles, or floats: rows_f32)
static int launch_colmin(Engine* E, const void* rows, bool rows_f32, int nRows, int cs_offset, bool decide) {
  if (!E->dColSlices) CHS_HIP(hipMalloc(&E->dColSlices, sizeof(double) * (size_t)CS_SLICES * E->N));
  const dim3 g1((E->N + CS_COLS - 1) / CS_COLS, CS_SLICES);
  const bool stream = chs_grid_exceeds_cache((size_t)E->N, E->dtype == CHS_F32 ? 4 : 8);
  if (rows_f32) {
    if (stream) k_colsum_slices<float, true><<<g1, PW_THREADS, 0, E->stream>>>((const float*)rows, nRows, E->N, E->dState, E->dc.adaptive_time, E->dColSlices, cs_offset);
    else k_colsum_slices<float, false><<<g1, PW_THREADS, 0, E->stream>>>((const float*)rows, nRows, E->N, E->dState, E->dc.adaptive_time, E->dColSlices, cs_offset);
  } else {
    if (stream) k_colsum_slices<double, true><<<g1, PW_THREADS, 0, E->stream>>>((const double*)rows, nRows, E->N, E->dState, E->dc.adaptive_time, E->dColSlices, cs_offset);
    else k_colsum_slices<double, false><<<g1, PW_THREADS, 0, E->stream>>>((const double*)rows, nRows, E->N, E->dState, E->dc.adaptive_time, E->dColSlices, cs_offset);
  }
  k_colmin_slices<<<E->nColMinBlocks, PW_THREADS, 0, E->stream>>>(E->dColSlices, E->N, E->dState, E->dc.adaptive_time,
                                                                  E->dPartColMin, cs_offset, E->dc, decide ? 1 : 0);
  E->nColMinCur = E->nColMinBlocks;
  return CHS_OK;
}

int chs_launch_colmin_rows(Engine* E, int cs_offset, bool decide) {
  chs_slot_begin(E, SLOT_MISC);
  const int rc = launch_colmin(E, E->dPartColRows, E->dtype == CHS_F32, E->nRowBlocks, cs_offset, decide);
  chs_slot_end(E, SLOT_MISC);
  if (rc) return rc;
  CHS_HIP(hipGetLastError());
  return CHS_OK;
}

int chs_launch_pre(Engine* E) {
  chs_slot_begin(E, SLOT_PRE);
  if (E->dc.adaptive_time && E->engine == CHS_ENGINE_DIRECT) {
    const int rcm = launch_colmin(E, E->dPartCol, false, E->nBands, 0);
    if (rcm) { chs_slot_end(E, SLOT_PRE); return rcm; }
  }
  k_pre<<<1, PW_THREADS, 0, E->stream>>>(E->dc, E->dState, E->dPartMu, E->nPartMu, E->dPartColMin,
                                         E->nColMinBlocks);
  chs_slot_end(E, SLOT_PRE);
  CHS_HIP(hipGetLastError());
  return CHS_OK;
}

int chs_launch_spectral(Engine* E, const void* hmu) {
  const size_t total = (size_t)E->N * E->N;
  int blocks = (int)((total + PW_THREADS - 1) / PW_THREADS);
  if (blocks > 4096) blocks = 4096;
  chs_slot_begin(E, SLOT_SPEC);
  DISPATCH_T(E,
    (k_spectral<double><<<blocks, PW_THREADS, 0, E->stream>>>((double*)E->dHat, (const double*)hmu, E->dLambda, E->N,
                                                               E->dState)),
    (k_spectral<float><<<blocks, PW_THREADS, 0, E->stream>>>((float*)E->dHat, (const float*)hmu, E->dLambda, E->N,
                                                              E->dState)));
  chs_slot_end(E, SLOT_SPEC);
  CHS_HIP(hipGetLastError());
  return CHS_OK;
}

int chs_launch_sum(Engine* E, int ignore_halt) {
  chs_slot_begin(E, SLOT_MISC);
  DISPATCH_T(E,
    (k_sum<double><<<E->nBands, PW_THREADS, 0, E->stream>>>((const double*)E->dU, E->N, E->dState, E->dPartSum, ignore_halt)),
    (k_sum<float><<<E->nBands, PW_THREADS, 0, E->stream>>>((const float*)E->dU, E->N, E->dState, E->dPartSum, ignore_halt)));
  k_sum_fin<<<1, PW_THREADS, 0, E->stream>>>(E->dPartSum, E->nBands, E->N, E->dState, ignore_halt);
  chs_slot_end(E, SLOT_MISC);
  CHS_HIP(hipGetLastError());
  return CHS_OK;
}

int chs_launch_diag(Engine* E, int ignore_halt) {
  const dim3 grid((E->N + DIAG_BAND - 1) / DIAG_BAND, (E->N + PW_THREADS - 1) / PW_THREADS);
  chs_slot_begin(E, SLOT_DIAG);
  DISPATCH_T(E,
    (k_diag<double><<<grid, PW_THREADS, 0, E->stream>>>((const double*)E->dU, E->dc, E->dState,
                                                                   E->dPartDiag, ignore_halt)),
    (k_diag<float><<<grid, PW_THREADS, 0, E->stream>>>((const float*)E->dU, E->dc, E->dState,
                                                                  E->dPartDiag, ignore_halt)));
  chs_slot_end(E, SLOT_DIAG);
  CHS_HIP(hipGetLastError());
  return CHS_OK;
}

int chs_launch_fin(Engine* E, int prepare_mode, int fused) {
  chs_slot_begin(E, SLOT_FIN);
  DISPATCH_T(E,
    (k_fin<double><<<1, PW_THREADS, 0, E->stream>>>((const double*)E->dU, E->dc, E->dState, E->dPartDiag,
                                                     fused ? E->nRowBlocks : E->nDiagBlocks, E->dRows, E->rowsCap,
                                                     prepare_mode, E->dPartE2, E->nPartE2, fused)),
    (k_fin<float><<<1, PW_THREADS, 0, E->stream>>>((const float*)E->dU, E->dc, E->dState, E->dPartDiag,
                                                    fused ? E->nRowBlocks : E->nDiagBlocks, E->dRows, E->rowsCap,
                                                    prepare_mode, E->dPartE2, E->nPartE2, fused)));
  chs_slot_end(E, SLOT_FIN);
  CHS_HIP(hipGetLastError());
  return CHS_OK;
}

int chs_launch_jitter(Engine* E) {
  const size_t total = (size_t)E->N * E->N;
  const int blocks = jitter_blocks(total);
  chs_slot_begin(E, SLOT_MISC);
  DISPATCH_T(E,
    (k_jitter<double><<<blocks, PW_THREADS, 0, E->stream>>>((double*)E->dU, (const double*)E->dNoise, E->jitter, total,
                                                             E->dState, E->dPartSum)),
    (k_jitter<float><<<blocks, PW_THREADS, 0, E->stream>>>((float*)E->dU, (const float*)E->dNoise, E->jitter, total,
                                                            E->dState, E->dPartSum)));
  k_sum_fin<<<1, PW_THREADS, 0, E->stream>>>(E->dPartSum, blocks, E->N, E->dState, 0);
  chs_slot_end(E, SLOT_MISC);
  CHS_HIP(hipGetLastError());
  return CHS_OK;
}

// ---------------------------------------------------------------------------
// chs_test_math: device math primitives, elementwise (accuracy tests only)
// ---------------------------------------------------------------------------
__global__ void k_test_math(int which, const double* __restrict__ a, const double* __restrict__ b,
                            double* __restrict__ out, long long n) {
  __shared__ double2 ltab[CHS_LOGTAB_N];
  for (int t = threadIdx.x; t < CHS_LOGTAB_N; t += blockDim.x) ltab[t] = reinterpret_cast<const double2*>(chs_log_table)[t];
  __syncthreads();
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r;
  unsigned dom = 0;
  if (which == 5) { r = chs_log_unit_tab_f64(a[i], ltab, dom); if (dom > (unsigned)(CHS_LOGTAB_N - 1)) r = __builtin_nan(""); }
  else if (which == 0) r = chs_log_f64(a[i]);
  else if (which == 1) r = chs_log_ratio_f64(a[i], b[i]);
  else if (which == 4) r = chs_log_pos_f64(a[i]);
  else if (which == 2) r = chs_mu<double>(a[i], b[0], b[1], b[2], b[3]);
  else r = chs_energy_density<double>(a[i], b[0], b[1], b[2], b[3]);
  out[i] = r;
}

extern "C" int chs_test_math(int device, int which, const double* a, const double* b, double* out, int64_t n) {
  if (!a || !out || n <= 0 || which < 0 || which > 5) { chs_set_error("chs_test_math: bad argument"); return CHS_EINVAL; }
  CHS_HIP(hipSetDevice(device));
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  const size_t nb_b = (which == 1) ? (size_t)n : 4;
  CHS_HIP(hipMalloc(&da, sizeof(double) * n));
  CHS_HIP(hipMalloc(&db, sizeof(double) * nb_b));
  CHS_HIP(hipMalloc(&dout, sizeof(double) * n));
  CHS_HIP(hipMemcpy(da, a, sizeof(double) * n, hipMemcpyHostToDevice));
  if (b) CHS_HIP(hipMemcpy(db, b, sizeof(double) * nb_b, hipMemcpyHostToDevice));
  k_test_math<<<(unsigned)((n + 255) / 256), 256>>>(which, da, db, dout, (long long)n);
  CHS_HIP(hipGetLastError());
  CHS_HIP(hipDeviceSynchronize());
  CHS_HIP(hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost));
  hipFree(da); hipFree(db); hipFree(dout);
  return CHS_OK;
}
