// chs_fast.hip -- fast transform engine, host side: plans, twiddle tables and the launch sequence of a
// timestep.  The kernels are templates in chs_fast_kernels.h; their instantiations live in
// chs_fast_f64.hip / chs_fast_f32.hip.
#define CHS_FAST_MAIN_TU
#include "chs_fast_kernels.h"
#ifdef CHS_STAMPS
// diagnostic build with phase stamps: one translation unit (the stamp buffer is a device variable)
#define CHS_FAST_UNITY_INCLUDE
#include "chs_fast_f64.hip"
#include "chs_fast_f32.hip"
#endif

bool chs_fast_supported(int N, int dtype) {
  (void)dtype;  // both element types: N = 2^k, 128 <= N <= 8192
  return N == 128 || N == 256 || N == 512 || N == 1024 || N == 2048 || N == 4096 || N == 8192;
}

static void twiddle(long double num, long double den, long double& c, long double& s) {
  // exp(-2 pi i num/den) = c - i s
  static const long double PI = 3.14159265358979323846264338327950288419716939937510L;
  const long double a = 2.0L * PI * (num / den);
  c = cosl(a); s = sinl(a);
}

template <typename T>
static int build_tables(Engine* E, FastPlan* P) {
  const int N = P->N, M = N / 2, L1 = M / P->R0, L2 = L1 / P->RA, L3 = L2 / P->RB;
  std::vector<T> h;
  auto push = [&](long double r, long double i) { h.push_back((T)r); h.push_back((T)i); };
  P->off_tw0 = h.size();
  for (int k = 1; k < P->R0; ++k)
    for (int m = 0; m < L1; ++m) {
      long double c, s;
      twiddle((long double)(((long long)m * k) % M), (long double)M, c, s);
      push(c, -s);
    }
  P->off_twa = h.size();
  for (int k = 1; k < P->RA; ++k)
    for (int m = 0; m < L2; ++m) {
      long double c, s;
      twiddle((long double)(((long long)m * k) % L1), (long double)L1, c, s);
      push(c, -s);
    }
  P->off_twb = h.size();
  for (int k = 1; k < P->RB; ++k)
    for (int m = 0; m < L3; ++m) {
      long double c, s;
      twiddle((long double)(((long long)m * k) % L2), (long double)L2, c, s);
      push(c, -s);
    }
  push(0, 0);  // keep every table pointer inside the allocation even when a pass is absent
  static const long double PI = 3.14159265358979323846264338327950288419716939937510L;
  P->off_wp = h.size();
  for (int kk = 0; kk <= M; ++kk) {  // -i exp(-2 pi i kk/N) = -sin(th) - i cos(th)
    const long double th = 2.0L * PI * (long double)kk / (long double)N;
    push(-sinl(th), -cosl(th));
  }
  auto Tk = [&](int k, long double& r, long double& i) {  // s_k/2 exp(-i pi k/(2N))
    const long double s = (k == 0) ? sqrtl(1.0L / N) : sqrtl(2.0L / N);
    const long double ph = PI * (long double)k / (2.0L * N);
    r = 0.5L * s * cosl(ph); i = -0.5L * s * sinl(ph);
  };
  P->off_t1 = h.size();
  for (int kk = 0; kk <= M; ++kk) { long double r, i; Tk(kk, r, i); push(r, i); }
  P->off_t2 = h.size();
  for (int kk = 0; kk <= M; ++kk) { long double r, i; Tk(M - kk, r, i); push(r, -i); }
  // interleaved {lambda_k, sin^2(pi k/N)}: the eigenvalue table of utils.py:35 next to the weights of
  // the spectral form of np.gradient's sum of squares -- one 16-byte load per coefficient
  std::vector<double> lam(N), sq(2 * (size_t)N);
  CHS_HIP(hipMemcpy(lam.data(), E->dLambda, sizeof(double) * N, hipMemcpyDeviceToHost));
  for (int k = 0; k < N; ++k) {
    const long double v = sinl(PI * (long double)k / (long double)N);
    sq[2 * k] = lam[k];
    sq[2 * k + 1] = (double)(v * v);
  }
  // per-slot entries of the same two quantities for the packed fp32 spectral stage (FTables::lam4 / sin4): the four
  // coefficients {j, N-j, M-j, M+j} of slot j = 0..M (out-of-range ones only occur in slots that are never live),
  // and entry M+1 for the special lane's own slot {0, M/2, M, 3M/2}; 16-byte aligned inside the table buffer
  while (h.size() % 4) h.push_back((T)0);
  P->off_lam4 = h.size();
  auto quad = [&](int j, int t) { const int q[4] = {j, N - j, M - j, M + j}; return q[t] < 0 ? 0 : (q[t] > N - 1 ? N - 1 : q[t]); };
  const int own[4] = {0, M / 2, M, 3 * (M / 2)};
  for (int j = 0; j <= M + 1; ++j)
    for (int t = 0; t < 4; ++t) h.push_back((T)lam[j <= M ? quad(j, t) : own[t]]);
  P->off_sin4 = h.size();
  for (int j = 0; j <= M + 1; ++j)
    for (int t = 0; t < 4; ++t) h.push_back((T)sq[2 * (j <= M ? quad(j, t) : own[t]) + 1]);
  CHS_HIP(hipMalloc(&P->tables, h.size() * sizeof(T)));
  CHS_HIP(hipMemcpy(P->tables, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  CHS_HIP(hipMalloc(&E->dSinSq, sizeof(double) * 2 * N));
  CHS_HIP(hipMemcpy(E->dSinSq, sq.data(), sizeof(double) * 2 * N, hipMemcpyHostToDevice));
  return CHS_OK;
}

int chs_fast_init(Engine* E) {
  FastPlan* P = new FastPlan();
  const bool ok = (E->dtype == CHS_F32) ? chs_fast_bind_f32(E->N, P) : chs_fast_bind_f64(E->N, P);
  if (!ok) { delete P; chs_set_error("fast engine: unsupported N"); return CHS_EINVAL; }
  E->dTw = P;
  int rc = (E->dtype == CHS_F32) ? build_tables<float>(E, P) : build_tables<double>(E, P);
  if (rc) return rc;
  if ((rc = P->init(E))) return rc;
  // the row kernels write one partial record per workgroup, k_col one per column tile
  E->nRowBlocks = E->N / (P->threads / P->G);
  E->nPartMu = E->nRowBlocks;
  E->nPartE2 = P->col_tiles;
  CHS_HIP(hipMalloc(&E->dPartE2, sizeof(double) * (size_t)E->nPartE2));
  CHS_HIP(hipMalloc(&E->dPartRa, sizeof(double) * 8));
  // second set of partial sums for the deferred tail (chs_fast_step)
  E->partSet[0][0] = E->dPartDiag; E->partSet[0][1] = E->dPartMu; E->partSet[0][2] = E->dPartE2; E->partSet[0][3] = E->dPartRa;
  CHS_HIP(hipMalloc(&E->partSet[1][0], sizeof(double) * 4 * (size_t)(E->nDiagBlocks > E->N ? E->nDiagBlocks : E->N)));
  CHS_HIP(hipMalloc(&E->partSet[1][1], sizeof(double) * (size_t)(E->nBands > E->N ? E->nBands : E->N)));
  CHS_HIP(hipMalloc(&E->partSet[1][2], sizeof(double) * (size_t)E->nPartE2));
  CHS_HIP(hipMalloc(&E->partSet[1][3], sizeof(double) * 8));
  return chs_fast_rearm(E);
}

// What depends on the constants of a run rather than on (N, dtype): called by chs_fast_init and when a pooled engine
// is taken into use again (chs_create).
int chs_fast_rearm(Engine* E) {
  E->fusedAdapt = E->adaptOk && getenv("CHS_ADAPT_SWEEP") == nullptr;
  { const char* e = getenv("CHS_ADAPT_SPARSE"); E->adaptSparse = !(e && e[0] == '0'); }   // (read when an engine is taken into use)
  // (the coefficients of a firing step published ahead of the record, chs_tail.h: measured equal -- what a gated k_col pays is its
  // own check of the gate, not the wait for the decision -- so it stays an experiment switch: CHS_GATE_EARLY=1)
  { const char* e = getenv("CHS_GATE_EARLY"); E->gateEarly = (e && e[0] == '1'); }
  { const char* e = getenv("CHS_LAM_BY_COLMIN"); E->lamByColmin = !(e && e[0] == '0'); }
  if (E->dc.adaptive_time && E->fusedAdapt && !E->dPartColRows)
    CHS_HIP(hipMalloc(&E->dPartColRows, E->esz * (size_t)E->nRowBlocks * E->N));
  if (E->partSet[0][0]) {
    E->dPartDiag = E->partSet[0][0]; E->dPartMu = E->partSet[0][1];
    E->dPartE2 = E->partSet[0][2]; E->dPartRa = E->partSet[0][3];
  }
  E->parity = 0; E->tailSet = 0;
  E->tailDeferred = false; E->tailGated = false; E->preRider = false;
  E->stepCount = 0; E->storeU = true;
  return CHS_OK;
}

void chs_fast_free(Engine* E) {
  FastPlan* P = (FastPlan*)E->dTw;
  if (!P) return;
  if (P->tables) hipFree(P->tables);
  if (E->dSinSq) hipFree(E->dSinSq);
  // the "current" pointers alias one of the two sets: hand set 0 back to the generic owners
  if (E->partSet[0][0]) {
    E->dPartDiag = E->partSet[0][0]; E->dPartMu = E->partSet[0][1];
    E->dPartE2 = E->partSet[0][2]; E->dPartRa = E->partSet[0][3];
    for (int i = 0; i < 4; ++i) { hipFree(E->partSet[1][i]); E->partSet[1][i] = nullptr; E->partSet[0][i] = nullptr; }
  }
  if (E->dPartColRows) hipFree(E->dPartColRows);
  E->dPartColRows = nullptr;
  if (E->dPartE2) hipFree(E->dPartE2);
  if (E->dPartRa) hipFree(E->dPartRa);
  E->dPartRa = nullptr;
  E->dSinSq = nullptr; E->dPartE2 = nullptr;
  delete P;
  E->dTw = nullptr;
}

int chs_fast_dct2d(Engine* E, const void* in, void* out, bool inverse) {
  FastPlan* P = (FastPlan*)E->dTw;
  int rc;
  if (!inverse) {
    if ((rc = P->row_fwd(E, in, E->dT1, ROW_FWD_PLAIN))) return rc;
    return P->col(E, MODE_FWD_NATURAL, E->dT1, nullptr, E->dHat, out);
  }
  if ((rc = P->col(E, MODE_INV_NATURAL, nullptr, E->dT1, E->dHat, (void*)in))) return rc;
  return P->row_inv(E, ROW_INV_PLAIN, E->dT1, out, nullptr);
}

// U <- idctn(hat_U) from k_col's native order: the field of the last completed step when the fused
// row kernel has not been writing U (chs_fast_step) and a stop ended the call early.
int chs_fast_recover_u(Engine* E) {
  FastPlan* P = (FastPlan*)E->dTw;
  int rc;
  if ((rc = P->col(E, MODE_INV_NATIVE, nullptr, E->dT1, E->dHat, nullptr))) return rc;
  return P->row_inv(E, ROW_INV_PLAIN, E->dT1, E->dU, nullptr);
}

// hat_U <- dctn(U) in k_col's native order (solver.py:159)
int chs_fast_enter(Engine* E) {
  FastPlan* P = (FastPlan*)E->dTw;
  int rc;
  if ((rc = P->row_fwd(E, E->dU, E->dT1, ROW_FWD_PLAIN))) return rc;
  return P->col(E, MODE_FWD_NATIVE, E->dT1, nullptr, E->dHat, nullptr);
}

// T1 <- row DCT-II of EnergieEut(U): what the fused row kernel of the previous step would
// have left behind; needed once per solve_or_resume call.
// the set of partial-sum buffers the kernels of the running step write to
static void select_partial_set(Engine* E) {
  if (!E->partSet[0][0]) return;
  const int par = E->parity;
  E->dPartDiag = E->partSet[par][0]; E->dPartMu = E->partSet[par][1];
  E->dPartE2 = E->partSet[par][2]; E->dPartRa = E->partSet[par][3];
}

#ifndef CHS_ENTRY_REVERSE
#define CHS_ENTRY_REVERSE 0
#endif
// Entry of a call on the fused pipeline: hat_U <- dctn(U) (solver.py:159) and the prologue below with
// one sweep of U instead of two (k_row_fwd2), the row transform of U parked in the idle T2 buffer.
int chs_fast_enter_fused(Engine* E) {
  FastPlan* P = (FastPlan*)E->dTw;
  E->tailDeferred = false;
  select_partial_set(E);
  int rc;
  chs_slot_begin(E, SLOT_MU);
  rc = P->row_fwd2(E, E->dU, E->dT2, E->dT1);
  chs_slot_end(E, SLOT_MU);
  if (rc) return rc;
  // k_col<FWD_NATIVE> writes hat_U tile by tile in ascending order.  The first step's k_col walks the tiles in ASCENDING
  // order too (0; 1 = descending: what was written last is read first -- measured 2.3 % slower on a literal 20-step call,
  // profiles/r04_ab_dma.txt), whatever the parity of the steps of earlier calls; then the directions alternate (CHS_COL_ZIGZAG)
  if (CHS_ENTRY_REVERSE >= 0) E->stepCount = CHS_ENTRY_REVERSE;
  return P->col(E, MODE_FWD_NATIVE, E->dT2, nullptr, E->dHat, nullptr);
}

// Entry of a literal call that finds the first step's operand on the device: the previous call's last step was the fused
// row kernel (it stored U in full AND left T1 = the row transform of EnergieEut(U) with its sum of squares, exactly what
// the steps inside a call hand to each other), nothing has touched the field since.  hat_U = dctn(U) is recomputed here,
// literally (solver.py:159): row pass of U (streamed) into the idle T2 buffer, column pass into hat_U.  What is NOT
// recomputed is EnergieEut(U) and its row transform -- a function of the unchanged U, bit for bit what k_row_fwd2 would
// produce again.
int chs_fast_enter_hat(Engine* E) {
  FastPlan* P = (FastPlan*)E->dTw;
  int rc;
  static const int mode = [] { const char* e = getenv("CHS_ENTRY_HAT_MODE"); return e ? atoi(e) : (int)ROW_FWD_PLAIN; }();   // (cached stores: 86 us; non-temporal ones, ROW_FWD_STREAM: 190-210 us)
  if ((rc = P->row_fwd(E, E->dU, E->dT2, mode))) return rc;
  if (CHS_ENTRY_REVERSE >= 0) E->stepCount = CHS_ENTRY_REVERSE;
  return P->col(E, MODE_FWD_NATIVE, E->dT2, nullptr, E->dHat, nullptr);
}

int chs_fast_prologue(Engine* E) {
  FastPlan* P = (FastPlan*)E->dTw;
  E->tailDeferred = false;
  select_partial_set(E);  // sum(mu^2) must land where the first step's k_pre looks for it
  chs_slot_begin(E, SLOT_MU);
  const int rc = P->row_fwd(E, E->dU, E->dT1, ROW_FWD_POINTWISE);
  chs_slot_end(E, SLOT_MU);
  return rc;
}

// One timestep on the fused pipeline.  On entry T1 holds the row transform of
// EnergieEut(U_k) and partMu its sum of squares; on exit U_(k+1) is in HBM, the pointwise
// diagnostics partials of U_(k+1) are ready for k_fin and, with fuse_next, T1/partMu are
// ready for the next step.
// full_sim, fixed time step, no time limit: the tail of step s decides nothing the column pass of
// step s+1 needs.  It is deferred and rides as one extra workgroup in k_col of step s+1 -- no launch
// of its own, nothing waits for it.  Only NaN can stop such a run (one kernel later; the field is
// unspecified then anyway).  The partial sums alternate between two sets.
// Stop rules on the small grids (CHS_HAT_FLIP_MAX_N, fixed time step): k_col reads hat_U from one buffer and writes
// the other, alternating from step to step, so the tail can stay deferred -- when it stops the run, the buffer
// the carrying k_col READ is the state of the last completed step (run_steps points dHat at it and rebuilds U).
// The tiles of a small grid reach a gate before the riding tail has decided (N=512: 26.5 against 23.9 us/step);
// at N=4096 the gate costs 1 % and a third 134 MB array would not fit beside T and hat_U in the Infinity Cache.
#ifndef CHS_HAT_FLIP_MAX_N
#define CHS_HAT_FLIP_MAX_N 2048
#endif
static bool hat_flip(const Engine* E) {
  return !E->dc.adaptive_time && (!E->dc.full_sim || E->dc.time_limit_s > 0.0) && E->N <= CHS_HAT_FLIP_MAX_N &&
         !E->timer.on && E->partSet[0][0] != nullptr;
}
static bool can_defer_tail(const Engine* E) {
  return !E->dc.adaptive_time && ((!(E->dc.time_limit_s > 0.0) && E->dc.full_sim) || hat_flip(E)) && !E->timer.on &&
         E->partSet[0][0] != nullptr;
}
// Between the steps of a call nothing reads U from HBM (the adaptive step included, once the fused
// row kernel adds up its integrand), so the fused row kernel keeps it in registers.  Whenever a stop
// can end the call early (energy rule, time limit) the tail runs in stream order right behind the
// row kernel: hat_U is then still that of the last completed step and run_steps() rebuilds
// U = idctn(hat_U) once (chs_fast_recover_u).
static bool fused_adaptive(const Engine* E) {
  return E->dc.adaptive_time && E->fusedAdapt && E->dPartColRows != nullptr;
}
static bool can_skip_u(const Engine* E) {
  return !E->dc.adaptive_time || fused_adaptive(E);
}

int chs_fast_step(Engine* E, bool first, bool last) {
  FastPlan* P = (FastPlan*)E->dTw;
  int rc;
  const bool defer = can_defer_tail(E);
  E->storeU = !can_skip_u(E);
  select_partial_set(E);
  if (first) {
    // time-step control of the first step of the call; later steps get it from the tail.
    // The prologue wrote sum(mu^2) into the set that was current then: fold it in here.
    if (E->dc.adaptive_time) {
      if ((rc = chs_launch_mu_colsums(E, 0))) return rc;
    }
    if (defer) {
      E->preRider = true;  // rides as the extra workgroup of this step's k_col (no launch of its own)
    } else {
      if ((rc = chs_launch_pre(E))) return rc;
    }
  }
  chs_slot_begin(E, SLOT_SPEC);
  // T2 (columns inverted) overwrites T1 in place: every workgroup of k_col reads exactly the part
  // of the tile it later writes, every workgroup of the fused row kernel likewise for its rows --
  // one array less in the per-step working set (T + hat_U = 268 MB next to a 256 MB Infinity Cache)
  void* T2 = CHS_ALIAS_T ? E->dT1 : E->dT2;
  if (hat_flip(E)) {
    if (!E->dHat2) CHS_HIP(hipMalloc(&E->dHat2, (size_t)E->N * E->N * E->esz));
    E->hatFlip = true;
    rc = P->col(E, MODE_STEP, E->dT1, T2, E->dHat, E->dHat2);  // + the previous step's deferred tail
    void* t = E->dHat; E->dHat = E->dHat2; E->dHat2 = t;      // the next step reads what this one writes
  } else {
    rc = P->col(E, MODE_STEP, E->dT1, T2, E->dHat, nullptr);  // + the previous step's deferred tail
  }
  chs_slot_end(E, SLOT_SPEC);
  E->tailDeferred = false;
  if (rc) return rc;
  chs_slot_begin(E, SLOT_INV);
  const bool fa = fused_adaptive(E);
  // Adaptive step with the host able to follow the step counter: the step-size rule fires on every second step beyond
  // step 500 only (solver.py:177), and only THOSE steps need the column sums reduced and the next k_col gated (its
  // coefficients change).  On the others -- with nothing else armed that a tail could decide -- the reduction launches are
  // not issued at all (they used to return at once: two empty launches per step) and the bookkeeping rides ungated.
  // (CHS_ADAPT_SPARSE=0: as before round 4.)
  bool fires = true;
  if (E->dc.adaptive_time && E->csHost >= 0 && E->adaptSparse) {
    const long long cs_next = E->csHost + 1;          // the counter behind this step's record (chs_tail.h: cs_next)
    fires = (cs_next > 500 && (cs_next % 2) == 0);
  }
  if (E->csHost >= 0) E->csHost += 1;                 // (a halted run issues no-ops; the call's end sets the true value)
  if (last && E->keepResident) {
    // the last step of the call leaves the field in HBM like ROW_INV_DIAG, and with it what the first column
    // pass of a following call needs (T1, sum(mu^2)): that call then starts without an entry pass (run_steps)
    const bool store = E->storeU;
    E->storeU = true;
    rc = P->row_inv(E, ROW_INV_FUSED, T2, E->dU, E->dT1);
    E->storeU = store;
  } else {
    rc = P->row_inv(E, last ? ROW_INV_DIAG : (fa ? ROW_INV_FUSED_ADAPT : ROW_INV_FUSED), T2, E->dU, E->dT1);
  }
  chs_slot_end(E, SLOT_INV);
  if (rc) return rc;
  // A firing step with nothing else armed: the reduction's last block works out the coming step's coefficients itself
  // (k_colmin_slices, `decide`), so the next k_col needs no gate either -- its tiles read them from the state as ever
  const bool lam_by_colmin = !last && fa && fires && E->csHost >= 0 && E->adaptSparse && E->lamByColmin && E->dc.full_sim &&
                             !(E->dc.time_limit_s > 0.0) && !E->timer.on;
  if (!last && E->dc.adaptive_time && (fires || !fa)) {
    // column sums of the adaptive-step integrand of the NEXT step (solver.py:183); the record of
    // this step has not advanced computed_steps yet, hence the offset
    if ((rc = fa ? chs_launch_colmin_rows(E, 1, lam_by_colmin) : chs_launch_mu_colsums(E, 1))) return rc;
  }
  // Stop rules armed (energy rule, time limit) or an adaptive time step: the bookkeeping still rides in the
  // next k_col, whose other workgroups wait for its decision in front of their first global write (gated
  // tail, gate_wait) -- no 14 us one-block launch per step.  A run being profiled keeps the separate launch.
  const bool gate = !defer && !E->timer.on && E->partSet[0][0] != nullptr;
  if (last || (!defer && !gate)) return chs_launch_step_tail(E, last ? 0 : 1);
  // (an adaptive step whose rule does not fire decides nothing the next k_col needs, unless a stop rule is armed)
  const bool quiet = (!fires || lam_by_colmin) && fa && E->dc.full_sim && !(E->dc.time_limit_s > 0.0);
  E->tailDeferred = true;
  E->tailGated = gate && !quiet;
  // ... and one whose rule does fire, with nothing else armed, lets the tiles have their coefficients early (chs_tail.h)
  E->tailEarly = E->tailGated && fires && fa && E->dc.full_sim && !(E->dc.time_limit_s > 0.0) && E->csHost >= 0 && E->adaptSparse && E->gateEarly;
  E->tailSet = E->parity;
  E->parity ^= 1;
  return CHS_OK;
}

// Jitter path (solver.py:210-211 perturbs U between the inverse transform and the record):
// no fusion across the perturbation; the caller adds noise, k_sum, k_diag, k_fin.
int chs_fast_step_unfused(Engine* E) {
  FastPlan* P = (FastPlan*)E->dTw;
  int rc;
  if ((rc = chs_fast_prologue(E))) return rc;
  if (E->dc.adaptive_time) {
    if ((rc = chs_launch_mu_colsums(E, 0))) return rc;
  }
  if ((rc = chs_launch_pre(E))) return rc;
  chs_slot_begin(E, SLOT_SPEC);
  rc = P->col(E, MODE_STEP, E->dT1, E->dT2, E->dHat, nullptr);
  chs_slot_end(E, SLOT_SPEC);
  if (rc) return rc;
  chs_slot_begin(E, SLOT_INV);
  rc = P->row_inv(E, ROW_INV_PLAIN, E->dT2, E->dU, nullptr);
  chs_slot_end(E, SLOT_INV);
  return rc;
}
