// chs_tail.h -- the per-step bookkeeping of the fused pipeline: the record of step s (the thread-0
// part of k_fin) and the time-step control of step s+1 (the thread-0 part of k_pre), written as a
// workgroup-level device function so that it can run as its own one-block kernel (k_step_tail) or
// as one extra workgroup of the next step's k_col (chs_fast.hip).
#pragma once
#include "chs_common.h"

// thread-0 part of k_pre: solver.py:184-199,225 and utils.py:41-42
__device__ __forceinline__ void pre_update(const DevConsts& dc, DevState* st, double musq, bool adapt, double delt_dyn) {
#pragma clang fp contract(off)
  const double N2 = (double)dc.N * (double)dc.N;
  st->L2_cur = sqrt(musq) / N2;  // solver.py:225
  double delt = st->delt;
  if (adapt) {  // solver.py:184-188
    const double delt_new = fmax(dc.delt0, delt_dyn);
    if (delt_new / delt > 1.15)
      delt = 0.75 * delt + 0.25 * delt_new;
    else
      delt = delt_new;
    st->delt = delt;
    st->delt_coef = delt;  // solver.py:189-193: the grids are regenerated for the new step
  }
  // The grids stand for delt_coef, not delt: a resumed adaptive run keeps its adapted self.delt for the
  // time bookkeeping below but works with the grids of params.delt until the step is re-evaluated
  // (solver.py:154-155 against 189-193)
  const double lam1 = st->delt_coef / dc.delx2;  // utils.py:41-42
  st->lam1 = lam1;
  st->lam2 = dc.kappa_tilde * lam1 / dc.delx2;
  const double tds = st->time_delta_sum + delt;  // solver.py:195-199
  st->time_delta_sum = tds;
  const double tp = tds / dc.M_tilde;
  st->time_passed = tp;
  if (dc.time_limit_s > 0.0 && tp > dc.time_limit_s) {
    st->stop_reason = CHS_STOP_TIME_LIMIT;
    st->halt = 1;
  }
}

// thread-0 part of the per-step record: solver.py:230-249 ; timedata.py:8-10,51-63
__device__ __forceinline__ void fin_update(const DevConsts& dc, DevState* st, double E, double E2, double PS,
                                           double SA, double Ra, double* __restrict__ rows, long long rowsCap) {
  const long long k = st->rows_written;
  {
    // a ring: run_steps() copies the rows out batch by batch, long before a slot comes round again
    double* row = rows + (k % rowsCap) * 9;
    row[0] = (double)st->computed_steps; row[1] = E; row[2] = E2; row[3] = SA;
    row[4] = st->time_passed; row[5] = Ra; row[6] = st->L2_cur; row[7] = PS; row[8] = st->delt;
  }
  st->rows_written = k + 1;
  const double L2v = st->L2_cur, tp = st->time_passed;
  if (E != E || E2 != E2 || Ra != Ra || PS != PS || L2v != L2v || tp != tp || SA != SA) {
    st->nan_flag = 1;  // timedata.py:10 fires before computed_steps += 1
    st->halt = 1;
  } else {
    const long long cs = st->computed_steps + 1;  // solver.py:240
    st->computed_steps = cs;
    // solver.py:242-249 ; timedata.py:63
    if (!st->skip_check && st->E2_prev > E2 && E2 > st->E2_0) {
      st->tau0 = (double)cs;
      st->t0 = st->time_passed;
      if (!dc.full_sim) {
        st->stop_reason = CHS_STOP_ENERGY;
        st->halt = 1;
      } else {
        st->skip_check = 1;
      }
    }
    st->E2_prev = E2;
  }
}

// Inputs of one tail.
//   partDiag[nRow][4] = {sE, column-edge terms, sPS, cSA} and partRa from k_row_inv
//   partE2[nE2]       = spectral gradient sums from k_col
//   partMu[nMu]       = sum(mu^2) of the NEXT step's EnergieEut from k_row_inv (fused)
struct TailArgs {
  int enabled = 0, do_pre = 0;
  int gate = 0;                   // publish the decision to the other workgroups of the carrying k_col launch
  unsigned long long seq = 0;     // ... under this sequence number (DevState::decided)
  int gate_spins = 1 << 20;       // polls (s_sleep 16 between them, ~1 us each) before a waiting workgroup gives up
  int withhold = 0;               // test hook (CHS_TEST_GATE_WITHHOLD): the decision is never published
  int early = 0;                  // gated for the adaptive step's sake alone (no stop rule armed): the coefficients of the coming
                                  // step are published as soon as they are known, ahead of the record (step_tail_body)
  int pre_only = 0;  // first step of a call: no record yet, only the time-step control of the coming step
  int reverse = 0;  // (k_col rider, not a tail input) walk the column tiles in descending order this step
  DevConsts dc;
  const double* partDiag = nullptr; const double* partE2 = nullptr; const double* partMu = nullptr;
  const double* partColMin = nullptr; const double* partRa = nullptr;
  int nRow = 0, nE2 = 0, nMu = 0, nColMin = 0;
  const void* U = nullptr;  // the field of the recorded step: rows 0, 1, N-2, N-1 are read (np.gradient row edges)
  int f32 = 0;              // element type of U
  double* rows = nullptr;
  long long rowsCap = 0;
};

#define TAIL_NV 8
#ifndef TAIL_UN
#define TAIL_UN 8  // independent requests per thread and batch
#endif
#define TAIL_RED_DOUBLES(THREADS) (((THREADS) / 64 + 1) * (TAIL_NV + 1))

// One workgroup of THREADS threads; `red` = TAIL_RED_DOUBLES(THREADS) doubles of LDS.  Every input
// is requested up front, so the block pays one memory latency instead of a dozen.
template <int THREADS>
__device__ __forceinline__ void step_tail_body(const TailArgs& ta, DevState* __restrict__ st, double* red) {
  constexpr int NW = THREADS / 64, NV = TAIL_NV;
  double* tot = red + NW * (NV + 1);
  const DevConsts& dc = ta.dc;
  const int N = dc.N, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int do_pre = ta.do_pre;
  const long long cs_next = st->computed_steps + 1;
  const bool adapt = do_pre && dc.adaptive_time && cs_next > 500 && (cs_next % 2) == 0;
  double v[NV] = {0, 0, 0, 0, 0, 0, 0, 0};  // sE, sEdge, sPS, cSA, spectral, musq, -, -
  double mn = 1.0e300;
  // the state and Ra are thread 0's operands at the very end: asked for here, with everything else, they cost
  // no round trip of their own behind the reductions (nobody else writes the fields used below in the meantime)
  DevState loc;
  double Ra = 0.0;
  if (tid == 0) {
    loc = *st;
    if (!ta.pre_only) Ra = ta.partRa[0];
  }
  // EARLY DECISION.  When the launch is gated only because the step-size rule fires (full_sim, no time limit: nothing but
  // lam1 / lam2 of the coming step is what the tiles wait for), those two are a function of the state and of the column
  // minimum alone -- a few dozen numbers.  They are computed and published HERE, in front of the record's reductions (the
  // partial sums of thousands of workgroups, four edge rows of U: ~28 us under a streaming launch, where the tiles reach the
  // gate after ~12 us); the regular pre_update below arrives at the same two values bit for bit (same function, same inputs)
  // and writes them again with the rest of the state.  A NaN found later stops the run one kernel later, as it does for a
  // fixed time step (the field is unspecified then either way).
  bool published = false;
  if (ta.early && ta.gate && adapt && !ta.pre_only) {
    double m = 1.0e300;
    for (int i = tid; i < ta.nColMin; i += THREADS) m = fmin(m, ta.partColMin[i]);
    m = wave_min(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < NW; ++w) m = fmin(m, red[w]);
      DevState e = loc;
      pre_update(dc, &e, 0.0, true, m);
      st->lam1 = e.lam1; st->lam2 = e.lam2;
      if (!ta.withhold) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&st->decided, ta.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();   // (`red` is used again below)
    published = true;
  }
  if (ta.pre_only) {
    // sum(mu^2) of the entry kernel -> L2 and the time bookkeeping of the first step (k_pre's work; fixed
    // time step: nothing k_col of this step reads changes)
    for (int i0 = tid; i0 < ta.nMu; i0 += THREADS * TAIL_UN) {
      double x[TAIL_UN];
#pragma unroll
      for (int u = 0; u < TAIL_UN; ++u) { const int i = i0 + u * THREADS; x[u] = ta.partMu[i < ta.nMu ? i : 0]; }
#pragma unroll
      for (int u = 0; u < TAIL_UN; ++u) v[5] += (i0 + u * THREADS < ta.nMu) ? x[u] : 0.0;
    }
    v[5] = wave_sum(v[5]);
    if (lane == 0) red[wave] = v[5];
    __syncthreads();
    if (tid == 0) {
      double t = red[0];
      for (int w = 1; w < NW; ++w) t += red[w];
      pre_update(dc, &loc, t, false, 0.0);
      st->delt = loc.delt; st->delt_coef = loc.delt_coef; st->time_delta_sum = loc.time_delta_sum;
      st->time_passed = loc.time_passed; st->L2_cur = loc.L2_cur; st->lam1 = loc.lam1; st->lam2 = loc.lam2;
      st->stop_reason = loc.stop_reason; st->halt = loc.halt;
    }
    return;
  }
  // Batches of TAIL_UN independent requests per thread (a rolled loop would wait for every element before it
  // asks for the next: ~40 dependent round trips, 14 us as a kernel of its own and twice that while the tiles
  // of the carrying k_col stream in); out-of-range slots read element 0 and add zero, the order of the sums
  // is that of the plain loops.
  constexpr int UN = TAIL_UN;
  for (int i0 = tid; i0 < ta.nRow; i0 += THREADS * UN) {
    double2 a[UN], b[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = i0 + u * THREADS;
      const double2* q = reinterpret_cast<const double2*>(ta.partDiag + (size_t)(i < ta.nRow ? i : 0) * 4);
      a[u] = q[0]; b[u] = q[1];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const bool ok = i0 + u * THREADS < ta.nRow;
      v[0] += ok ? a[u].x : 0.0; v[1] += ok ? a[u].y : 0.0; v[2] += ok ? b[u].x : 0.0; v[3] += ok ? b[u].y : 0.0;
    }
  }
  auto sum_batched = [&](const double* __restrict__ src, int n, double& acc) {
    for (int i0 = tid; i0 < n; i0 += THREADS * UN) {
      double x[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) { const int i = i0 + u * THREADS; x[u] = src[i < n ? i : 0]; }
#pragma unroll
      for (int u = 0; u < UN; ++u) acc += (i0 + u * THREADS < n) ? x[u] : 0.0;
    }
  };
  sum_batched(ta.partE2, ta.nE2, v[4]);
  // one-sided row edges of np.gradient (solver.py:213-217): (U[1,c]-U[0,c])^2 + (U[N-1,c]-U[N-2,c])^2;
  // the row kernel adds the column edges, k_col the spectral bulk
  if (ta.U) {
    auto edges = [&](auto* uu) {  // (one instantiation per element type: no type branch inside the batches)
      const size_t last = (size_t)(N - 2) * N;
      for (int c0 = tid; c0 < N; c0 += THREADS * UN) {
        double a0[UN], a1[UN], b0[UN], b1[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int c = (c0 + u * THREADS < N) ? c0 + u * THREADS : 0;
          a0[u] = (double)uu[c]; a1[u] = (double)uu[N + c]; b0[u] = (double)uu[last + c]; b1[u] = (double)uu[last + N + c];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const double d0 = a1[u] - a0[u], d1 = b1[u] - b0[u];
          v[1] += (c0 + u * THREADS < N) ? d0 * d0 + d1 * d1 : 0.0;
        }
      }
    };
    if (ta.f32) edges((const float*)ta.U);
    else edges((const double*)ta.U);
  }
  if (do_pre) sum_batched(ta.partMu, ta.nMu, v[5]);
  if (adapt) {
    for (int i0 = tid; i0 < ta.nColMin; i0 += THREADS * UN) {
      double x[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) { const int i = i0 + u * THREADS; x[u] = ta.partColMin[i < ta.nColMin ? i : 0]; }
#pragma unroll
      for (int u = 0; u < UN; ++u) mn = (i0 + u * THREADS < ta.nColMin) ? fmin(mn, x[u]) : mn;
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  mn = wave_min(mn);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave * (NV + 1) + i] = v[i];
    red[wave * (NV + 1) + NV] = mn;
  }
  __syncthreads();
  if (tid <= NV) {
    double t = red[tid];
    for (int w = 1; w < NW; ++w) t = (tid < NV) ? t + red[w * (NV + 1) + tid] : fmin(t, red[w * (NV + 1) + tid]);
    tot[tid] = t;
  }
  __syncthreads();
  if (tid == 0) {
#pragma clang fp contract(off)
    const double N2 = (double)N * (double)N;
    const double L2sq = dc.L * dc.L;
    // np.gradient's sum of squares from the spectrum + the one-sided edge rows/columns (see k_fin)
    const double sG = (4.0 * tot[4] + 3.0 * tot[1]) / (4.0 * dc.delx * dc.delx);
    const double E2 = 0.5 * dc.Amr * dc.kappa_tilde * L2sq * (sG / N2);
    const double E = dc.Amr * L2sq * (tot[0] / N2) + E2;
    // work on a register copy of the state (loaded at the top): one wide load and one wide store instead of a
    // chain of dependent global round trips
    fin_update(dc, &loc, E, E2, tot[2] / N2, tot[3] / N2, Ra, ta.rows, ta.rowsCap);
    if (do_pre && !loc.halt) pre_update(dc, &loc, tot[5], adapt, tot[NV]);
    // write back everything except meanU: when the tail rides in k_col of the NEXT step, that
    // kernel may be storing it at this very moment (it is the only other writer of the state)
    st->delt = loc.delt; st->delt_coef = loc.delt_coef; st->time_delta_sum = loc.time_delta_sum; st->time_passed = loc.time_passed;
    st->tau0 = loc.tau0; st->t0 = loc.t0; st->E2_0 = loc.E2_0; st->E2_prev = loc.E2_prev;
    st->L2_cur = loc.L2_cur; st->lam1 = loc.lam1; st->lam2 = loc.lam2;
    st->computed_steps = loc.computed_steps; st->rows_written = loc.rows_written;
    st->skip_check = loc.skip_check; st->stop_reason = loc.stop_reason; st->nan_flag = loc.nan_flag;
    st->halt = loc.halt;
    if (ta.gate && !ta.withhold && !published) {
      // everything above becomes visible at agent scope before the sequence number does: the waiting
      // workgroups (any CU, any XCD) read halt / lam1 / lam2 with agent-scope (L1-bypassing) loads behind it.
      // The explicit waits stand on both sides of the write-back: the state stores have left this wavefront
      // before it starts, and it has completed before the flag goes out (hipcc may drop the fence's own wait
      // when it can prove the wavefront's vmcnt scoreboard empty, cdna_hip_programming.md Guideline 16).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&st->decided, ta.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The other side of the gate: called by every thread of a workgroup of the carrying k_col launch in front of
// its first global write.  Returns the published halt flag and the coefficients of this step.  One lane
// polls (the bookkeeping workgroup is block 0, dispatched first, and waits for nobody: it always gets
// there; by the time a tile workgroup has staged and transformed its tile it has long finished), bounded:
// a workgroup that gives up raises gate_timeout AND halt, so that every later kernel of the call is a no-op
// and the call ends with an error instead of stepping on with a column pass that did not happen.
// Ordering (MI355X_MICROARCH.md, inter-workgroup visibility): the producer wrote the state with plain stores,
// drained them, released at agent scope and then stored the sequence number.  The consumer's polling lane polls
// relaxed, issues ONE agent-scope acquire once the poll has matched and then reads the payload itself (agent-scope
// loads); the other wavefronts get the values through LDS behind the workgroup barrier.  `box` = 4 doubles of LDS.
// `early` = what the polling lane read from the sequence word in front of the forward passes (a load whose round trip ran under
// them): when it already shows this launch's number -- every workgroup but those of the first round -- no poll is needed at all.
__device__ __forceinline__ int gate_wait(DevState* __restrict__ st, unsigned long long seq, int spins, double* box,
                                         double& lam1, double& lam2, unsigned long long early = ~0ull) {
  if (threadIdx.x == 0) {
    int it = 0, tmo = 0;
    while (early != seq && __hip_atomic_load(&st->decided, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) {
      __builtin_amdgcn_s_sleep(16);
      // (a sibling that has already given up spares the others the full wait)
      if (++it > spins || ((it & 63) == 0 && __hip_atomic_load(&st->gate_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        tmo = 1;
        __hip_atomic_store(&st->gate_timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->halt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    // ONE agent-scope acquire in the polling lane, once per workgroup and step, behind the matched poll: the payload
    // loads below are ordered behind the poll by the memory model, not by what gfx950 happens to do with sc1 loads
    // (gated modes only -- stop rules armed or an adaptive step)
#ifndef CHS_GATE_ACQUIRE
#define CHS_GATE_ACQUIRE 1
#endif
#if CHS_GATE_ACQUIRE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
    asm volatile("" ::: "memory");
#endif
    const int halt = __hip_atomic_load(&st->halt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    box[0] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&st->lam1), __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT));
    box[1] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&st->lam2), __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT));
    box[2] = (halt | tmo) ? 1.0 : 0.0;
  }
  __syncthreads();
  lam1 = box[0];
  lam2 = box[1];
  return box[2] != 0.0;
}
