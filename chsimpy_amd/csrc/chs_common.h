// chs_common.h -- internal declarations shared by the HIP translation units.
// gfx950 (MI355X) only: 64-wide wavefronts are assumed throughout.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "chs_hip.h"

#define CHS_WAVE 64

// Cache policy of the step loop: T and hat_U together against the 256 MiB Infinity Cache of an
// MI355X.  Where they exceed it, what is touched once per step (hat_U, the partial rows of the adaptive step) is moved
// with non-temporal accesses and T, which both kernels read and write, keeps the cache.
__host__ __device__ constexpr bool chs_grid_exceeds_cache(size_t N, size_t elem_bytes) {
  return 2 * N * N * elem_bytes > ((size_t)256 << 20);
}

// ---------------------------------------------------------------------------
// Device-resident scalar state of one simulation: what chsimpy/solver.py keeps
// in `self.*` / `self.solution.*` between iterations, plus the reduction
// results handed from one kernel of a step to the next.
// ---------------------------------------------------------------------------
struct DevState {
  double delt;            // solver.py:54,185-188
  double time_delta_sum;  // solver.py:51,195
  double time_passed;     // solver.py:52,196
  double tau0, t0;        // solver.py:243-244
  double E2_0, E2_prev;   // timedata.py:63 operands
  double L2_cur;          // ||EnergieEut||_F / N^2 of the running step (solver.py:225)
  double meanU;           // mean of the field the next statistics sweep refers to (solver.py:223)
  double lam1, lam2;      // utils.py:41-42 for delt_coef
  double delt_coef;       // the delt the coefficient grids CHeig/Seig stand for: params.delt from the entry of
                          // every solve_or_resume call (solver.py:154-155 reloads solution.Seig/CHeig, which
                          // solution.py:52-55 built once) until the adaptive step regenerates them (189-193)
  long long computed_steps;  // solver.py:134,240
  long long rows_written;    // rows produced by the running chs_step_n call
  int skip_check;            // solver.py:50,249
  int stop_reason;           // CHS_STOP_*
  int nan_flag;              // timedata.py:10
  int halt;                  // != 0: every later kernel of the call is a no-op
  // Gated tail (chs_fast.hip, k_col<MODE_STEP>): the sequence number of the last bookkeeping that rode in
  // a k_col as its extra workgroup and has been published (agent-scope release); the other workgroups of
  // that launch wait for it in front of their spectral stage.  gate_timeout: a wait gave up (never seen).
  unsigned long long decided;
  int gate_timeout;
  int colmin_ticket;      // arrivals of k_colmin_slices' blocks when the last of them decides the coming step's coefficients
};

// Read-only scalars, passed to kernels by value.
struct DevConsts {
  int N;
  int adaptive_time;
  int full_sim;
  int pad_;
  double RT, BRT, B, A0, A1, Amr, kappa_tilde, L, delx, delx2;
  double delt0;  // params.delt (lower bound of the adaptive step, solver.py:184)
  double delt_max, M_tilde, threshold, time_limit_s;
  double invN2;  // 1/N^2
};

// ---------------------------------------------------------------------------
// Deterministic block reductions (fixed tree: DPP/shuffle inside a wave, LDS
// across waves).  Callers pass a __shared__ scratch of >= 32 doubles.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
  return v;
}
// Returns the block total in every thread.  All threads of the block must call.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < nw; ++w) t += scratch[w];
  return t;
}
__device__ __forceinline__ double block_min(double v, double* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_min(v);
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double t = scratch[0];
  for (int w = 1; w < nw; ++w) t = fmin(t, scratch[w]);
  return t;
}

// Wavefront and lane number WITHOUT keeping threadIdx.x alive in a VGPR across a register-heavy kernel.  A kernel at its
// register limit spills such a long-lived index, and a 4-byte spill reload sits in the in-order vmcnt queue behind every
// store issued before it.  The wavefront's number goes to an SGPR once; the lane number is two v_mbcnt wherever it is
// needed (volatile: never carried from one use to the next either).  One-dimensional blocks of whole wavefronts only.
__device__ __forceinline__ int chs_wave_id() {
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  __builtin_assume(w >= 0 && w < 16);   // (what is derived from it stays unsigned 32-bit offset arithmetic)
  return w;
}
__device__ __forceinline__ int chs_lane_id() {
  int x;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
  __builtin_assume(x >= 0 && x < 64);
  return x;
}
// wave_sum with the lane number taken afresh: __shfl_down computes its own lane number with a pure builtin, which the
// compiler hoists out of loops and keeps (or spills) like any other loop invariant.  Same tree, same sums in lane 0.
__device__ __forceinline__ double wave_sum_fresh(double v) {
  const int ln = chs_lane_id();
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int src = ((ln + off) & 63) << 2;
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(src, (int)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(src, (int)(b >> 32));
    v += __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
  }
  return v;  // valid in lane 0
}
// block_sum with the caller's wavefront number (SGPR) instead of threadIdx.x
__device__ __forceinline__ double block_sum_w(double v, double* scratch, int wave, int nw) {
  v = wave_sum_fresh(v);
  __syncthreads();
  if (chs_lane_id() == 0) scratch[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < nw; ++w) t += scratch[w];
  return t;
}
// block_sum_store in two halves: `begin` right where the values are complete (wavefront sums into `red`: no register
// holds them any longer -- in the fused row kernel that is five accumulators less across the whole forward transform),
// `end` behind a barrier at the end of the kernel (wavefront 0's lane 0 adds up and gets out[]; returns true there).
template <int NV, bool FRESH = true>
__device__ __forceinline__ void block_reduce_begin(const double (&v)[NV], double* red /* >= NW*NV */, int wave) {
  double w[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) w[i] = FRESH ? wave_sum_fresh(v[i]) : wave_sum(v[i]);
  if (chs_lane_id() == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave * NV + i] = w[i];
  }
}
template <int NV, int NW>
__device__ __forceinline__ bool block_reduce_end(const double* red, double* __restrict__ out, int wave) {
  __syncthreads();
  const bool first = (wave == 0) && (chs_lane_id() == 0);
  if (first) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double t = red[i];
#pragma unroll
      for (int w = 1; w < NW; ++w) t += red[w * NV + i];
      out[i] = t;
    }
  }
  return first;
}

// NV per-thread values -> block totals (NW waves per block), left in out[0..NV) of thread 0.
// One barrier, no loops with run-time trip counts: meant for the END of a register-heavy
// kernel, where control flow around live register arrays would cost spills.
template <int NV, int NW>
__device__ __forceinline__ void block_sum_store(const double (&v)[NV], double* red /* >= NW*NV */,
                                                double* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double w[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) w[i] = wave_sum(v[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave * NV + i] = w[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double t = red[i];
#pragma unroll
      for (int w = 1; w < NW; ++w) t += red[w * NV + i];
      out[i] = t;
    }
  }
}

// ---------------------------------------------------------------------------
// Host-side engine object behind chs_handle.
// ---------------------------------------------------------------------------
struct Engine;

// Per-step kernel slots for chs_profile_steps / chs_kernel_name.
struct StepTimer {
  bool on = false;
  double ms[CHS_NKERNELS];
  long long calls[CHS_NKERNELS];
  struct Span { int slot; hipEvent_t a, b; };
  std::vector<Span> pending;     // recorded, not yet harvested
  std::vector<hipEvent_t> pool;  // free events
  hipEvent_t cur = nullptr;      // start event of the open span
};

#define CHS_STAGE_BYTES (8u << 20)
struct Engine {
  // two pinned staging chunks of this handle for field up/downloads (allocated at the first transfer)
  void* stageBuf[2] = {nullptr, nullptr};
  hipEvent_t stageEv[2] = {nullptr, nullptr};
  chs_consts hc;       // as given
  DevConsts dc;        // device view
  int engine = 0;      // resolved CHS_ENGINE_*
  int N = 0;
  int dtype = CHS_F64;
  size_t esz = 8;
  hipStream_t stream = nullptr;
  bool prepared = false;
  bool have_U = false;
  bool hat_valid = false;  // dHat matches dU's history (CHS_STEP_CARRY_HAT)
  bool resident = false;   // ... and T1 / partMu hold the row transform of EnergieEut(U) and its sum of squares: the
                           // next call continues without an entry pass (the last fused step left them, chs_fast_step)
  bool lamByColmin = true;    // CHS_LAM_BY_COLMIN (chs_fast_rearm): the reduction's last block sets the coming step's coefficients
  bool gateEarly = false;     // CHS_GATE_EARLY=1 (chs_fast_rearm): experiment, measured equal
  bool tailEarly = false;     // the gated bookkeeping of the coming k_col publishes its coefficients ahead of the record
  bool adaptSparse = true;    // CHS_ADAPT_SPARSE (chs_fast_rearm)
  long long csHost = -1;      // the device's computed_steps as the host can follow it (prepare / set_state / end of a call, +1 per
                              // issued step): lets the adaptive path issue the step-size machinery only on the steps whose rule
                              // fires (chs_fast_step); -1 = not known
  bool stateCached = false;   // hState[0] is the device state as the last call left it (chs_get_state without a round trip)
  bool keepResident = false;  // this call's last step runs the fused row kernel so that the next call can continue

  // device buffers (element type per dtype)
  void* dU = nullptr;      // field U, row-major N x N
  void* dMU = nullptr;     // EnergieEut (direct engine) / scratch
  void* dT1 = nullptr;     // transform scratch
  void* dT2 = nullptr;     // transform scratch
  void* dHat = nullptr;    // hat_U (direct: natural order; fast: engine-native order)
  void* dSlab = nullptr;   // one allocation holding dT1 | dHat (chs_create)
  void* dHat2 = nullptr;   // second hat_U buffer of the small grids' stop-rule runs (chs_fast_step), allocated on demand
  void* dHatCall = nullptr;  // ... the buffer hat_U was in when the running call began
  bool hatFlip = false;    // ... this call alternates the two
  void* dNoise = nullptr;  // jitter noise (optional)
  double jitter = 0.0;
  double* dLambda = nullptr;   // lam_i, N doubles
  void* dD = nullptr;          // direct engine: orthonormal DCT-II matrix D[k][n]
  void* dTw = nullptr;         // fast engine: twiddle tables
  DevState* dState = nullptr;
  double* dRows = nullptr;     // timedata rows of the running call
  long long rowsCap = 0;
  double* dPartMu = nullptr;   // per-block sum(mu^2)
  double* dPartCol = nullptr;  // per-band column sums for the adaptive step
  double* dPartColMin = nullptr;
  double* dPartDiag = nullptr; // per-block [sE, sG, sPS, cSA]
  double* dPartSum = nullptr;  // per-block sum(U)
  double* dPartE2 = nullptr;   // per-column-tile spectral gradient sums (fast engine)
  double* dSinSq = nullptr;    // sin^2(pi k/N), k = 0..N-1 (fast engine)
  double* dPartRa = nullptr;   // Ra of the step, written by the row kernel (fast engine)
  int nPartE2 = 0;
  int nRowBlocks = 0;          // workgroups of the fast row kernels (diag partials)
  int nBands = 0;              // row bands used by the pointwise kernels
  int nPartMu = 0;             // number of sum(mu^2) partials the running engine produces
  double* dPartMuAux = nullptr; // scratch partials of the column-sum-only sweep
  int nDiagBlocks = 0;
  int nColMinBlocks = 0;
  int nColMinCur = 0;  // entries of dPartColMin the last column-sum launch wrote

  // deferred tail: the bookkeeping of step s rides as one extra workgroup in k_col of step s+1;
  // the partial sums ping-pong between two sets so that step s+1 does not overwrite what it reads
  double* partSet[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};  // Diag, Mu, E2, Ra
  int parity = 0;
  bool tailDeferred = false;  // the tail of the previous step is still to run
  bool tailGated = false;     // ... and the other workgroups of that k_col wait for its decision (stop rules, adaptive dt)
  unsigned long long gateSeq = 0;
  bool testGateWithhold = false;  // test hook (CHS_TEST_GATE_WITHHOLD=1, read by chs_step_n in builds with -DCHS_TEST_HOOKS=1): gated tails never publish
  bool preRider = false;      // the first step's time-step control rides in its k_col (deferred-tail mode)
  unsigned stepCount = 0;     // k_col<MODE_STEP> launches so far (tile walk direction alternates)
  // jitter noise generated on the device: numpy's PCG64 stream continued from the host generator's state
  bool jitterPcg = false;
  unsigned long long pcgState[2] = {0, 0}, pcgInc[2] = {0, 0};  // {hi, lo}
  bool adaptOk = false;       // the configuration has the fused adaptive row kernel
  bool fusedAdapt = false;    // the fused row kernel adds up the adaptive-step integrand itself
  void* dPartColRows = nullptr;    // [nRowBlocks][N] partial column sums of that integrand, in the engine's element type
  double* dColSlices = nullptr;    // [CS_SLICES][N] first stage of their reduction (chs_launch_colmin_rows)
  bool storeU = true;         // the fused row kernel writes U on intermediate steps (chs_fast_step)
  int tailSet = 0;            // ... on this partial set
  hipEvent_t evA = nullptr, evB = nullptr;
  // pinned host mirror of the device state: slot 0 = the state at the end of a call, slots 1..4 = the
  // polls behind the batches of a long call (run_steps)
  DevState* hState = nullptr;
  double* hRows = nullptr;    // pinned staging of the rows of a short call
  hipEvent_t evPoll[4] = {nullptr, nullptr, nullptr, nullptr};
  int batchSteps = 1024;      // steps issued between two looks at the device's halt flag (run_steps)
  double lastStepMs = 0.0;
  StepTimer timer;
  std::vector<double> hostTmp;
  std::vector<double> hLambda;  // the eigenvalue table the engine was created with (engine pool: chs_create)
};

void chs_set_error(const std::string& s);
int chs_hip_fail(hipError_t e, const char* what, const char* file, int line);

#define CHS_HIP(call)                                                    \
  do {                                                                   \
    hipError_t e__ = (call);                                             \
    if (e__ != hipSuccess) return chs_hip_fail(e__, #call, __FILE__, __LINE__); \
  } while (0)

// kernel-slot bracket used by the launchers
void chs_slot_begin(Engine* E, int slot);
void chs_slot_end(Engine* E, int slot);

// slots
enum {
  SLOT_MU = 0,      // direct: pointwise EnergieEut           | fast: row pass forward (mu + DCT-II rows)
  SLOT_PRE = 1,     // k_pre (+ the column-sum minimum): L2, adaptive dt, time bookkeeping
  SLOT_FWD = 2,     // direct: the two forward products       | fast: (unused)
  SLOT_SPEC = 3,    // direct: spectral update                | fast: column pass (DCT-II, update, DCT-III)
  SLOT_INV = 4,     // direct: the two inverse products       | fast: row pass inverse (DCT-III rows)
  SLOT_DIAG = 5,    // energy / statistics sweep
  SLOT_FIN = 6,     // k_fin: timedata row, stop rule
  SLOT_MISC = 7
};

// ---- pointwise / reduction launchers (chs_pointwise.hip) -------------------
int chs_launch_mu(Engine* E);                 // dU -> dMU, partials
int chs_launch_mu_colsums(Engine* E, int cs_offset);
int chs_fast_rearm(Engine* E);
int chs_launch_colmin_rows(Engine* E, int cs_offset, bool decide = false);  // min over the columns of sum(dPartColRows)  // dU -> column-sum minimum of the adaptive-step integrand only
struct TailArgs;
TailArgs chs_tail_args(const Engine* E, int set, int do_pre);  // set < 0: the current partial-sum pointers
int chs_launch_step_tail(Engine* E, int do_pre);  // fused pipeline: record of step s + time-step control of step s+1
int chs_launch_pre(Engine* E);                // partials -> state (L2, delt, time)
int chs_launch_call_begin(Engine* E);         // entry of a solve_or_resume call: re-arm the loop, coefficients of params.delt
int chs_launch_spectral(Engine* E, const void* hmu);  // dHat <- (dHat + Seig*hmu)/CHeig (natural order)
int chs_launch_sum(Engine* E, int ignore_halt);  // meanU <- mean(dU)
int chs_launch_diag(Engine* E, int ignore_halt);  // dU -> diag partials
int chs_launch_fin(Engine* E, int prepare_mode, int fused = 0);
int chs_launch_jitter(Engine* E);
int chs_launch_jitter_pcg(Engine* E);
int chs_launch_init_pcg(Engine* E, double base, double scale, const unsigned long long state[2], const unsigned long long inc[2]);  // U += jitter*(2*r-1), r from the PCG64 stream; advances E->pcgState by N*N
int chs_pointwise_alloc(Engine* E);
void chs_pointwise_free(Engine* E);

// ---- direct engine (chs_direct.hip) ----------------------------------------
int chs_direct_init(Engine* E);
void chs_direct_free(Engine* E);
int chs_direct_dct2d(Engine* E, const void* in, void* out, void* tmp, bool inverse);

// ---- fast engine (chs_fast.hip) ---------------------------------------------
bool chs_fast_supported(int N, int dtype);
int chs_fast_recover_u(Engine* E);  // U <- idctn(hat_U): the field of the last completed step
int chs_fast_init(Engine* E);
void chs_fast_free(Engine* E);
int chs_fast_dct2d(Engine* E, const void* in, void* out, bool inverse);  // natural in/out (tests)
int chs_fast_enter(Engine* E);   // hat_U <- dctn(U) in engine-native order (solver.py:159)
int chs_fast_enter_fused(Engine* E);       // both of them with one sweep of U
int chs_fast_enter_hat(Engine* E);         // hat_U <- dctn(U) alone: the first step's operand T1 is still on the device
int chs_fast_prologue(Engine* E);          // T1 <- row DCT of EnergieEut(U) for the first step of a call
int chs_fast_step(Engine* E, bool first, bool last); // [k_pre,] k_col, fused row kernel, k_step_tail
int chs_fast_step_unfused(Engine* E);      // jitter path: every kernel separate, U complete in HBM
