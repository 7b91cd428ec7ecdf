// chs_api.hip -- the C ABI of include/chs_hip.h: handle lifetime, host<->device
// traffic and the per-timestep launch sequence (chsimpy/solver.py:84-252).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "chs_common.h"

static thread_local std::string g_err;

void chs_set_error(const std::string& s) { g_err = s; }
int chs_hip_fail(hipError_t e, const char* what, const char* file, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d in `%s`", (int)e, hipGetErrorString(e), file, line, what);
  g_err = buf;
  return CHS_EHIP;
}
extern "C" const char* chs_last_error(void) { return g_err.c_str(); }
// what this library was built from: sha256 of csrc/* + include/* and the extra flags of the build (chsimpy_amd/_build.py
// passes -DCHS_PROVENANCE; the loader compares it with the tree, chsimpy_amd/_lib.py)
#ifndef CHS_PROVENANCE
#define CHS_PROVENANCE "CHS_SRC_HASH=unknown;CHS_FLAGS=;"
#endif
extern "C" const char* chs_version(void) { return "chsimpy_amd 0.4 (gfx950) " CHS_PROVENANCE; }

// ---------------------------------------------------------------------------
// kernel-slot timing (chs_profile_steps)
// ---------------------------------------------------------------------------
static hipEvent_t timer_event(StepTimer& t) {
  if (!t.pool.empty()) { hipEvent_t e = t.pool.back(); t.pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  hipEventCreate(&e);
  return e;
}
void chs_slot_begin(Engine* E, int slot) {
  (void)slot;
  if (!E->timer.on) return;
  E->timer.cur = timer_event(E->timer);
  hipEventRecord(E->timer.cur, E->stream);
}
void chs_slot_end(Engine* E, int slot) {
  if (!E->timer.on) return;
  hipEvent_t b = timer_event(E->timer);
  hipEventRecord(b, E->stream);
  E->timer.pending.push_back({slot, E->timer.cur, b});
  E->timer.cur = nullptr;
}
static void timer_harvest(Engine* E) {
  hipStreamSynchronize(E->stream);
  for (auto& s : E->timer.pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      E->timer.ms[s.slot] += ms;
      E->timer.calls[s.slot] += 1;
    }
    E->timer.pool.push_back(s.a);
    E->timer.pool.push_back(s.b);
  }
  E->timer.pending.clear();
}

// ---------------------------------------------------------------------------
#define CHS_ROWS_STAGED 4096  // rows of the pinned staging buffer for short calls (run_steps)
#define CHS_ROWS_RING 65536  // rows of the device ring (4.7 MB), a multiple of the batch size of run_steps

// timedata rows of the running call: one allocation for the handle's lifetime, used as a ring (run_steps)
static int ensure_rows(Engine* E) {
  if (E->dRows) return CHS_OK;
  CHS_HIP(hipMalloc(&E->dRows, sizeof(double) * 9 * (size_t)CHS_ROWS_RING));
  E->rowsCap = CHS_ROWS_RING;
  return CHS_OK;
}

static void free_engine(Engine* E) {
  if (!E) return;
  hipSetDevice(E->hc.device);
  if (E->engine == CHS_ENGINE_DIRECT) chs_direct_free(E);
  if (E->engine == CHS_ENGINE_FAST) chs_fast_free(E);
  chs_pointwise_free(E);
  hipFree(E->dU); hipFree(E->dMU); hipFree(E->dT2);
  if (E->dSlab) {
    // T and hat_U came as one allocation; the two hat_U pointers may have changed places (stop-rule runs of the small grids)
    void* slab_hat = (char*)E->dSlab + (size_t)E->N * E->N * E->esz;
    if (E->dHat && E->dHat != slab_hat) hipFree(E->dHat);
    if (E->dHat2 && E->dHat2 != slab_hat) hipFree(E->dHat2);
    hipFree(E->dSlab);
  } else {
    hipFree(E->dT1); hipFree(E->dHat);
    if (E->dHat2) hipFree(E->dHat2);
  }
  hipFree(E->dNoise); hipFree(E->dLambda); hipFree(E->dState); hipFree(E->dRows);
  for (auto e : E->timer.pool) hipEventDestroy(e);
  if (E->evA) hipEventDestroy(E->evA);
  if (E->evB) hipEventDestroy(E->evB);
  for (int i = 0; i < 4; ++i) if (E->evPoll[i]) hipEventDestroy(E->evPoll[i]);
  if (E->hState) hipHostFree(E->hState);
  if (E->hRows) hipHostFree(E->hRows);
  for (int i = 0; i < 2; ++i) {
    if (E->stageBuf[i]) hipHostFree(E->stageBuf[i]);
    if (E->stageEv[i]) hipEventDestroy(E->stageEv[i]);
  }
  if (E->stream) hipStreamDestroy(E->stream);
  delete E;
}

// The device view of the constants of one run.
static void fill_consts(Engine* E, const chs_consts* c) {
  const int N = c->N;
  E->hc = *c;
  DevConsts& d = E->dc;
  memset(&d, 0, sizeof d);
  d.N = N; d.adaptive_time = c->adaptive_time; d.full_sim = c->full_sim;
  d.RT = c->RT; d.BRT = c->BRT; d.B = c->B; d.A0 = c->A0; d.A1 = c->A1; d.Amr = c->Amr;
  d.kappa_tilde = c->kappa_tilde; d.L = c->L; d.delx = c->delx;
  d.delx2 = c->delx * c->delx;  // solution.py:29 `self.delx ** 2`
  d.delt0 = c->delt; d.delt_max = c->delt_max; d.M_tilde = c->M_tilde; d.threshold = c->threshold;
  d.time_limit_s = c->time_limit_s;
  d.invN2 = 1.0 / ((double)N * (double)N);
}
static DevState initial_state(const Engine* E) {
  DevState s0;
  memset(&s0, 0, sizeof s0);
  s0.delt = E->hc.delt;
  s0.delt_coef = E->hc.delt;
  s0.lam1 = E->hc.delt / E->dc.delx2;
  s0.lam2 = E->hc.kappa_tilde * s0.lam1 / E->dc.delx2;
  return s0;
}
static void read_env_hooks(Engine* E) {
  // with a stop rule armed a call usually ends early: shorter batches, fewer launches issued behind the stop
  // (a launch that finds the stop flag set costs ~2 us; the reference's default run stops at step 1674 of 1e6)
  E->batchSteps = (!E->dc.full_sim || E->dc.time_limit_s > 0.0) ? 256 : 1024;
  if (const char* bs = getenv("CHS_BATCH_STEPS")) {  // test hook: small batches exercise the polling path
    const long v = atol(bs);
    if (v >= 1 && v <= CHS_ROWS_RING / 8) E->batchSteps = (int)v;
  }
}

// Engine pool.  An ensemble creates one engine per member, all of one size: chs_destroy parks up to
// CHS_POOL_MAX engines (fields of at most CHS_POOL_FIELD_BYTES) instead of freeing ~15 device buffers, a stream
// and the pinned areas, and chs_create takes a parked engine of the same (device, N, dtype, transform engine,
// eigenvalue table) into use again: new constants, initial state, every per-run flag reset; the buffers are all
// written before they are read in a run.  CHS_ENGINE_POOL=0 switches it off.
#define CHS_POOL_MAX 4
#define CHS_POOL_FIELD_BYTES ((size_t)160 << 20)
#define CHS_POOL_TOTAL_BYTES ((size_t)3 << 30)  // device memory the parked engines may hold together
namespace {
std::mutex g_pool_mu;
std::vector<Engine*> g_pool;
bool pool_enabled() { const char* e = getenv("CHS_ENGINE_POOL"); return !(e && e[0] == '0'); }
// device bytes of an engine, to the accuracy the cap needs: its field-sized arrays (U, MU, T1, T2, hat_U [, the second
// hat_U of the small grids' stop-rule runs, the adaptive step's partial rows])
size_t engine_bytes(const Engine* E) {
  const size_t nb = (size_t)E->N * E->N * E->esz;
  return nb * (5 + (E->dHat2 ? 1 : 0) + (E->dNoise ? 1 : 0)) + (E->dPartColRows ? (size_t)E->nRowBlocks * E->N * E->esz : 0);
}
}
// Frees every parked engine (the process's only library-owned state besides the handles): for a caller that
// is done with a device, and registered by the Python binding to run at interpreter exit.
extern "C" int chs_pool_count(void) {
  std::lock_guard<std::mutex> lock(g_pool_mu);
  return (int)g_pool.size();
}
extern "C" int chs_pool_clear(void) {
  std::vector<Engine*> all;
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    all.swap(g_pool);
  }
  for (Engine* E : all) free_engine(E);
  return CHS_OK;
}
static int rearm(Engine* E, const chs_consts* c) {
  fill_consts(E, c);
  const DevState s0 = initial_state(E);
  CHS_HIP(hipMemcpy(E->dState, &s0, sizeof s0, hipMemcpyHostToDevice));
  E->prepared = false; E->have_U = false; E->hat_valid = false; E->resident = false;
  E->stateCached = false; E->keepResident = false;
  E->jitter = 0.0; E->jitterPcg = false;
  if (E->dNoise) { hipFree(E->dNoise); E->dNoise = nullptr; }
  E->nColMinCur = 0; E->lastStepMs = 0.0; E->timer.on = false;
  read_env_hooks(E);
  if (E->engine == CHS_ENGINE_FAST) return chs_fast_rearm(E);
  return CHS_OK;
}

extern "C" int chs_create(const chs_consts* c, const double* lambda, chs_handle* out) {
  if (!c || !lambda || !out) { chs_set_error("chs_create: null argument"); return CHS_EINVAL; }
  *out = nullptr;
  if (c->N < 8 || c->N > 16384) { chs_set_error("chs_create: N must be in [8, 16384]"); return CHS_EINVAL; }
  if (c->dtype != CHS_F64 && c->dtype != CHS_F32) { chs_set_error("chs_create: bad dtype"); return CHS_EINVAL; }
  int ndev = 0;
  CHS_HIP(hipGetDeviceCount(&ndev));
  if (c->device < 0 || c->device >= ndev) {
    chs_set_error("chs_create: no such HIP device (the engine needs a GPU; there is no CPU fallback)");
    return CHS_EINVAL;
  }
  CHS_HIP(hipSetDevice(c->device));
  if (pool_enabled()) {
    int want = c->engine;
    if (want == CHS_ENGINE_AUTO) want = chs_fast_supported(c->N, c->dtype) ? CHS_ENGINE_FAST : CHS_ENGINE_DIRECT;
    Engine* P = nullptr;
    {
      std::lock_guard<std::mutex> lock(g_pool_mu);
      for (size_t i = 0; i < g_pool.size(); ++i) {
        Engine* Q = g_pool[i];
        if (Q->hc.device == c->device && Q->N == c->N && Q->dtype == c->dtype && Q->engine == want &&
            memcmp(Q->hLambda.data(), lambda, sizeof(double) * (size_t)c->N) == 0) {
          P = Q;
          g_pool.erase(g_pool.begin() + (long)i);
          break;
        }
      }
    }
    if (P) {
      const int rcp = rearm(P, c);
      if (rcp) { free_engine(P); return rcp; }
      *out = (chs_handle)P;
      return CHS_OK;
    }
  }
  Engine* E = new (std::nothrow) Engine();
  if (!E) { chs_set_error("out of host memory"); return CHS_EINVAL; }
  E->hc = *c;
  E->N = c->N;
  E->dtype = c->dtype;
  E->esz = (c->dtype == CHS_F64) ? 8 : 4;
  const int N = c->N;
  int eng = c->engine;
  if (eng == CHS_ENGINE_AUTO) eng = chs_fast_supported(N, c->dtype) ? CHS_ENGINE_FAST : CHS_ENGINE_DIRECT;
  if (eng == CHS_ENGINE_FAST && !chs_fast_supported(N, c->dtype)) {
    delete E;
    chs_set_error("chs_create: the fast engine needs N = 2^k in [128, 8192]");
    return CHS_EINVAL;
  }
  if (eng != CHS_ENGINE_FAST && eng != CHS_ENGINE_DIRECT) { delete E; chs_set_error("bad engine"); return CHS_EINVAL; }
  E->engine = eng;
  fill_consts(E, c);
  DevConsts& d = E->dc;

  int rc = CHS_OK;
  auto fail = [&](int code) { free_engine(E); return code; };
#define TRY_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { chs_hip_fail(e__, #call, __FILE__, __LINE__); return fail(CHS_EHIP); } } while (0)
  TRY_HIP(hipStreamCreateWithFlags(&E->stream, hipStreamNonBlocking));
  TRY_HIP(hipEventCreate(&E->evA));
  TRY_HIP(hipEventCreate(&E->evB));
  for (int i = 0; i < 4; ++i) TRY_HIP(hipEventCreateWithFlags(&E->evPoll[i], hipEventDisableTiming));
  TRY_HIP(hipHostMalloc((void**)&E->hState, sizeof(DevState) * 5, hipHostMallocDefault));
  TRY_HIP(hipHostMalloc((void**)&E->hRows, sizeof(double) * 9 * CHS_ROWS_STAGED, hipHostMallocDefault));
  const size_t nb = (size_t)N * N * E->esz;
  TRY_HIP(hipMalloc(&E->dU, nb));
  TRY_HIP(hipMalloc(&E->dMU, nb));
  // CHS_SLAB=1 (experiment): T (the step loop's in-place operand) and hat_U in ONE allocation, back to back -- the two arrays
  // a step touches as one contiguous range, at N=4096 fp64 exactly the 256 MiB of the Infinity Cache, instead of two ranges an
  // arbitrary distance apart.  Measured equal (three engines each, both creation orders: 0.9997 / 1.0033; N=8192 fp32 0.9994):
  // whatever makes two engines of one library differ by up to 1.2 % is not the distance between the two arrays.
  {
    const char* e = getenv("CHS_SLAB");
    if (e && e[0] == '1') {
      TRY_HIP(hipMalloc(&E->dSlab, 2 * nb));
      E->dT1 = E->dSlab;
      E->dHat = (char*)E->dSlab + nb;
    } else {
      TRY_HIP(hipMalloc(&E->dT1, nb));
      TRY_HIP(hipMalloc(&E->dHat, nb));
    }
  }
  TRY_HIP(hipMalloc(&E->dT2, nb));
  TRY_HIP(hipMalloc(&E->dLambda, sizeof(double) * N));
  TRY_HIP(hipMemcpy(E->dLambda, lambda, sizeof(double) * N, hipMemcpyHostToDevice));
  TRY_HIP(hipMalloc(&E->dState, sizeof(DevState)));
  const DevState s0 = initial_state(E);
  TRY_HIP(hipMemcpy(E->dState, &s0, sizeof s0, hipMemcpyHostToDevice));
  E->hLambda.assign(lambda, lambda + N);
  if ((rc = ensure_rows(E))) return fail(rc);
  read_env_hooks(E);
  if ((rc = chs_pointwise_alloc(E))) return fail(rc);
  if (eng == CHS_ENGINE_DIRECT) rc = chs_direct_init(E); else rc = chs_fast_init(E);
  if (rc) return fail(rc);
#undef TRY_HIP
  *out = (chs_handle)E;
  return CHS_OK;
}

extern "C" int chs_destroy(chs_handle h) {
  Engine* E = (Engine*)h;
  if (!E) return CHS_OK;
  if (pool_enabled() && (size_t)E->N * E->N * E->esz <= CHS_POOL_FIELD_BYTES) {
    hipSetDevice(E->hc.device);
    if (hipStreamSynchronize(E->stream) == hipSuccess) {
      std::vector<Engine*> evicted;
      {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        g_pool.push_back(E);
        // the least recently parked ones make room: at most CHS_POOL_MAX engines, CHS_POOL_TOTAL_BYTES together
        size_t total = 0;
        for (const Engine* Q : g_pool) total += engine_bytes(Q);
        while (g_pool.size() > 1 && (g_pool.size() > CHS_POOL_MAX || total > CHS_POOL_TOTAL_BYTES)) {
          total -= engine_bytes(g_pool.front());
          evicted.push_back(g_pool.front());
          g_pool.erase(g_pool.begin());
        }
      }
      for (Engine* Q : evicted) free_engine(Q);
      return CHS_OK;
    }
  }
  free_engine(E);
  return CHS_OK;
}

extern "C" int chs_engine(chs_handle h) { return h ? ((Engine*)h)->engine : CHS_EINVAL; }

// The two pinned staging chunks of a handle (allocated at its first transfer and kept with it -- a parked engine
// keeps them too: pinning 16 MB costs milliseconds, an ensemble creates an engine per member).  Per handle, so
// that concurrent members of an ensemble move their fields at the same time.
static int stage_pair(Engine* E) {
  if (!E->stageBuf[0]) {
    for (int i = 0; i < 2; ++i) {
      CHS_HIP(hipHostMalloc(&E->stageBuf[i], CHS_STAGE_BYTES, hipHostMallocDefault));
      CHS_HIP(hipEventCreateWithFlags(&E->stageEv[i], hipEventDisableTiming));
    }
  }
  return CHS_OK;
}
// Field upload through the same two pinned chunks as the download below: the host fills (fp32: narrows) chunk k+1
// while the DMA of chunk k runs.
static int upload(Engine* E, void* dst, const double* src) {
  const size_t n = (size_t)E->N * E->N;
  const size_t esz = E->esz;
  int rc = stage_pair(E);
  if (rc) return rc;
  const size_t per = CHS_STAGE_BYTES / esz;
  const size_t nch = (n + per - 1) / per;
  for (size_t k = 0; k < nch; ++k) {
    const size_t o = k * per, cnt = (o + per <= n) ? per : n - o;
    if (k >= 2) CHS_HIP(hipEventSynchronize(E->stageEv[k & 1]));  // the copy that last used this chunk has finished
    if (E->dtype == CHS_F64) {
      memcpy(E->stageBuf[k & 1], src + o, cnt * 8);
    } else {
      float* f = (float*)E->stageBuf[k & 1];
      for (size_t i = 0; i < cnt; ++i) f[i] = (float)src[o + i];
    }
    CHS_HIP(hipMemcpyAsync((char*)dst + o * esz, E->stageBuf[k & 1], cnt * esz, hipMemcpyHostToDevice, E->stream));
    CHS_HIP(hipEventRecord(E->stageEv[k & 1], E->stream));
  }
  CHS_HIP(hipStreamSynchronize(E->stream));
  return CHS_OK;
}
// Field download through two pinned staging chunks: the DMA of chunk k+1 runs while the host copies (fp32: widens)
// chunk k into the caller's pageable array -- a direct hipMemcpy into pageable memory took 2-44 ms for 32 MB.
static int download(Engine* E, double* dst, const void* src) {
  const size_t n = (size_t)E->N * E->N;
  const size_t esz = E->esz;
  CHS_HIP(hipStreamSynchronize(E->stream));
  const int rcs = stage_pair(E);
  if (rcs) return rcs;
  const size_t per = CHS_STAGE_BYTES / esz;  // elements per chunk
  const size_t nch = (n + per - 1) / per;
  auto issue = [&](size_t k) -> hipError_t {
    const size_t o = k * per, cnt = (o + per <= n) ? per : n - o;
    hipError_t e = hipMemcpyAsync(E->stageBuf[k & 1], (const char*)src + o * esz, cnt * esz, hipMemcpyDeviceToHost, E->stream);
    if (e != hipSuccess) return e;
    return hipEventRecord(E->stageEv[k & 1], E->stream);
  };
  CHS_HIP(issue(0));
  for (size_t k = 0; k < nch; ++k) {
    if (k + 1 < nch) CHS_HIP(issue(k + 1));
    CHS_HIP(hipEventSynchronize(E->stageEv[k & 1]));
    const size_t o = k * per, cnt = (o + per <= n) ? per : n - o;
    if (E->dtype == CHS_F64) {
      memcpy(dst + o, E->stageBuf[k & 1], cnt * 8);
    } else {
      const float* f = (const float*)E->stageBuf[k & 1];
      for (size_t i = 0; i < cnt; ++i) dst[o + i] = (double)f[i];
    }
  }
  return CHS_OK;
}

extern "C" int chs_set_U(chs_handle h, const double* host_U) {
  Engine* E = (Engine*)h;
  if (!E || !host_U) { chs_set_error("chs_set_U: null argument"); return CHS_EINVAL; }
  E->resident = false;
  E->stateCached = false;
  CHS_HIP(hipSetDevice(E->hc.device));
  CHS_HIP(hipStreamSynchronize(E->stream));
  int rc = upload(E, E->dU, host_U);
  if (rc) return rc;
  E->have_U = true;
  E->hat_valid = false;
  return CHS_OK;
}

extern "C" int chs_get_U(chs_handle h, double* host_U) {
  Engine* E = (Engine*)h;
  if (!E || !host_U) { chs_set_error("chs_get_U: null argument"); return CHS_EINVAL; }
  if (!E->have_U) { chs_set_error("chs_get_U: no field uploaded"); return CHS_ESTATE; }
  CHS_HIP(hipSetDevice(E->hc.device));
  return download(E, host_U, E->dU);
}

extern "C" int chs_get_state(chs_handle h, chs_state* out) {
  Engine* E = (Engine*)h;
  if (!E || !out) { chs_set_error("chs_get_state: null argument"); return CHS_EINVAL; }
  DevState s;
  if (E->stateCached) {
    s = E->hState[0];  // what the last chs_step_n fetched behind its last kernel; nothing has run since
  } else {
    CHS_HIP(hipSetDevice(E->hc.device));
    CHS_HIP(hipStreamSynchronize(E->stream));
    CHS_HIP(hipMemcpy(&s, E->dState, sizeof s, hipMemcpyDeviceToHost));
  }
  out->delt = s.delt; out->time_delta_sum = s.time_delta_sum; out->time_passed = s.time_passed;
  out->tau0 = s.tau0; out->t0 = s.t0; out->computed_steps = s.computed_steps;
  out->skip_check = s.skip_check; out->stop_reason = s.stop_reason;
  return CHS_OK;
}

extern "C" int chs_set_state(chs_handle h, const chs_state* in) {
  Engine* E = (Engine*)h;
  if (!E || !in) { chs_set_error("chs_set_state: null argument"); return CHS_EINVAL; }
  E->resident = false;
  E->stateCached = false;
  CHS_HIP(hipSetDevice(E->hc.device));
  CHS_HIP(hipStreamSynchronize(E->stream));
  DevState s;
  CHS_HIP(hipMemcpy(&s, E->dState, sizeof s, hipMemcpyDeviceToHost));
  s.delt = in->delt; s.time_delta_sum = in->time_delta_sum; s.time_passed = in->time_passed;
  s.tau0 = in->tau0; s.t0 = in->t0; s.computed_steps = in->computed_steps;
  s.skip_check = in->skip_check; s.stop_reason = in->stop_reason;
  // (the coefficients follow params.delt, not this delt: every call reloads them, k_call_begin)
  CHS_HIP(hipMemcpy(E->dState, &s, sizeof s, hipMemcpyHostToDevice));
  E->csHost = s.computed_steps;
  return CHS_OK;
}

// solver.py:84-135
extern "C" int chs_prepare(chs_handle h, double row0[9]) {
  Engine* E = (Engine*)h;
  if (!E || !row0) { chs_set_error("chs_prepare: null argument"); return CHS_EINVAL; }
  if (!E->have_U) { chs_set_error("chs_prepare: chs_set_U first"); return CHS_ESTATE; }
  E->resident = false;
  E->stateCached = false;
  CHS_HIP(hipSetDevice(E->hc.device));
  int rc;
  if ((rc = chs_launch_sum(E, 1))) return rc;
  if ((rc = chs_launch_diag(E, 1))) return rc;
  if ((rc = chs_launch_fin(E, 1))) return rc;
  CHS_HIP(hipStreamSynchronize(E->stream));
  CHS_HIP(hipMemcpy(row0, E->dRows, sizeof(double) * 9, hipMemcpyDeviceToHost));
  E->prepared = true;
  E->csHost = 1;   // (solver.py:128-135: computed_steps = 1 behind the step-0 record)
  for (int i = 0; i < 9; ++i)
    if (row0[i] != row0[i]) { chs_set_error("chs_prepare: NaN in the step-0 record (timedata.py:10)"); return CHS_ENAN; }
  return CHS_OK;
}

// hat_U <- dctn(U), solver.py:159
static int enter(Engine* E) {
  if (E->engine == CHS_ENGINE_DIRECT) return chs_direct_dct2d(E, E->dU, E->dHat, E->dT1, false);
  return chs_fast_enter(E);
}

// one iteration of solver.py:165-249
static int one_step(Engine* E, bool first, bool last) {
  int rc;
  const bool jitter = (E->dNoise || E->jitterPcg) && E->jitter > 0.0 && E->jitter < 0.1;
  if (E->engine == CHS_ENGINE_FAST && !jitter) {
    // fused pipeline: k_col, k_row_inv (+ record partials + next step's row pass), k_step_tail
    return chs_fast_step(E, first, last);
  }
  if (E->engine == CHS_ENGINE_DIRECT) {
    if ((rc = chs_launch_mu(E))) return rc;        // 166-175
    if ((rc = chs_launch_pre(E))) return rc;       // 177-199, 225
    chs_slot_begin(E, SLOT_FWD);
    rc = chs_direct_dct2d(E, E->dMU, E->dT2, E->dT1, false);  // dctn(EnergieEut), 201
    chs_slot_end(E, SLOT_FWD);
    if (rc) return rc;
    if ((rc = chs_launch_spectral(E, E->dT2))) return rc;      // 201-206
    chs_slot_begin(E, SLOT_INV);
    rc = chs_direct_dct2d(E, E->dHat, E->dU, E->dT1, true);   // 208
    chs_slot_end(E, SLOT_INV);
    if (rc) return rc;
  } else {
    if ((rc = chs_fast_step_unfused(E))) return rc;
  }
  if (jitter) {                                    // 210-211
    // the perturbation and the mean of the perturbed field (meanU, for PS) in one sweep
    if ((rc = E->jitterPcg ? chs_launch_jitter_pcg(E) : chs_launch_jitter(E))) return rc;
  }
  if ((rc = chs_launch_diag(E, 0))) return rc;     // 213-228
  if ((rc = chs_launch_fin(E, 0))) return rc;      // 230-249
  return CHS_OK;
}

// Steps are issued in batches.  Behind every batch the device state is copied into a pinned slot;
// before batch b+1 goes out, the slot behind batch b-1 is looked at (the device is then busy with
// batch b, so it never idles): once `halt` is up -- energy stop, time limit, NaN -- nothing more is
// issued.  The reference's defaults are ntmax = 1e6 with full_sim = False (parameters.py:42,50): a run
// that stops after a thousand steps must not queue three million empty launches behind the stop.
// The rows of finished batches are copied out of the device ring as they complete.
#ifndef CHS_BATCH_STEPS
#define CHS_BATCH_STEPS 1024
#endif
static_assert(CHS_ROWS_RING % CHS_BATCH_STEPS == 0 && CHS_ROWS_RING >= 8 * CHS_BATCH_STEPS, "ring and batch size");

static int copy_rows_out(Engine* E, double* rows, int64_t from, int64_t to) {
  // ring -> caller's array, rows [from, to); at most two pieces
  while (from < to) {
    const int64_t slot = from % E->rowsCap;
    int64_t n = to - from;
    if (slot + n > E->rowsCap) n = E->rowsCap - slot;
    CHS_HIP(hipMemcpy(rows + from * 9, E->dRows + slot * 9, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToHost));
    from += n;
  }
  return CHS_OK;
}

static int run_steps(Engine* E, int64_t nsteps, int flags, double* rows, int64_t* steps_done, bool profile) {
  if (!E->prepared) { chs_set_error("chs_step_n: not prepared (solver.py:139)"); return CHS_ESTATE; }
  if (nsteps < 0) nsteps = 0;
  CHS_HIP(hipSetDevice(E->hc.device));
  E->stateCached = false;
#if CHS_TEST_HOOKS
  {
    const char* gw = getenv("CHS_TEST_GATE_WITHHOLD");  // test hook: provoke the gate's timeout path
    E->testGateWithhold = gw && gw[0] == '1';
  }
#endif
  int rc;
  if ((rc = ensure_rows(E))) return rc;
  if (profile) {
    E->timer.on = true;
    for (int i = 0; i < CHS_NKERNELS; ++i) { E->timer.ms[i] = 0; E->timer.calls[i] = 0; }
  }
  CHS_HIP(hipEventRecord(E->evA, E->stream));
  E->timer.on = false;  // the entry work is not a per-step kernel
  // a new call re-arms the loop and reloads the coefficients of params.delt (k_call_begin)
  if ((rc = chs_launch_call_begin(E))) return rc;
  const bool fused = (E->engine == CHS_ENGINE_FAST) && !((E->dNoise || E->jitterPcg) && E->jitter > 0.0 && E->jitter < 0.1);
  const bool derive = !((flags & CHS_STEP_CARRY_HAT) && E->hat_valid);
  // Fixed time step on the fused pipeline: the last step of a call leaves hat_U, the row transform of
  // EnergieEut(U) and its sum of squares on the device, and a call that finds them continues the loop where it
  // stopped -- hat_U is the array the reference would recompute as dctn(idctn(hat_U)) (solver.py:159), equal up
  // to rounding; CHS_STEP_REDERIVE_HAT asks for the literal recomputation.
  // A caller that asks for the literal re-derivation (CHS_STEP_REDERIVE_HAT) gets hat_U = dctn(U) recomputed at every
  // call; what such a call still takes over from its predecessor is the OTHER thing the last fused step leaves: T1 = the
  // row transform of EnergieEut(U) and its sum of squares -- a function of the unchanged field U alone, which k_row_fwd2
  // would only compute again bit for bit -- when the caller allows it (CHS_STEP_KEEP_T1; off by default: measured, it
  // buys nothing at N=4096 -- the entry shrinks from 215 to 152 us, the call's last step, now the fused kernel, grows by
  // as much: profiles/r04_ab_entry.txt).
  const bool keep_t1 = (flags & CHS_STEP_KEEP_T1) != 0;
  const bool rederive = (flags & CHS_STEP_REDERIVE_HAT) != 0;
  E->keepResident = fused && !E->dc.adaptive_time && !profile && !(flags & CHS_STEP_LAST_CALL) && (!rederive || keep_t1);
  const bool cont = fused && E->resident && E->hat_valid && !rederive && !profile;
  const bool cont_t1 = fused && E->resident && rederive && keep_t1 && !profile;
  if (nsteps > 0) E->resident = false;
  if (cont && nsteps > 0) {
    // nothing to do: T1, partMu and hat_U are in place
  } else if (cont_t1 && nsteps > 0) {
    // hat_U = dctn(U), literally; T1 and partMu are in place
    if ((rc = chs_fast_enter_hat(E))) return rc;
    E->hat_valid = true;
  } else if (derive && fused && nsteps > 0) {
    // hat_U = dctn(U) and the first step's row transform of EnergieEut(U) from one sweep of U
    if ((rc = chs_fast_enter_fused(E))) return rc;
    E->hat_valid = true;
  } else {
    if (derive) {
      if ((rc = enter(E))) return rc;
    }
    E->hat_valid = true;
    E->timer.on = profile;
    if (fused && nsteps > 0) {
      if ((rc = chs_fast_prologue(E))) { E->timer.on = false; return rc; }
    }
  }
  E->timer.on = profile;
  E->dHatCall = E->dHat;  // (chs_fast_step may alternate two hat_U buffers from here on)
  E->hatFlip = false;
  int64_t issued = 0, copied = 0;
  int batch = 0;
  bool stopped = false;
  while (issued < nsteps && !stopped) {
    int64_t nb = nsteps - issued;
    if (nb > E->batchSteps) nb = E->batchSteps;
    for (int64_t s = issued; s < issued + nb; ++s) {
      if ((rc = one_step(E, s == 0, s == nsteps - 1))) { E->timer.on = false; return rc; }
      if (profile && (s % 32) == 31) timer_harvest(E);
    }
    issued += nb;
    if (issued < nsteps) {
      const int slot = 1 + (batch & 3);
      CHS_HIP(hipMemcpyAsync(&E->hState[slot], E->dState, sizeof(DevState), hipMemcpyDeviceToHost, E->stream));
      CHS_HIP(hipEventRecord(E->evPoll[batch & 3], E->stream));
      if (batch >= 1) {
        const int prev = (batch - 1) & 3;
        CHS_HIP(hipEventSynchronize(E->evPoll[prev]));
        const DevState& ps = E->hState[1 + prev];
        if (rows && ps.rows_written > copied) {
          if ((rc = copy_rows_out(E, rows, copied, ps.rows_written))) return rc;
          copied = ps.rows_written;
        }
        if (ps.halt) stopped = true;
      }
      ++batch;
    }
  }
  if (profile) { timer_harvest(E); E->timer.on = false; }
  CHS_HIP(hipEventRecord(E->evB, E->stream));
  CHS_HIP(hipMemcpyAsync(&E->hState[0], E->dState, sizeof(DevState), hipMemcpyDeviceToHost, E->stream));
  // a short call fetches its rows in the same breath (pinned staging): one wait for state and record
  const bool staged = rows && copied == 0 && nsteps > 0 && nsteps <= CHS_ROWS_STAGED;
  if (staged) CHS_HIP(hipMemcpyAsync(E->hRows, E->dRows, sizeof(double) * 9 * (size_t)nsteps, hipMemcpyDeviceToHost, E->stream));
  CHS_HIP(hipStreamSynchronize(E->stream));
  float ms = 0.f;
  CHS_HIP(hipEventElapsedTime(&ms, E->evA, E->evB));
  E->lastStepMs = ms;
  const DevState s = E->hState[0];
  if (s.gate_timeout) {
    // A workgroup gave up waiting for the riding bookkeeping (never seen outside the test hook): the call's
    // remaining kernels were no-ops, the loop's device state is not that of a completed step.  The handle stays
    // usable, but only through a new field: the fused row kernel has not been storing U and the counters have
    // advanced by the steps that did finish, so nothing on the device is a consistent state to step on from --
    // the next chs_step_n returns CHS_ESTATE until chs_set_U / chs_init_U_pcg64 and chs_prepare have run.
    E->hat_valid = false; E->resident = false; E->stateCached = false;
    E->tailDeferred = false; E->tailGated = false;
    E->prepared = false; E->have_U = false; E->csHost = -1;
    if (steps_done) *steps_done = 0;
    chs_set_error("internal: a workgroup gave up waiting for the step's bookkeeping (gated tail)");
    return CHS_EHIP;
  }
  if (E->hatFlip) {
    // two hat_U buffers alternated per ISSUED step; the one that holds the state behind the steps that were
    // COMPLETED is the call's first one after an even number of them, the other one after an odd number
    void* other = (E->dHat == E->dHatCall) ? E->dHat2 : E->dHat;
    const bool even = ((s.rows_written < nsteps ? s.rows_written : nsteps) & 1) == 0;
    void* valid = even ? E->dHatCall : other;
    E->dHat2 = (valid == E->dHatCall) ? other : E->dHatCall;
    E->dHat = valid;
  }
  if (s.halt && s.stop_reason != CHS_STOP_NONE && !s.nan_flag && fused && !E->storeU && s.rows_written < nsteps) {
    // the energy rule or the time limit ended the call before its last step and the row kernel has
    // been keeping U in registers: hat_U is that of the last completed step, rebuild the field from
    // it (solver.py:197-199 breaks before U is updated, 242-251 returns the U of the stopping step)
    DevState r = s;
    r.halt = 0;
    CHS_HIP(hipMemcpy(E->dState, &r, sizeof r, hipMemcpyHostToDevice));
    if ((rc = chs_fast_recover_u(E))) return rc;
    CHS_HIP(hipStreamSynchronize(E->stream));
    CHS_HIP(hipMemcpy(E->dState, &s, sizeof s, hipMemcpyHostToDevice));
  }
  if (s.halt) E->hat_valid = false;  // (a deferred tail lets k_col run once past a NaN stop)
  if (nsteps > 0) E->resident = E->keepResident && !s.halt && s.rows_written >= nsteps;
  E->stateCached = true;  // (the recovery above leaves the device state equal to s)
  E->csHost = s.computed_steps;
  int64_t done = s.rows_written;
  if (done > nsteps) done = nsteps;
  if (steps_done) *steps_done = done;
  if (staged) {
    memcpy(rows, E->hRows, sizeof(double) * 9 * (size_t)done);
  } else if (rows && done > copied) {
    if ((rc = copy_rows_out(E, rows, copied, done))) return rc;
  }
  if (rows) {
    // solver.py:230 `domtime = self.time_passed ** (1 / 3)` with the host libm
    for (int64_t i = 0; i < done; ++i) rows[i * 9 + 4] = pow(rows[i * 9 + 4], 1.0 / 3.0);
  }
  if (s.nan_flag) {
    chs_set_error("NaN in a recorded scalar (timedata.py:10): U left (0,1)");
    return CHS_ENAN;
  }
  return CHS_OK;
}

extern "C" int chs_step_n(chs_handle h, int64_t nsteps, int32_t flags, double* rows, int64_t* steps_done) {
  Engine* E = (Engine*)h;
  if (!E) { chs_set_error("chs_step_n: null handle"); return CHS_EINVAL; }
  if (nsteps > 0 && !rows) { chs_set_error("chs_step_n: rows is null"); return CHS_EINVAL; }
  return run_steps(E, nsteps, flags, rows, steps_done, false);
}

extern "C" int chs_profile_steps(chs_handle h, int64_t nsteps, double ms[CHS_NKERNELS], int64_t calls[CHS_NKERNELS]) {
  Engine* E = (Engine*)h;
  if (!E || !ms || !calls) { chs_set_error("chs_profile_steps: null argument"); return CHS_EINVAL; }
  std::vector<double> rows((size_t)(nsteps > 0 ? nsteps : 1) * 9);
  int64_t done = 0;
  int rc = run_steps(E, nsteps, 0, rows.data(), &done, true);
  for (int i = 0; i < CHS_NKERNELS; ++i) { ms[i] = E->timer.ms[i]; calls[i] = E->timer.calls[i]; }
  return rc;
}

extern "C" double chs_last_step_ms(chs_handle h) { return h ? ((Engine*)h)->lastStepMs : -1.0; }

extern "C" const char* chs_kernel_name(chs_handle h, int slot) {
  Engine* E = (Engine*)h;
  if (!E || slot < 0 || slot >= CHS_NKERNELS) return nullptr;
  static const char* direct[CHS_NKERNELS] = {"k_mu", "k_pre", "k_gemm x2 (dctn)", "k_spectral", "k_gemm x2 (idctn)",
                                             "k_diag", "k_fin", "misc"};
  static const char* fast[CHS_NKERNELS] = {"k_row_fwd (prologue)", "k_pre", nullptr, "k_col", "k_row_inv (fused)",
                                           "k_diag", "k_step_tail", "misc"};
  return E->engine == CHS_ENGINE_DIRECT ? direct[slot] : fast[slot];
}

extern "C" int chs_init_U_pcg64(chs_handle h, double base, double scale, const uint64_t state[2], const uint64_t inc[2]) {
  Engine* E = (Engine*)h;
  if (!E || !state || !inc) { chs_set_error("chs_init_U_pcg64: null argument"); return CHS_EINVAL; }
  E->resident = false;
  E->stateCached = false;
  CHS_HIP(hipSetDevice(E->hc.device));
  const unsigned long long st[2] = {state[0], state[1]}, ic[2] = {inc[0], inc[1]};
  const int rc = chs_launch_init_pcg(E, base, scale, st, ic);
  if (rc) return rc;
  CHS_HIP(hipStreamSynchronize(E->stream));
  E->have_U = true;
  E->hat_valid = false;
  return CHS_OK;
}

extern "C" int chs_set_jitter_pcg64(chs_handle h, double jitter, const uint64_t state[2], const uint64_t inc[2]) {
  Engine* E = (Engine*)h;
  if (!E || !state || !inc) { chs_set_error("chs_set_jitter_pcg64: null argument"); return CHS_EINVAL; }
  CHS_HIP(hipSetDevice(E->hc.device));
  if (E->dNoise) { hipFree(E->dNoise); E->dNoise = nullptr; }
  E->jitter = jitter;
  E->jitterPcg = true;
  E->pcgState[0] = state[0]; E->pcgState[1] = state[1];
  E->pcgInc[0] = inc[0]; E->pcgInc[1] = inc[1];
  return CHS_OK;
}

extern "C" int chs_set_jitter_noise(chs_handle h, double jitter, const double* host_noise) {
  Engine* E = (Engine*)h;
  if (!E) { chs_set_error("chs_set_jitter_noise: null handle"); return CHS_EINVAL; }
  CHS_HIP(hipSetDevice(E->hc.device));
  E->jitter = jitter;
  E->jitterPcg = false;
  if (!host_noise) {  // switch jitter off
    if (E->dNoise) { hipFree(E->dNoise); E->dNoise = nullptr; }
    return CHS_OK;
  }
  if (!E->dNoise) CHS_HIP(hipMalloc(&E->dNoise, (size_t)E->N * E->N * E->esz));
  CHS_HIP(hipStreamSynchronize(E->stream));
  return upload(E, E->dNoise, host_noise);
}

extern "C" int chs_dctn(chs_handle h, const double* host_in, double* host_out, int inverse) {
  Engine* E = (Engine*)h;
  if (!E || !host_in || !host_out) { chs_set_error("chs_dctn: null argument"); return CHS_EINVAL; }
  E->resident = false;
  E->stateCached = false;
  CHS_HIP(hipSetDevice(E->hc.device));
  CHS_HIP(hipStreamSynchronize(E->stream));
  // dMU/dT2 are scratch between steps
  int rc = upload(E, E->dMU, host_in);
  if (rc) return rc;
  // make sure a stale halt flag does not turn the transform into a no-op
  DevState s;
  CHS_HIP(hipMemcpy(&s, E->dState, sizeof s, hipMemcpyDeviceToHost));
  const int halt = s.halt;
  s.halt = 0;
  CHS_HIP(hipMemcpy(E->dState, &s, sizeof s, hipMemcpyHostToDevice));
  if (E->engine == CHS_ENGINE_DIRECT) rc = chs_direct_dct2d(E, E->dMU, E->dT2, E->dT1, inverse != 0);
  else rc = chs_fast_dct2d(E, E->dMU, E->dT2, inverse != 0);
  if (rc) return rc;
  rc = download(E, host_out, E->dT2);
  s.halt = halt;
  CHS_HIP(hipMemcpy(E->dState, &s, sizeof s, hipMemcpyHostToDevice));
  return rc;
}

extern "C" int chs_get_mu(chs_handle h, double* host_mu) {
  Engine* E = (Engine*)h;
  if (!E || !host_mu) { chs_set_error("chs_get_mu: null argument"); return CHS_EINVAL; }
  if (!E->have_U) { chs_set_error("chs_get_mu: no field uploaded"); return CHS_ESTATE; }
  E->resident = false;
  E->stateCached = false;
  CHS_HIP(hipSetDevice(E->hc.device));
  DevState s;
  CHS_HIP(hipStreamSynchronize(E->stream));
  CHS_HIP(hipMemcpy(&s, E->dState, sizeof s, hipMemcpyDeviceToHost));
  const int halt = s.halt;
  s.halt = 0;
  CHS_HIP(hipMemcpy(E->dState, &s, sizeof s, hipMemcpyHostToDevice));
  int rc = chs_launch_mu(E);
  if (rc) return rc;
  rc = download(E, host_mu, E->dMU);
  s.halt = halt;
  CHS_HIP(hipMemcpy(E->dState, &s, sizeof s, hipMemcpyHostToDevice));
  return rc;
}
