/* chs_hip.h -- C ABI of the MI355X (gfx950) Cahn-Hilliard timestep engine.
 *
 * Drop-in boundary for ONE hot path of uncertaintyhub/chsimpy: the per-timestep
 * solver loop.  The reference has no FFI of its own; the seam is the Python
 * method pair
 *     Solver.prepare()                 chsimpy/solver.py:84-135
 *     Solver.solve_or_resume(nsteps)   chsimpy/solver.py:137-252
 * (called from chsimpy/simulator.py:41,43,69 and examples/benchmark.py:72-74).
 * A ctypes-backed `Solver` look-alike (chsimpy_amd/solver.py) binds exactly the
 * entry points below; INTEGRATION.md shows the stub a chsimpy maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every host buffer, the
 *     library owns the device buffers for the lifetime of the handle;
 *   - one opaque handle per simulation, one HIP stream and one pair of pinned staging
 *     chunks per handle -> 8 handles can live on 8 devices of a node, and several on one
 *     device move their fields concurrently (ensemble runs, chsimpy/experiment.py:84-126).
 *     Library-owned state outside the handles: the thread-local last-error string and the
 *     pool of parked engines (a mutex-protected free list, see chs_create / chs_pool_clear);
 *   - every call is synchronous on return;
 *   - return value: CHS_OK (0) or a negative CHS_E* code; chs_last_error()
 *     gives the text.
 */
#ifndef CHS_HIP_H
#define CHS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHS_OK 0
#define CHS_EINVAL (-1)   /* bad argument / unsupported configuration            */
#define CHS_EHIP (-2)     /* a HIP runtime call failed                           */
#define CHS_ENAN (-3)     /* a recorded scalar became NaN (chsimpy/timedata.py:10) */
#define CHS_ESTATE (-4)   /* call order violated (e.g. step before prepare:
                             the assert at chsimpy/solver.py:139)                */

/* arithmetic type of the device path */
#define CHS_F64 0
#define CHS_F32 1

/* transform engine */
#define CHS_ENGINE_AUTO 0   /* fast for N = power of two >= 128, else direct     */
#define CHS_ENGINE_DIRECT 1 /* dense cosine-matrix products, any N >= 8          */
#define CHS_ENGINE_FAST 2   /* LDS-staged FFT-based row/column DCT passes        */

/* stop reasons (chsimpy/solver.py:133,198,246) */
#define CHS_STOP_NONE 0
#define CHS_STOP_ENERGY 1
#define CHS_STOP_TIME_LIMIT 2

/* Scalars the loop reads.  They are what chsimpy/solution.py:25-55 derives from
 * chsimpy/parameters.py:24-61; the host wrapper computes them exactly as the
 * reference does and hands them over. */
typedef struct chs_consts {
  int32_t N;              /* grid is N x N                       parameters.py:25 */
  int32_t dtype;          /* CHS_F64 | CHS_F32                                    */
  int32_t device;         /* HIP device ordinal                                   */
  int32_t engine;         /* CHS_ENGINE_*                                         */
  int32_t adaptive_time;  /* parameters.py:57 ; solver.py:177-193                 */
  int32_t full_sim;       /* parameters.py:50 ; solver.py:245-249                 */
  double RT;              /* R*temp                              solution.py:31   */
  double BRT;             /* B*R*temp                            solution.py:32   */
  double B;               /*                                     parameters.py:31 */
  double A0, A1;          /* func_A0(temp), func_A1(temp)        solution.py:34-35 */
  double Amr;             /* 1/Am                                solution.py:33   */
  double kappa_tilde;     /*                                     solution.py:39-48 */
  double L;               /*                                     parameters.py:26 */
  double delx;            /* L/(N-1)                             solution.py:28   */
  double delt;            /* initial time step                   parameters.py:36 */
  double delt_max;        /*                                     parameters.py:37 */
  double M_tilde;         /*                                     parameters.py:38 */
  double threshold;       /*                                     parameters.py:41 */
  double time_limit_s;    /* time_max*60, <= 0: none             solver.py:151-153 */
} chs_consts;

/* Solver-side state that survives between solve_or_resume calls
 * (chsimpy/solver.py:50-54) plus the Solution counters (solution.py:57-61). */
typedef struct chs_state {
  double delt;            /* solver.py:54, updated by 185-188   */
  double time_delta_sum;  /* solver.py:51,195                   */
  double time_passed;     /* solver.py:52,196                   */
  double tau0;            /* solution.py:58 ; solver.py:243     */
  double t0;              /* solution.py:59 ; solver.py:244     */
  int64_t computed_steps; /* solution.py:60 ; solver.py:134,240 */
  int32_t skip_check;     /* solver.py:50,249                   */
  int32_t stop_reason;    /* CHS_STOP_*                         */
} chs_state;

typedef struct chs_handle_s* chs_handle;

/* Create an engine for one simulation.  `lambda` is the host-computed 1-D table
 * lam_i = 2cos(pi i/(N-1)) - 2, i = 0..N-1 (chsimpy/utils.py:34-36); the N x N
 * grids CHeig/Seig of utils.py:39-49 are never materialised, they are formed on
 * the fly from this table and the current delt.
 * Environment (read here, test hooks): CHS_ADAPT_SWEEP=1 keeps the separate sweep
 * of U for the adaptive-step column sums instead of the fused row kernel's;
 * CHS_BATCH_STEPS=n issues the steps of a call n at a time (default 1024) -- between
 * batches the host looks at the device's stop flag, see chs_step_n.
 * chs_destroy parks up to four engines (fields up to 160 MB) instead of freeing them and chs_create takes a
 * parked engine of the same device, N, dtype, transform engine and lambda table into use again with the new
 * constants -- an ensemble creates one engine per member; a run on such an engine is bit for bit the run on
 * a new one.  The parked engines hold at most 3 GiB of device memory together (the least recently parked
 * ones are freed first); chs_pool_clear() frees them all.  CHS_ENGINE_POOL=0 (read at both calls) switches
 * the pool off.  A handle must not be used after chs_destroy either way.
 * CHS_TEST_GATE_WITHHOLD=1 (test hook, read by chs_step_n; compiled in only when the library is built with
 * -DCHS_TEST_HOOKS=1, which __graft_entry__.build() does -- a build without it ignores the variable): the in-launch
 * bookkeeping of stop-rule / adaptive runs never publishes its decision, so that the waiting workgroups run into
 * their bounded timeout -> chs_step_n returns CHS_EHIP. */
int chs_create(const chs_consts* consts, const double* lambda, chs_handle* out);
int chs_destroy(chs_handle h);
/* Free every parked engine of this process (all devices); chs_pool_count: how many are parked right now. */
int chs_pool_clear(void);
int chs_pool_count(void);

/* Upload the N x N row-major float64 field (the reference's U_init / solution.U).
 * Replaces `U = self.U_init.copy()` (solver.py:85) and `U = self.solution.U`
 * (solver.py:158). */
int chs_set_U(chs_handle h, const double* host_U);
/* The same without a host array, for the reference's default start field (solver.py:78-82):
 * U[r][c] = base + scale * (rand - 0.5), rand = draw r*N+c of numpy's PCG64 `Generator.random`
 * started from the 128-bit `state` / `inc` ({high word, low word}) -- bit-identical to
 * `XXX + XXX*0.01*(rng.random((N,N)) - 0.5)` with base = XXX, scale = XXX*0.01. */
int chs_init_U_pcg64(chs_handle h, double base, double scale, const uint64_t state[2], const uint64_t inc[2]);
/* Download the current field (solver.py:251 `self.solution.U = U`). */
int chs_get_U(chs_handle h, double* host_U);

/* Step-0 diagnostics of solver.py:100-127 on the current field.  Fills
 * row0 = [it=0, E, E2, SA=0, domtime=0, Ra, L2=0, PS, delt] (column order of
 * chsimpy/timedata.py:9) and resets computed_steps=1, tau0=t0=0,
 * stop_reason=None (solver.py:128-135).  delt / time_delta_sum / skip_check
 * are NOT reset, exactly as in the reference. */
int chs_prepare(chs_handle h, double row0[9]);

/* Run up to `nsteps` iterations of solver.py:165-249 on the device without any
 * per-step host synchronisation (the launches go out in batches; once the device
 * has raised its stop flag no further batch is issued, so a run that stops early
 * does not pay for the rest of ntmax).  On entry hat_U = dctn(U) is re-derived
 * (solver.py:159) and the coefficients CHeig/Seig are those of consts.delt again
 * (solver.py:154-155) until the adaptive step regenerates them (189-193).  The caller passes the iteration count of
 * range(itbegin, nsteps) (solver.py:160-165).  `rows` receives one 9-column
 * timedata row per completed step ([nsteps][9], column order timedata.py:9);
 * `*steps_done` how many were completed (fewer than nsteps after an energy or
 * time-limit stop, which are decided on the device, solver.py:197-199,242-249).
 * Returns CHS_ENAN when a recorded scalar is NaN (timedata.py:10); the rows up
 * to and including the NaN row are still returned, the field is unspecified then
 * (the reference raises before it assigns solution.U, solver.py:251).
 * The device keeps the field of intermediate steps in registers where nothing can
 * observe it; chs_get_U after the call returns the field of the last completed step
 * in every mode -- also after an energy or time-limit stop, where it is rebuilt from
 * hat_U -- exactly as `self.solution.U = U` after the reference's loop
 * (solver.py:197-199, 242-251).
 * CHS_EHIP with "gave up waiting" (the bounded wait of the in-launch bookkeeping ran out; only ever seen with
 * the test hook): *steps_done = 0 and the device holds no consistent state of a completed step -- the handle
 * is un-prepared and field-less afterwards: chs_step_n returns CHS_ESTATE until chs_set_U / chs_init_U_pcg64
 * and chs_prepare have been called again. */
int chs_step_n(chs_handle h, int64_t nsteps, int32_t flags, double* rows, int64_t* steps_done);
/* flags for chs_step_n */
#define CHS_STEP_CARRY_HAT 1 /* do not re-derive hat_U on entry: continue the loop of the
                                previous call (used to feed per-step host jitter noise while
                                keeping the reference's carried hat_U, solver.py:206-211) */

#define CHS_STEP_REDERIVE_HAT 2 /* recompute hat_U = dctn(U) on entry (the literal solver.py:159) even when the
                                  previous call left the loop's state on the device.  Without it a call with a
                                  fixed time step that follows a completed call (no new field, state or noise
                                  in between) continues that call's loop: hat_U is carried instead of being
                                  recomputed from the field it was inverted to -- the same array up to
                                  rounding -- and the sequence of calls gives bit for bit what one call gives.
                                  (implies that the call's last step prepares no continuation, like
                                  CHS_STEP_LAST_CALL, unless CHS_STEP_KEEP_T1 is given too) */
#define CHS_STEP_KEEP_T1 8     /* with CHS_STEP_REDERIVE_HAT: hat_U = dctn(U) is recomputed on entry, but the row
                                  transform of EnergieEut(U) for the first step -- a function of the unchanged field
                                  alone, which the last step of a completed predecessor left on the device as every
                                  step inside a call does -- is taken over instead of being computed again (bit for
                                  bit the same run); such a call's own last step prepares it for a successor */
#define CHS_STEP_LAST_CALL 4   /* the run ends with this call: its last step does not prepare a continuation (the
                                  forward row pass of a step that will not come); a later call is still correct,
                                  it enters through hat_U = dctn(U) */

int chs_get_state(chs_handle h, chs_state* out);
int chs_set_state(chs_handle h, const chs_state* in);

/* U += jitter * (2*noise - 1) with host-supplied noise (solver.py:210-211).  The
 * reference draws `noise` from its seeded host generator; keeping the draw on the
 * host keeps its stream bit-identical.  hat_U is left untouched, as in the
 * reference. */
int chs_set_jitter_noise(chs_handle h, double jitter, const double* host_noise);
/* The same with the noise drawn on the device: the stream of numpy's PCG64 (`Generator.random`,
 * the reference's default generator, solver.py:78-82,211) continued from the 128-bit `state` and
 * `inc` of the host generator ({high word, low word}); every step consumes N*N draws in C order.
 * The caller advances its own generator by N*N per completed step afterwards
 * (`bit_generator.advance`).  jitter outside (0, 0.1) or a later chs_set_jitter_noise switches it off. */
int chs_set_jitter_pcg64(chs_handle h, double jitter, const uint64_t state[2], const uint64_t inc[2]);

/* Test / diagnostic hooks (not on the reference seam). */
/* 2-D orthonormal DCT-II (inverse=0) or DCT-III (inverse=1) of a host array
 * through the handle's transform engine: scipy.fftpack.dctn/idctn(norm='ortho')
 * as called at solver.py:159,201,208. */
int chs_dctn(chs_handle h, const double* host_in, double* host_out, int inverse);
/* EnergieEut of solver.py:166-175 for the current field. */
int chs_get_mu(chs_handle h, double* host_mu);
/* Device math primitives evaluated elementwise on `device` (accuracy tests):
 * which = 0: log(a)   1: log(a/b)   2: EnergieEut(a) with (RT,BRT,A0,A1) = b[0..3]
 *         3: bulk energy density(a) with (RT,B,A0,A1) = b[0..3]  (solver.py:218-221)
 *         4: log(a) for a > 0 (division-based variant)
 *         5: log(a) for a > 0, table-driven (the variant the fused row kernel uses) */
int chs_test_math(int device, int which, const double* a, const double* b, double* out, int64_t n);
/* Which engine the handle resolved to (CHS_ENGINE_DIRECT / CHS_ENGINE_FAST). */
int chs_engine(chs_handle h);

/* Measurement hooks used by bench.py. */
#define CHS_NKERNELS 8
/* Names of the per-step kernels of the resolved engine, slot i (NULL beyond). */
const char* chs_kernel_name(chs_handle h, int slot);
/* Run `nsteps` steps with HIP events bracketing every kernel launch on the
 * handle's stream; ms[i] receives the summed device time of slot i, calls[i]
 * its launch count.  Results are identical to chs_step_n. */
int chs_profile_steps(chs_handle h, int64_t nsteps, double ms[CHS_NKERNELS], int64_t calls[CHS_NKERNELS]);
/* Device time (ms, HIP events on the handle's stream) of the last chs_step_n. */
double chs_last_step_ms(chs_handle h);

const char* chs_last_error(void);
const char* chs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CHS_HIP_H */
