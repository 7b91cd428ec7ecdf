// chs_fast_pers.h -- k_col_pers: the column pass of a timestep (k_col<MODE_STEP>'s work) as a PERSISTENT kernel
// whose workgroups fetch the tile of their next work item into LDS while they transform the current one.
//
// Why.  k_col runs 2 workgroups per CU (256 VGPRs), each a chain  stage-in -> passes -> spectral update -> passes
// -> stage-out.  The stage-in alone is a quarter of a workgroup's life (phase stamps) and it is
// pure waiting: 64 KB per workgroup arrive at the ~11 B/cycle a CU gets while every CU asks at once.  With only one
// sibling workgroup on the CU nothing covers it.  Here every CU keeps its two workgroups for the whole launch; a
// workgroup draws half-tiles (the items k_col<MODE_STEP> gives one workgroup each) from a per-XCD queue, and while it
// is in the radix passes of item i the first half of item i+1's tile streams into a spare LDS buffer by LDS-DMA
// (global_load_lds_dwordx4: no VGPRs, no instructions besides the issue -- the register-prefetch version of round 1
// spilled); the second half follows into the exchange scratch the moment the stage-out of item i has left it.
//
// LDS map of a workgroup (fp64 N=4096: 76.8 KB, two per CU):
//   S    exchange scratch of the passes | staging of the stage-out | landing zone of round 1 (rows N/2..N-1)
//   A    landing zone of round 0 (rows 0..N/2-1) of the NEXT item            (ROWS x 16 bytes)
//   ltw  tw0[k=1] | twa | twb : the pass twiddles; the rest of tw0 (k = 2..R0-1, 7/8 of the table) does not fit any
//        more and is formed as powers of the k=1 entry (tw0_load<POW>)
// Landing-zone image: one 16-byte unit per tile row (the workgroup's C columns), unit of row 4m+e at position
// 4m + (e ^ ((m>>2)&3)): an LDS-DMA piece writes 1 KiB linearly, so the bank spread the padded image of k_col gets
// from its pitch comes from permuting the rows INSIDE each quad on the source side (the lanes of a piece fetch the
// same 16 lines as unpermuted: coalescing is unchanged); quad reads are 2-way conflicted at worst.
//
// Counting the DMA.  It is issued from inline assembly: hipcc would otherwise wait for it in front of every later
// ds_read (it cannot tell the landing zone from the exchange scratch) and at every __syncthreads.  A wavefront's
// vector-memory operations complete in order, so "all but my N youngest are done" (s_waitcnt vmcnt(N)) covers every
// DMA issued before those N -- the waits below name how many younger operations there are at least.
#pragma once

// byte address of an LDS object inside the workgroup's allocation (what DS instructions and M0 take)
__device__ __forceinline__ unsigned lds_byte_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// one LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to 1 KiB of LDS at `lds_dst` (wave-uniform)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <class C>
struct PersStage {
  using T = typename C::T;
  static constexpr int ROWS = C::N / 2;                 // rows per round = 16-byte units per landing zone
  static constexpr int NW = C::THREADS / 64;
  static constexpr int NDMA = ROWS / (64 * NW);         // DMA pieces per wavefront and round
  static constexpr int ZONE = ROWS * C::C;              // elements of a landing zone
  static constexpr int JR = C::R0 / 4;                  // pass-0 half-indices j per round
  static constexpr int S_ELEMS = (col_lds_elems<C>() > ZONE) ? col_lds_elems<C>() : ZONE;
  static constexpr int TW_ELEMS = 2 * (C::L1 + (C::RA > 1 ? (C::RA - 1) * C::L2 : 0) + (C::RB > 1 ? (C::RB - 1) * C::L3 : 0));
  static constexpr size_t LDS_BYTES = (size_t)(S_ELEMS + ZONE + TW_ELEMS) * sizeof(T);
  static constexpr bool OK = (C::C * sizeof(T) == 16) && (C::G % 64 == 0) && (ROWS % (64 * NW) == 0) && (C::R0 >= 4) &&
                             (2 * LDS_BYTES + 1024 <= 160 * 1024);
  // tile row (inside its round) held by unit p, and the unit of element e of quad mm
  static __device__ __forceinline__ int row_of_unit(int p) { const int mq = p >> 2; return 4 * mq + ((p & 3) ^ ((mq >> 2) & 3)); }
  static __device__ __forceinline__ int unit_of(int mm, int e) { return 4 * mm + (e ^ ((mm >> 2) & 3)); }
};

// the bookkeeping workgroup's code once per kernel (it has two call sites below; inlined twice it is a fifth of the
// kernel's text, and two CUs share a 64 KB instruction cache)
template <int THREADS>
__device__ __attribute__((noinline)) void step_tail_call(const TailArgs& ta, DevState* __restrict__ st, double* red) {
  step_tail_body<THREADS>(ta, st, red);
}

#define CHS_QUEUE_SETS 64   // counter sets of the item queues: launch n uses set n % 64 and clears set (n + 32) % 64

template <class C>
__global__ __launch_bounds__(C::THREADS, C::WPS) void k_col_pers(const typename C::T* __restrict__ Tin, typename C::T* __restrict__ Tout,
                                                       typename C::T* __restrict__ hat, typename C::T* __restrict__ nat,
                                                       FTables<typename C::T> tb, const double* __restrict__ lam,
                                                       const double* __restrict__ sinsq, DevState* __restrict__ st,
                                                       double* __restrict__ partE2, TailArgs ta, unsigned* __restrict__ queues,
                                                       unsigned launch) {
  using T = typename C::T;
  using V = typename C::V;
  using CS = ColStage<C>;
  using PS = PersStage<C>;
  static_assert(sizeof(T) == 8, "the persistent column pass is built for the fp64 configurations");
  __shared__ double red[32];
  __shared__ int box[4];   // [0] stop flag at entry, then the item index drawn by thread 0; [1] who runs the deferred bookkeeping
  // Every launch -- also one that finds the stop flag up -- clears the queue set of the launch 32 launches ahead.
  if (blockIdx.x == 0 && threadIdx.x < 8) queues[((launch + CHS_QUEUE_SETS / 2) % CHS_QUEUE_SETS) * 8 + threadIdx.x] = 0u;
  if (threadIdx.x == 0) box[0] = st->halt;
  __syncthreads();
  if (box[0]) return;
  T* lds = reinterpret_cast<T*>(chs_dyn_lds);
  T* S = lds;
  T* A = lds + PS::S_ELEMS;
  T* ltw = A + PS::ZONE;
  const int l0 = threadIdx.x % C::G, sub0 = threadIdx.x / C::G;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane0 = threadIdx.x & 63;
  // Gated bookkeeping (stop rules, adaptive time step): workgroup 0 does it first and then works like the others, who
  // wait for its decision in front of the spectral stage of their FIRST item.
  if (ta.enabled && ta.gate && blockIdx.x == 0) {
    step_tail_call<C::THREADS>(ta, st, reinterpret_cast<double*>(chs_dyn_lds));
    __syncthreads();
  }
  // pass twiddles -> LDS: tw0's k = 1 entries, then twa | twb (contiguous behind tw0 in the table buffer)
  {
    constexpr int N0 = 2 * C::L1, NM = PS::TW_ELEMS - N0;
    for (int i = 2 * threadIdx.x; i < N0; i += 2 * C::THREADS) *reinterpret_cast<double2*>(ltw + i) = *reinterpret_cast<const double2*>(tb.tw0 + i);
    for (int i = 2 * threadIdx.x; i < NM; i += 2 * C::THREADS) *reinterpret_cast<double2*>(ltw + N0 + i) = *reinterpret_cast<const double2*>(tb.twa + i);
  }
  FTables<T> tbp = tb;
  tbp.tw0 = ltw;
  tbp.twa = ltw + 2 * C::L1;
  tbp.twb = tbp.twa + (C::RA > 1 ? 2 * (C::RA - 1) * C::L2 : 0);

  // ---- the item queues: one per (logical) XCD = blockIdx % 8 -- workgroups are dealt round-robin over the XCDs, so
  // the two halves of a tile, consecutive in one queue, meet in one L2 (speed only, as in k_col)
  const int xcd = blockIdx.x & 7;
  constexpr int IPX = (C::N / C::C) / 8;   // items per queue
  unsigned* qctr = queues + (launch % CHS_QUEUE_SETS) * 8 + xcd;
  auto draw = [&]() { return (int)atomicAdd(qctr, 1u); };
  auto item_tile = [&](int j, int& ct, int& hh) {
    ct = xcd + 8 * (j / CS::Q); hh = j % CS::Q;
    if (ta.reverse) ct = C::N / C::CT - 1 - ct;
  };
  auto dma_round = [&](int j, int rho, T* zone) {
    int ct, hh;
    item_tile(j, ct, hh);
    const int lane = launder(lane0);
    const T* tile = Tin + (size_t)ct * C::N * C::CT + (size_t)rho * PS::ROWS * C::CT + hh * C::C;
    const unsigned base = lds_byte_addr(zone);
#pragma unroll
    for (int i = 0; i < PS::NDMA; ++i) {
      const int piece = i * PS::NW + wave;
      const int row = PS::row_of_unit(piece * 64 + lane);
      glds16(tile + (size_t)row * C::CT, __builtin_amdgcn_readfirstlane(base + piece * 1024));
    }
  };
  auto read_quads = [&](const T* zone, int rho, V* z) {
    const int l = launder(l0), sub = launder(sub0);
#pragma unroll
    for (int q = 0; q < C::NP0; ++q) {
      const int m1 = l + C::G * q, m2 = C::L1 - 1 - m1;
#pragma unroll
      for (int jj = 0; jj < PS::JR; ++jj) {
        const int j = rho * PS::JR + jj;
        T q1[4], q2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          q1[e] = zone[PS::unit_of(m1 + C::L1 * jj, e) * C::C + sub];
          q2[e] = zone[PS::unit_of(m2 + C::L1 * jj, e) * C::C + sub];
        }
        pack_quads<C>(q1, q2, q, j, z);
      }
    }
  };

  // items are drawn two ahead: the number of the next one must be at hand when its prefetch starts, and a draw is a
  // round trip to memory
  if (threadIdx.x == 0) { box[0] = draw(); box[2] = draw(); }
  __syncthreads();
  int item = box[0], next = box[2];
  __syncthreads();
  if (item < IPX) dma_round(item, 0, A);
  double lam1 = st->lam1, lam2 = st->lam2;  // (gated launches read them again behind the gate)
  bool gate_open = !(ta.enabled && ta.gate);

  // (diagnostic build -DCHS_STAMPS: phase stamps of every workgroup's SECOND item -- steady state -- into buffer 2)
  [[maybe_unused]] int it_no = 0;
#define PSTAMP(I) do { if (it_no == 1) STAMP(2, I); } while (0)
#pragma unroll 1
  while (item < IPX) {
    PSTAMP(0);
    // (laundered: whatever is derived from the lane indices is recomputed in every iteration instead of being
    // hoisted out of the loop and kept in registers -- or spilled -- across its whole body)
    const int l = launder(l0), sub = launder(sub0);
    int ct, hh;
    item_tile(item, ct, hh);
    const int bid = xcd + 8 * item;           // the block number k_col<MODE_STEP> would give this item
    const int kc = ct * C::CT + hh * C::C + sub;
    T* scr = S + (size_t)sub * C::SCR;
    T* hcol = hat + (size_t)kc * C::N;
    T* hout = (nat != nullptr) ? nat + (size_t)kc * C::N : hcol;
    const int kc_u = __builtin_amdgcn_readfirstlane(kc);
    const double lc = lam[kc_u];
    const double sqc = sinsq[2 * kc_u + 1];
    // round 1 of this item -> S
    dma_round(item, 1, S);
    // round 0 (issued an item ago, or just above for the first item): every operation but my NDMA youngest is done
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PS::NDMA) : "memory");
    __syncthreads();
    PSTAMP(1);
    V z[C::E];
    read_quads(A, 0, z);
    PSTAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PSTAMP(3);
    read_quads(S, 1, z);
    __syncthreads();   // A and S are free again
    PSTAMP(4);
    // the first half of the next item's tile streams into A from here on, under the passes
    if (next < IPX) dma_round(next, 0, A);
    fwd_passes<C, true>(z, scr, tbp, launder(l));
    PSTAMP(5);
    if (!gate_open) {
      if (gate_wait(st, ta.seq, ta.gate_spins, red, lam1, lam2)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no DMA in flight towards an LDS allocation that is given back)
        return;
      }
      gate_open = true;
    }
    if (threadIdx.x == 0) box[0] = draw();   // the item after next; read behind the barriers of the stage-out
    PSTAMP(6);
    // ---- recombination / spectral stage / adjoint recombination, in place per slot (as k_col<MODE_STEP>)
    struct Fetched { double2 ls[4]; V h01, h23; };
    double e2 = 0.0;
    T h00 = T(0);
    recombine<C, true, true, true>(z, tb, l,
      [&](int pbase, const int* idx) {
        Fetched p;
#pragma unroll
        for (int t = 0; t < 4; ++t) p.ls[t] = reinterpret_cast<const double2*>(sinsq)[idx[t]];
        const int hp = hat_pair_index<C>(pbase, fc_opaque(l));
        p.h01 = ldc<T>(hcol, hp);
        p.h23 = ldc<T>(hcol, hp + C::G);
        return p;
      },
      [&](int pbase, const int*, V& Ya, V& Yb, bool live, const Fetched& p) {
        T y[4] = {cx_re(Ya), cx_im(Ya), cx_re(Yb), cx_im(Yb)};
        const T hold[4] = {cx_re(p.h01), cx_im(p.h01), cx_re(p.h23), cx_im(p.h23)};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const T h = chs_spectral<T>(hold[t], y[t], p.ls[t].x, lc, lam1, lam2);
          y[t] = h;
          const double term = (double)h * (double)h * (p.ls[t].y + sqc);
          e2 += live ? term : 0.0;
          if (pbase + t == 0 && live) h00 = h;  // lane 0 holds kr = 0 at position 0 (its own slot)
        }
        asm volatile("" : "+v"(e2));  // do not postpone the 2E energy terms
        Ya = cx_make(y[0], y[1]); Yb = cx_make(y[2], y[3]);
      },
      [&](int pbase, const int*, V& Ya, V& Yb, bool live) {
        if (live) {
          const int hp = hat_pair_index<C>(pbase, fc_opaque(l));
          stc<T>(hout, hp, Ya);
          stc<T>(hout, hp + C::G, Yb);
        }
      });
    if (l == 0 && kc == 0) st->meanU = (double)h00 / (double)C::N;  // ortho DC term = sum(U)/N (solver.py:223)
    PSTAMP(7);
    inv_passes<C, true>(z, scr, tbp, launder(l));
    PSTAMP(8);
    // ---- stage out through S: quads -> tile rows (k_col's padded image)
    T* tile = Tout + (size_t)ct * C::N * C::CT;
#pragma unroll
    for (int rho = 0; rho < 2; ++rho) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < C::NP0; ++q) {
        const int m1 = launder(l) + C::G * q, m2 = C::L1 - 1 - m1;
#pragma unroll
        for (int jj = 0; jj < CS::JR; ++jj) {
          const int j = rho * CS::JR + jj;
          T q1[4], q2[4];
          unpack_quads<C>(z, q, j, q1, q2);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            S[(m1 + C::L1 * jj) * CS::LP + e * C::C + sub] = q1[e];
            S[(m2 + C::L1 * jj) * CS::LP + e * C::C + sub] = q2[e];
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < CS::PER; ++i) {
        const int f = CS::PW * (launder((int)threadIdx.x) + i * C::THREADS);
        const int lo = CS::loff(f);
        T* dst = tile + CS::goff(rho, f, hh);
        const T a = S[lo], b = S[lo + 1];
        *reinterpret_cast<double2*>(dst) = make_double2(a, b);
      }
    }
    PSTAMP(9);
    // the tile's share of the spectral gradient sum (one barrier: S is free behind it)
    const double acc1[1] = {e2};
    double tot1[1];
    block_sum_store<1, C::THREADS / 64>(acc1, red, tot1);
    if (threadIdx.x == 0) partE2[bid] = tot1[0];
    PSTAMP(10);
    item = next;
    next = box[0];
    ++it_no;
  }
#undef PSTAMP
  // ---- the deferred bookkeeping (nobody waits for it): the first workgroup of the launch to run out of items does
  // it while the others finish theirs -- it fills the gap the end of the launch leaves anyway
  if (ta.enabled && !ta.gate) {
    if (threadIdx.x == 0) box[1] = (atomicExch(&st->tail_claim, ta.claim_seq) != ta.claim_seq) ? 1 : 0;
    __syncthreads();
    if (box[1]) step_tail_call<C::THREADS>(ta, st, reinterpret_cast<double*>(chs_dyn_lds));
  }
}
