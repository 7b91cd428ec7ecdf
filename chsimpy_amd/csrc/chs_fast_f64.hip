// chs_fast_f64.hip -- the fp64 configurations of the fast transform engine (a translation unit of its own:
// the kernel instantiations of the two element types compile in parallel).
#if !defined(CHS_STAMPS) || defined(CHS_FAST_UNITY_INCLUDE)
#include "chs_fast_kernels.h"

// fp64 configurations: <T, N, G, THREADS, R0, RA, RB, RL, pad1, pad2, padL, waves/SIMD>
// Small grids are bound by one workgroup life per launch (a handful of workgroups, nothing to overlap
// with): 8 complex values per lane instead of 16 -- radix-4 end passes -- halve the dependent
// instruction stream of a lane and double the number of workgroups (CHS_SMALL_E8).
// ... and small workgroups: the grid spreads over more CUs (N=512: 128 workgroups of 4 transforms instead of 64 of
// 8) -- N=128 20.3 -> 16.1 us/step, N=256 21.3 -> 16.9, N=512 24.0 -> 20.8 on one box; N=1024 wants 256 threads
// (37.5 us with 128, 32.1 with 256), and so does N=2048 (128 threads there mean 2-column tiles, 16-byte pieces
// for the row kernels: 87.7 instead of 59.3 us).
using F128 = FCfg<double, 128, 8, 64, 4, 4, 1, 4, 1, 0, 1, 2>;
using F256 = FCfg<double, 256, 16, 64, 4, 8, 1, 4, 1, 0, 1, 2>;
using F512 = FCfg<double, 512, 32, 128, 4, 4, 4, 4, 1, 1, 1, 2>;
// k_col likes two transforms per workgroup (two of the 4 columns of a tile), the row kernels four (13.1 / 12.7 us against 14.3 / 15.3)
using F512C = FCfg<double, 512, 32, 64, 4, 4, 4, 4, 1, 1, 1, 2, 4>;
using F1024 = FCfg<double, 1024, 64, 256, 4, 8, 4, 4, 2, 1, 4, 2>;
// N = 2048: 16 complex values per lane (an 8-per-lane variant with two wavefronts per transform measured equal)
// (paired 16-byte exchange items, FCfg::PAIR, with k_col's pass-0 twiddles as powers to make room: 17.39 k -> 17.60 k
// steps/s for a single run, at 76 instead of 35 KB of LDS per workgroup -- not taken: members of an ensemble share the
// CUs; the powers alone: no change.  profiles/r03_ab_xpair.txt)
using F2048 = FCfg<double, 2048, 64, 256, 8, 16, 1, 8, 2, 0, 8, 2>;
using F2048C = F2048;
#ifndef CHS_F2048C_LAM_SGPR
#define CHS_F2048C_LAM_SGPR 0   // (with it: 16 bytes of scratch)
#endif
template <> struct ColLamSgpr<F2048C> { static constexpr bool value = (CHS_F2048C_LAM_SGPR != 0); };
// (the pass twiddles of the row kernel in LDS, which pay from N = 4096 upwards, cost 1 % here: 17.1 k against 17.25 k steps/s)
// N = 4096: two wavefronts per transform, 16 complex values per lane, four radix passes
#ifndef CHS_ROW_WPS
#define CHS_ROW_WPS 4
#endif
#ifndef CHS_ROW_THREADS
#define CHS_ROW_THREADS 256
#endif
// row kernels: CHS_ROW_THREADS/128 rows per workgroup, tiles of 4 columns
#ifndef CHS_ROW_PADL
#define CHS_ROW_PADL 4   // (16 before round 3: 2 KB of scratch per workgroup that the middle-pass twiddles now use)
#endif
#ifndef CHS_F4096_CT
#define CHS_F4096_CT 4  // columns per tile of the T layout (8: 64-byte row pieces; measured)
#endif
#ifndef CHS_PAD1
#define CHS_PAD1 2
#endif
#ifndef CHS_PAD2
#define CHS_PAD2 1
#endif
#ifndef CHS_COL_PADL
#define CHS_COL_PADL 16
#endif
using F4096 = FCfg<double, 4096, 128, CHS_ROW_THREADS, 8, 4, 8, 8, CHS_PAD1, CHS_PAD2, CHS_ROW_PADL, CHS_ROW_WPS, CHS_F4096_CT>;
// The fused row kernel keeps the twiddles of its two middle passes in LDS (4 KB; with the pad of the last exchange at 4
// instead of 16 four workgroups still fit a CU: 4 x 39.8 KB): row kernel 102.6 -> 94.4 us, 4544 -> 4750 steps/s on one
// box (profiles/r03_ab_occupancy.txt); the smaller pad alone changes nothing.
#ifndef CHS_F4096_ROW_TW_LDS
#define CHS_F4096_ROW_TW_LDS 3
#endif
template <> struct RowTwLds<F4096> { static constexpr int value = CHS_F4096_ROW_TW_LDS; };
// k_col runs best with the full register file of two waves per SIMD (no spills; the compiler
// uses the room to keep more loads in flight): measured 305 -> 191 us per launch
#ifndef CHS_COL_WPS
#define CHS_COL_WPS 2
#endif
#ifndef CHS_COL_THREADS
#define CHS_COL_THREADS 256
#endif
// k_col: CHS_COL_THREADS/128 of the 4 columns of a tile per workgroup
// (paired 16-byte exchange items -- 27 instead of 39 barriers, 463 instead of 559 LDS instructions, 78 instead of 45 KB
// of LDS -- measured equal: 4829 against 4829 steps/s over three interleaved rounds, profiles/r03_ab_xpair.txt)
// CHS_F4096C_E32=1 (experiment, VERDICT round 3 item 1b, unrolled form): one wavefront per column, 32 values per lane, radices
// 16.8.16 -- two exchanges instead of three, no workgroup barrier inside the passes (wave-local groups), 128-thread workgroups
#ifndef CHS_F4096C_E32
#define CHS_F4096C_E32 0
#endif
#if CHS_F4096C_E32
using F4096C = FCfg<double, 4096, 64, 128, 16, 8, 1, 16, CHS_PAD1, CHS_PAD2, CHS_COL_PADL, CHS_COL_WPS, CHS_F4096_CT>;
#else
using F4096C = FCfg<double, 4096, 128, CHS_COL_THREADS, 8, 4, 8, 8, CHS_PAD1, CHS_PAD2, CHS_COL_PADL, CHS_COL_WPS, CHS_F4096_CT>;
#endif
// pass-0 twiddles: the k = 1 entries in LDS, the others as their powers (tw0_load<POW>): k_col 122-124 -> 119-121 us
// against the whole table in LDS (profiles/r03_ab_tw2.txt); all from L2: 139 us
#ifndef CHS_F4096C_TW_LDS
#define CHS_F4096C_TW_LDS 2
#endif
template <> struct ColTwLds<F4096C> { static constexpr int value = CHS_F4096C_TW_LDS; };
#ifndef CHS_F4096C_LAM_SGPR
#define CHS_F4096C_LAM_SGPR 1
#endif
template <> struct ColLamSgpr<F4096C> { static constexpr bool value = (CHS_F4096C_LAM_SGPR != 0); };
// Stage-in by LDS-DMA (round 4; chs_fast_kernels.h: DmaStage) in this one-workgroup-per-item kernel -- every piece of
// both halves of the tile requested at kernel entry, one wait, no staging registers, 32 ds_write_b64 per thread fewer,
// parity green -- measured 2.8 % SLOWER per step (k_col +5 %), in-process A/B on one box: 0.2201 against 0.2139 ms
// (profiles/r04_ab_dma.txt).  Off; kept as a switch.
#ifndef CHS_F4096C_DMA
#define CHS_F4096C_DMA 0
#endif
template <> struct ColDma<F4096C> { static constexpr bool value = (CHS_F4096C_DMA != 0); };

// fp64 at N = 8192: the shape of the fp32 configuration of that size (four wavefronts per transform,
// two rows or two of a tile's four columns per 512-thread workgroup)
// (tiles of 8 columns here: the arrays do not fit the Infinity Cache at this size and the row kernels' 32-byte
// pieces cost more than k_col's sharing of a line among four workgroups -- row 585 -> 482 us, k_col 680 -> 755 us)
#ifndef CHS_F8192_CT
#define CHS_F8192_CT 8
#endif
using F8192 = FCfg<double, 8192, 256, 512, 8, 8, 8, 8, 2, 1, 16, 4, CHS_F8192_CT>;
using F8192C = FCfg<double, 8192, 256, 512, 8, 8, 8, 8, 2, 1, 16, 2, CHS_F8192_CT>;
// k_col (one 512-thread workgroup per CU either way): pass-0 twiddles as powers of the k = 1 entries in LDS 729-736 us,
// the whole table in LDS 754-756, all from L2 767-768; one column per 256-thread workgroup, two per CU: 790-794
// (profiles/r03_ab_f8192.txt)
#ifndef CHS_F8192C_TW_LDS
#define CHS_F8192C_TW_LDS 2
#endif
template <> struct ColTwLds<F8192C> { static constexpr int value = CHS_F8192C_TW_LDS; };
template <> struct ColLamSgpr<F8192C> { static constexpr bool value = false; };   // (with it: 44 bytes of scratch)
#ifndef CHS_F8192_ROW_TW_LDS
#define CHS_F8192_ROW_TW_LDS 3
#endif
template <> struct RowTwLds<F8192> { static constexpr int value = CHS_F8192_ROW_TW_LDS; };


bool chs_fast_bind_f64(int N, FastPlan* P) {
  switch (N) {
    case 128: bind<F128>(P); break;
    case 256: bind<F256>(P); break;
    case 512: bind<F512, F512C>(P); break;
    case 1024: bind<F1024>(P); break;
    case 2048: bind<F2048, F2048C>(P); break;
    case 4096: bind<F4096, F4096C>(P); break;
    case 8192: bind<F8192, F8192C>(P); break;
    default: return false;
  }
  return true;
}
#endif
