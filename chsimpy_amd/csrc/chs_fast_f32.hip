// chs_fast_f32.hip -- the fp32 configurations of the fast transform engine (see chs_fast_f64.hip).
#if !defined(CHS_STAMPS) || defined(CHS_FAST_UNITY_INCLUDE)
#include "chs_fast_kernels.h"

// fp32 configurations (BASELINE.json configs[3]: N = 8192 fp32): a complex value is one packed register pair
// (chs_cx.h), so 32 values per lane occupy the 64 VGPRs of 16 fp64 ones; reductions stay in fp64.
//
// N = 8192 fp32: two wavefronts per transform, 32 complex values per lane, THREE radix-16 passes (two exchanges
// instead of the three of 8.8.8.8: the exchanges with their workgroup barriers, not the packed arithmetic, are what
// the passes of this size cost -- profiles/r03_stamps_n8192_fp32.txt), 256-thread workgroups: two rows, or two of a
// tile's eight columns.  k_col keeps only the k = 1 entries of the pass-0 twiddle table (and the middle-pass table)
// in LDS and forms the other 14 as their powers: the whole table (30 KB) would cost the second workgroup of a CU.
// Measured on one box (profiles/r03_ab_occupancy.txt): 256 lanes x 16 values, 8.8.8.8: k_col 385-405 us, 1512-1554
// steps/s; this shape with all pass twiddles from L2 349 us, 1660; with the k = 1 entries in LDS 313 us, 1763; with the
// whole table in LDS (one workgroup per CU) 475 us.  The rows take 252-254 us in either shape.
#ifndef CHS_G8192_WPS
#define CHS_G8192_WPS 3
#endif
#ifndef CHS_G8192C_WPS
#define CHS_G8192C_WPS 2
#endif
#ifndef CHS_F32_CT
#define CHS_F32_CT 8  // columns per tile of the fp32 T layout at N >= 2048: 32-byte row pieces as in fp64 (4: N=8192 23 % slower)
#endif
#ifndef CHS_G8192C_TW_LDS
#define CHS_G8192C_TW_LDS 2
#endif
// The row kernels exchange the real and the imaginary parts one after the other (as fp64 does): half the exchange
// scratch, three workgroups per CU instead of two -- fused row kernel 217 -> 206 us (profiles/r03_ab_split.txt); k_col,
// whose staging buffer is as large as its paired scratch, gains nothing from it (split exchanges with the whole pass-0
// table in LDS: 343 against 316 us; with three workgroups per CU, 140 B of spills: 370-382; 4 columns per 512-thread
// workgroup, 280 B of spills: 359).
#ifndef CHS_G8192_XPAIR
#define CHS_G8192_XPAIR 0
#endif
using G8192 = FCfg<float, 8192, 128, 256, 16, 16, 1, 16, 2, 1, 16, CHS_G8192_WPS, CHS_F32_CT, CHS_G8192_XPAIR>;
#ifndef CHS_G8192C_THREADS
#define CHS_G8192C_THREADS 256
#endif
#ifndef CHS_G8192C_XPAIR
#define CHS_G8192C_XPAIR -1
#endif
using G8192C = FCfg<float, 8192, 128, CHS_G8192C_THREADS, 16, 16, 1, 16, 2, 1, 16, CHS_G8192C_WPS, CHS_F32_CT, CHS_G8192C_XPAIR>;
template <> struct ColTwLds<G8192C> { static constexpr int value = CHS_G8192C_TW_LDS; };
#ifndef CHS_G8192_ROW_TW_LDS
#define CHS_G8192_ROW_TW_LDS 2
#endif
template <> struct RowTwLds<G8192> { static constexpr int value = CHS_G8192_ROW_TW_LDS; };
#ifndef CHS_G4096_THREADS
#define CHS_G4096_THREADS 256
#endif
// N = 4096 fp32: the shape of fp64 N = 4096 (64 lanes x 32 values with 16.8.16, wave-local exchanges, measured equal in
// k_col and 6 % slower in the rows)
using G4096 = FCfg<float, 4096, 128, CHS_G4096_THREADS, 8, 4, 8, 8, 2, 1, 16, 4, CHS_F32_CT>;
using G4096C = FCfg<float, 4096, 128, CHS_G4096_THREADS, 8, 4, 8, 8, 2, 1, 16, 2, CHS_F32_CT>;
#ifndef CHS_G4096C_TW_LDS
#define CHS_G4096C_TW_LDS 2  // k = 1 entries + powers: k_col 63.5 -> 61.8 us against the whole table (profiles/r03_ab_tw2.txt)
#endif
template <> struct ColTwLds<G4096C> { static constexpr int value = CHS_G4096C_TW_LDS; };
#ifndef CHS_G4096_ROW_TW_LDS
#define CHS_G4096_ROW_TW_LDS 3
#endif
template <> struct RowTwLds<G4096> { static constexpr int value = CHS_G4096_ROW_TW_LDS; };
// fp32 below N = 4096: the shapes of the fp64 configurations (groups inside one wavefront)
// (8 complex values per lane with radix-4 end passes and small workgroups, as in fp64: chs_fast_f64.hip)
using G128 = FCfg<float, 128, 8, 64, 4, 4, 1, 4, 1, 0, 1, 2>;
using G256 = FCfg<float, 256, 16, 64, 4, 8, 1, 4, 1, 0, 1, 2>;
using G512 = FCfg<float, 512, 32, 128, 4, 4, 4, 4, 1, 1, 1, 2>;
using G1024 = FCfg<float, 1024, 64, 256, 4, 8, 4, 4, 2, 1, 4, 2>;
// N=512: k_col in smaller workgroups than the row kernels (two of a tile's four columns), as in fp64: 18.0 -> 17.2 us
// (N=1024 the same way, both types: no gain)
using G512C = FCfg<float, 512, 32, 64, 4, 4, 4, 4, 1, 1, 1, 2, 4>;
#ifndef CHS_F32_CT_SMALL
#define CHS_F32_CT_SMALL 8  // N = 2048 fp32: 8 columns per tile (32-byte row pieces; +7 % against 4)
#endif
using G2048 = FCfg<float, 2048, 64, 256, 8, 16, 1, 8, 2, 0, 8, 2, (CHS_F32_CT_SMALL ? CHS_F32_CT_SMALL : 4)>;
// (row-kernel twiddles in LDS: 26.5 k -> 23.5-24.2 k steps/s here: a copy and a block barrier per workgroup for groups that
// otherwise need none)

bool chs_fast_bind_f32(int N, FastPlan* P) {
  switch (N) {
    case 128: bind<G128>(P); break;
    case 256: bind<G256>(P); break;
    case 512: bind<G512, G512C>(P); break;
    case 1024: bind<G1024>(P); break;
    case 2048: bind<G2048>(P); break;
    case 4096: bind<G4096, G4096C>(P); break;
    case 8192: bind<G8192, G8192C>(P); break;
    default: return false;
  }
  return true;
}
#endif
