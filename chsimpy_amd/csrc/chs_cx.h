// chs_cx.h -- the two-component value the transform core computes on, and its arithmetic.
//
// A complex value (re, im) of the FFT network -- and, in the pointwise part of the row kernels, a PAIR of
// neighbouring real grid points -- lives in
//   fp64:  struct D2 { double x, y; }     two independent 64-bit registers; every operation below compiles to the
//                                          scalar v_*_f64 instructions (negations and component choices are operand
//                                          modifiers / register names: free);
//   fp32:  v2f (ext_vector float2)         ONE 64-bit VGPR pair; every operation below is ONE packed instruction
//                                          (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32): the component choice is the
//                                          instruction's op_sel / op_sel_hi, the negations its neg_lo / neg_hi.
// So a complex add is one instruction, a rotation by -+i or a conjugation costs nothing (it is folded into the
// instruction that consumes the value) and a complex product is two instructions -- half the fp32 instruction
// stream of the component-wise code, with no moves to pair operands up (the pair is also what travels through the
// LDS exchange as one 8-byte item).  hipcc does not fold mixed swaps/negations into packed-f32 modifiers by
// itself, hence the inline assembly; the asm statements are pure (no `volatile`): the compiler may still
// schedule, CSE and delete them.
//
// pk_add / pk_mul / pk_fma<S0, H0, S1, H1, NL0, NH0, NL1, NH1>(a, b[, c]):
//   d.x = op( +-a[S0], +-b[S1] [, c.x] )     S* = which component feeds the LOW result  (0 = x, 1 = y)
//   d.y = op( +-a[H0], +-b[H1] [, c.y] )     H* = which component feeds the HIGH result
//   NL* / NH* = negate that operand in the low / high result.
#pragma once
#include <hip/hip_runtime.h>

typedef float v2f __attribute__((ext_vector_type(2)));
struct D2 {
  double x, y;
};

template <typename T> struct CxSel;
template <> struct CxSel<double> { using type = D2; };
template <> struct CxSel<float> { using type = v2f; };
template <typename T> using Cx = typename CxSel<T>::type;

__device__ __forceinline__ D2 cx_make(double r, double i) { D2 d; d.x = r; d.y = i; return d; }
__device__ __forceinline__ v2f cx_make(float r, float i) { v2f d; d.x = r; d.y = i; return d; }

// ---- fp64: component-wise ---------------------------------------------------------------------------------
template <int S0, int H0, int S1, int H1, int NL0, int NH0, int NL1, int NH1>
__device__ __forceinline__ D2 pk_add(D2 a, D2 b) {
  const double al = S0 ? a.y : a.x, ah = H0 ? a.y : a.x, bl = S1 ? b.y : b.x, bh = H1 ? b.y : b.x;
  D2 d;
  d.x = (NL0 ? -al : al) + (NL1 ? -bl : bl);
  d.y = (NH0 ? -ah : ah) + (NH1 ? -bh : bh);
  return d;
}
template <int S0, int H0, int S1, int H1, int NL0, int NH0, int NL1, int NH1>
__device__ __forceinline__ D2 pk_mul(D2 a, D2 b) {
  const double al = S0 ? a.y : a.x, ah = H0 ? a.y : a.x, bl = S1 ? b.y : b.x, bh = H1 ? b.y : b.x;
  D2 d;
  d.x = (NL0 ? -al : al) * (NL1 ? -bl : bl);
  d.y = (NH0 ? -ah : ah) * (NH1 ? -bh : bh);
  return d;
}
template <int S0, int H0, int S1, int H1, int NL0, int NH0, int NL1, int NH1>
__device__ __forceinline__ D2 pk_fma(D2 a, D2 b, D2 c) {
  const double al = S0 ? a.y : a.x, ah = H0 ? a.y : a.x, bl = S1 ? b.y : b.x, bh = H1 ? b.y : b.x;
  D2 d;
  d.x = __builtin_fma(NL0 ? -al : al, NL1 ? -bl : bl, c.x);
  d.y = __builtin_fma(NH0 ? -ah : ah, NH1 ? -bh : bh, c.y);
  return d;
}
// second operand a compile-time constant (fp64: literals of the instructions)
template <int S0, int H0, int S1, int H1, int NL0, int NH0, int NL1, int NH1>
__device__ __forceinline__ D2 pk_mul_k(D2 a, D2 k) { return pk_mul<S0, H0, S1, H1, NL0, NH0, NL1, NH1>(a, k); }
template <int S0, int H0, int S1, int H1, int NL0, int NH0, int NL1, int NH1>
__device__ __forceinline__ D2 pk_fma_k(D2 a, D2 k, D2 c) { return pk_fma<S0, H0, S1, H1, NL0, NH0, NL1, NH1>(a, k, c); }

// ---- fp32: one packed instruction each ----------------------------------------------------------------------
#define CHS_PK_MODS2 " op_sel:[%3,%4] op_sel_hi:[%5,%6] neg_lo:[%7,%8] neg_hi:[%9,%10]"
#define CHS_PK_MODS3 " op_sel:[%4,%5,0] op_sel_hi:[%6,%7,1] neg_lo:[%8,%9,0] neg_hi:[%10,%11,0]"
#define CHS_PK_IMMS "n"(S0), "n"(S1), "n"(H0), "n"(H1), "n"(NL0), "n"(NL1), "n"(NH0), "n"(NH1)
template <int S0, int H0, int S1, int H1, int NL0, int NH0, int NL1, int NH1>
__device__ __forceinline__ v2f pk_add(v2f a, v2f b) {
  v2f d;
  asm("v_pk_add_f32 %0, %1, %2" CHS_PK_MODS2 : "=v"(d) : "v"(a), "v"(b), CHS_PK_IMMS);
  return d;
}
template <int S0, int H0, int S1, int H1, int NL0, int NH0, int NL1, int NH1>
__device__ __forceinline__ v2f pk_mul(v2f a, v2f b) {
  v2f d;
  asm("v_pk_mul_f32 %0, %1, %2" CHS_PK_MODS2 : "=v"(d) : "v"(a), "v"(b), CHS_PK_IMMS);
  return d;
}
template <int S0, int H0, int S1, int H1, int NL0, int NH0, int NL1, int NH1>
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) {
  v2f d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" CHS_PK_MODS3 : "=v"(d) : "v"(a), "v"(b), "v"(c), CHS_PK_IMMS);
  return d;
}
// second operand a constant: an SGPR pair (gfx950 packed-f32 instructions take no literal)
template <int S0, int H0, int S1, int H1, int NL0, int NH0, int NL1, int NH1>
__device__ __forceinline__ v2f pk_mul_k(v2f a, v2f k) {
  v2f d;
  asm("v_pk_mul_f32 %0, %1, %2" CHS_PK_MODS2 : "=v"(d) : "v"(a), "s"(k), CHS_PK_IMMS);
  return d;
}
template <int S0, int H0, int S1, int H1, int NL0, int NH0, int NL1, int NH1>
__device__ __forceinline__ v2f pk_fma_k(v2f a, v2f k, v2f c) {
  v2f d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" CHS_PK_MODS3 : "=v"(d) : "v"(a), "s"(k), "v"(c), CHS_PK_IMMS);
  return d;
}

// ---- complex arithmetic on either representation --------------------------------------------------------------
//                                                                S0 H0 S1 H1 NL0 NH0 NL1 NH1
template <class V> __device__ __forceinline__ V cx_add(V a, V b)    { return pk_add<0, 1, 0, 1, 0, 0, 0, 0>(a, b); }  // a + b
template <class V> __device__ __forceinline__ V cx_sub(V a, V b)    { return pk_add<0, 1, 0, 1, 0, 0, 1, 1>(a, b); }  // a - b
template <class V> __device__ __forceinline__ V cx_add_mi(V a, V b) { return pk_add<0, 1, 1, 0, 0, 0, 0, 1>(a, b); }  // a + (-i) b
template <class V> __device__ __forceinline__ V cx_add_pi(V a, V b) { return pk_add<0, 1, 1, 0, 0, 0, 1, 0>(a, b); }  // a + (+i) b
template <class V> __device__ __forceinline__ V cx_addc(V a, V b)   { return pk_add<0, 1, 0, 1, 0, 0, 0, 1>(a, b); }  // a + conj(b)
template <class V> __device__ __forceinline__ V cx_subc(V a, V b)   { return pk_add<0, 1, 0, 1, 0, 0, 1, 0>(a, b); }  // a - conj(b)
template <class V> __device__ __forceinline__ V cx_sub_cj(V a, V b) { return pk_add<0, 1, 0, 1, 0, 1, 1, 0>(a, b); }  // conj(a - b)
// a * w
template <class V> __device__ __forceinline__ V cx_mul(V a, V w) {
  const V t = pk_mul<0, 0, 0, 1, 0, 0, 0, 0>(a, w);   // (ar wr,  ar wi)
  return pk_fma<1, 1, 1, 0, 0, 0, 1, 0>(a, w, t);     // (-ai wi + .., ai wr + ..)
}
// a * conj(w)
template <class V> __device__ __forceinline__ V cx_mulc(V a, V w) {
  const V t = pk_mul<0, 0, 0, 1, 0, 0, 0, 1>(a, w);   // (ar wr, -ar wi)
  return pk_fma<1, 1, 1, 0, 0, 0, 0, 0>(a, w, t);     // (ai wi + .., ai wr + ..)
}
// conj(a * w)
template <class V> __device__ __forceinline__ V cx_mul_cj(V a, V w) {
  const V t = pk_mul<0, 0, 0, 1, 0, 0, 0, 1>(a, w);   // (ar wr, -ar wi)
  return pk_fma<1, 1, 1, 0, 0, 0, 1, 1>(a, w, t);     // (-ai wi + .., -ai wr + ..)
}
// a * k for a compile-time constant k
template <class V> __device__ __forceinline__ V cx_mul_k(V a, V k) {
  const V t = pk_mul_k<0, 0, 0, 1, 0, 0, 0, 0>(a, k);
  return pk_fma_k<1, 1, 1, 0, 0, 0, 1, 0>(a, k, t);
}

// fp32: the two instructions of a complex product in ONE asm statement.  hipcc assumes that the result of an asm
// statement may be subject to the dst-sel forwarding hazard and puts an s_nop in front of an instruction that reads
// it right away (here it is not: full 32-bit writes); inside one statement the dependent pair issues back to back.
#define CHS_CX_MUL2(NAME, MULMODS, FMAMODS, KC)                                                          \
  __device__ __forceinline__ v2f NAME(v2f a, v2f w) {                                                    \
    v2f d;                                                                                                \
    asm("v_pk_mul_f32 %0, %1, %2 " MULMODS "\n\tv_pk_fma_f32 %0, %1, %2, %0 " FMAMODS                     \
        : "=&v"(d) : "v"(a), KC(w));                                                                      \
    return d;                                                                                             \
  }
//                    t = (ar wr, +-ar wi)                         d = (+-ai wi + t.x, +-ai wr + t.y)
CHS_CX_MUL2(cx_mul,    "op_sel_hi:[0,1]",              "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]", "v")
CHS_CX_MUL2(cx_mulc,   "op_sel_hi:[0,1] neg_hi:[0,1]", "op_sel:[1,1,0] op_sel_hi:[1,0,1]", "v")
CHS_CX_MUL2(cx_mul_cj, "op_sel_hi:[0,1] neg_hi:[0,1]", "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]", "v")
CHS_CX_MUL2(cx_mul_k,  "op_sel_hi:[0,1]",              "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]", "s")
#undef CHS_CX_MUL2

// (r, i) of a value as scalars
__device__ __forceinline__ double cx_re(D2 a) { return a.x; }
__device__ __forceinline__ double cx_im(D2 a) { return a.y; }
__device__ __forceinline__ float cx_re(v2f a) { return a.x; }
__device__ __forceinline__ float cx_im(v2f a) { return a.y; }
