// gnx_exp.h — float64 exp() whose constants never occupy vector registers (device helper shared by k_gnofix.hip and k_smooth_crf.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

// The device library's exp(), inlined inside loops, keeps its 18 reduction / polynomial constants in VECTOR registers for the whole
// kernel once the scalar file is full (k_gnofix: 18 of 128; k_smooth_crf_ck: 20 of 168).  Here every constant is moved into a scalar
// pair by a volatile asm right where it is used, so nothing can be hoisted, and the multiply-add takes it as its one scalar operand:
// Cody-Waite reduction (n = rint(x / ln 2), r = x - n ln 2 in two parts, fdlibm's split: n ln2_hi is exact), degree-13 Taylor
// polynomial of e^r (|r| <= 0.347: truncation 1.3e-17 relative), v_ldexp_f64 (overflow -> inf, underflow -> denormal / 0, NaN -> NaN,
// like exp).  Within 1 ulp of libm on [-745, 710] (2e7 arguments checked on the host against glibc).
#if defined(__HIP_DEVICE_COMPILE__)
template <uint32_t LO, uint32_t HI>
__device__ __forceinline__ double gnx_sconst() {  // a float64 constant in a scalar pair the optimiser cannot see through (nor hoist)
  uint32_t lo, hi;
  asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "n"(LO), "n"(HI));
  return __hiloint2double((int)hi, (int)lo);
}
// N arguments at once: every constant is moved into its scalar pair ONCE for all of them (1 / N of the scalar instructions per
// result — they are issue slots of the wave's in-order stream like any other) and the N dependent chains interleave.
template <uint32_t LO, uint32_t HI, int N>
__device__ __forceinline__ void gnx_fma_sc(double (&a)[N], const double (&b)[N]) {  // a = a * b + constant (the instruction's scalar operand)
  const double c = gnx_sconst<LO, HI>();
#pragma unroll
  for (int i = 0; i < N; ++i) asm("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(c));
}
template <uint32_t LO, uint32_t HI, int N>
__device__ __forceinline__ void gnx_fma_cs(double (&r)[N], const double (&n)[N]) {  // r = n * constant + r
  const double c = gnx_sconst<LO, HI>();
#pragma unroll
  for (int i = 0; i < N; ++i) asm("v_fma_f64 %0, %1, %2, %0" : "+v"(r[i]) : "v"(n[i]), "s"(c));
}
template <int N>
__device__ __forceinline__ void gnx_exp_scN(double (&x)[N]) {  // x[i] = exp(x[i])
  double n[N], p[N];
  {
    const double c = gnx_sconst<0x652b82feu, 0x3ff71547u>();  // 1 / ln 2
#pragma unroll
    for (int i = 0; i < N; ++i) asm("v_mul_f64 %0, %1, %2" : "=v"(n[i]) : "v"(x[i]), "s"(c));
  }
#pragma unroll
  for (int i = 0; i < N; ++i) n[i] = rint(n[i]);
  gnx_fma_cs<0xfee00000u, 0xbfe62e42u, N>(x, n);   // r = x - n ln 2: high part (fdlibm's split: n ln2_hi is exact)
  gnx_fma_cs<0x35793c76u, 0xbdea39efu, N>(x, n);   // low part
  {
    const double c = gnx_sconst<0x13a86d09u, 0x3de61246u>();  // 1/13!
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = c;
  }
  gnx_fma_sc<0xeff8d898u, 0x3e21eed8u, N>(p, x);   // 1/12!
  gnx_fma_sc<0x67f544e4u, 0x3e5ae645u, N>(p, x);   // 1/11!
  gnx_fma_sc<0xb7789f5cu, 0x3e927e4fu, N>(p, x);   // 1/10!
  gnx_fma_sc<0xa556c734u, 0x3ec71de3u, N>(p, x);   // 1/9!
  gnx_fma_sc<0x1a01a01au, 0x3efa01a0u, N>(p, x);   // 1/8!
  gnx_fma_sc<0x1a01a01au, 0x3f2a01a0u, N>(p, x);   // 1/7!
  gnx_fma_sc<0x16c16c17u, 0x3f56c16cu, N>(p, x);   // 1/6!
  gnx_fma_sc<0x11111111u, 0x3f811111u, N>(p, x);   // 1/5!
  gnx_fma_sc<0x55555555u, 0x3fa55555u, N>(p, x);   // 1/4!
  gnx_fma_sc<0x55555555u, 0x3fc55555u, N>(p, x);   // 1/3!
#pragma unroll
  for (int i = 0; i < N; ++i) {
    p[i] = fma(p[i], x[i], 0.5);
    p[i] = fma(p[i], x[i], 1.0);
    p[i] = fma(p[i], x[i], 1.0);
    x[i] = __builtin_ldexp(p[i], (int)n[i]);
  }
}
__device__ __forceinline__ double gnx_exp_sc(double x) {
  double v[1] = {x};
  gnx_exp_scN<1>(v);
  return v[0];
}

// ---- the logistic bases' sigmoid and row normaliser: ONE definition for every logistic kernel (k_base_logistic_i8 / _i8_dl / _p2 /
// _p2f are bit-identical to each other and stay so).  The compiler's IEEE division is eleven float64 instructions (two v_div_scale,
// v_rcp, four fma, mul, fma, v_div_fmas, v_div_fixup) and a logistic output divided twice — 1 / (1 + e^-t), then p / sum(p): the
// epilogues were a fifth of the 2-bit passes (the shorter sequence below changed no kernel's time measurably: DESIGN.md 8).
// The operands here are benign (1 + e^-t in [1, 3e307], sum(p) in [3e-308, A]: normal numbers, no scaling needed):
//   gnx_rcp_nr(y)   v_rcp_f64 + two Newton steps: within 1 ulp of 1 / y (five instructions)
//   gnx_sigmoid     1 / (1 + e^-t) with -t capped at 708 (e^708 = 3e307 stays finite: beyond it the reference's own value is below
//                   1e-307 — or, past 709.78, exactly 0 — and its row normaliser 0 / 0)
//   normaliser      p * gnx_rcp_nr(sum): one reciprocal per row, one multiplication per class
// against the reference's float64 division each result differs by at most ~2 ulp; the tests' bar on B is 1e-12.
__device__ __forceinline__ double gnx_rcp_nr(double y) {
  double r = __builtin_amdgcn_rcp(y);
  double e = __builtin_fma(-y, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-y, r, 1.0);
  return __builtin_fma(r, e, r);
}
template <int N>
__device__ __forceinline__ void gnx_sigmoidN(double (&v)[N]) {  // in: t (logit + intercept); out: 1 / (1 + e^-t); N independent chains
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __builtin_fmin(-v[i], 708.0);
  gnx_exp_scN<N>(v);
  double r[N], e[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { v[i] = 1.0 + v[i]; r[i] = __builtin_amdgcn_rcp(v[i]); }
#pragma unroll
  for (int i = 0; i < N; ++i) e[i] = __builtin_fma(-v[i], r[i], 1.0);
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = __builtin_fma(r[i], e[i], r[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) e[i] = __builtin_fma(-v[i], r[i], 1.0);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __builtin_fma(r[i], e[i], r[i]);
}
__device__ __forceinline__ double gnx_sigmoid(double t) {
  double v[1] = {t};
  gnx_sigmoidN<1>(v);
  return v[0];
}
#else   // host pass: never called
__device__ inline double gnx_exp_sc(double x) { return x; }
template <int N>
__device__ inline void gnx_exp_scN(double (&)[N]) {}
__device__ inline double gnx_rcp_nr(double y) { return y; }
template <int N>
__device__ inline void gnx_sigmoidN(double (&)[N]) {}
__device__ inline double gnx_sigmoid(double t) { return t; }
#endif

// xgboost's Softmax exponentiates a float32 margin difference with expf; glibc's expf is correctly rounded in all but a vanishing
// share of cases, which a float64 evaluation rounded to float32 reproduces (the oracle does the same).  ONE function for every softmax
// of the tree smoother on the device — the initial smoother pass, Gnofix's candidates and its re-evaluation — so that equal margins
// give equal probabilities whichever kernel evaluates them.
__device__ __forceinline__ float gnx_softmax_exp(float d) { return (float)gnx_exp_sc((double)d); }
