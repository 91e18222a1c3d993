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
template <uint32_t LO, uint32_t HI>
__device__ __forceinline__ double gnx_fma_sc(double a, double b) {  // a * b + constant, the constant as the instruction's scalar operand
  const double c = gnx_sconst<LO, HI>();
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
  return d;
}
template <uint32_t LO, uint32_t HI>
__device__ __forceinline__ double gnx_fma_cs(double a, double b) {  // a * constant + b
  const double c = gnx_sconst<LO, HI>();
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(c), "v"(b));
  return d;
}
__device__ __forceinline__ double gnx_exp_sc(double x) {
  double n;
  {
    const double c = gnx_sconst<0x652b82feu, 0x3ff71547u>();  // 1 / ln 2
    asm("v_mul_f64 %0, %1, %2" : "=v"(n) : "v"(x), "s"(c));
  }
  n = rint(n);
  double r = gnx_fma_cs<0xfee00000u, 0xbfe62e42u>(n, x);   // -ln 2, high part
  r = gnx_fma_cs<0x35793c76u, 0xbdea39efu>(n, r);          // -ln 2, low part
  double p = gnx_fma_sc<0xeff8d898u, 0x3e21eed8u>(gnx_sconst<0x13a86d09u, 0x3de61246u>(), r);  // 1/13! r + 1/12!
  p = gnx_fma_sc<0x67f544e4u, 0x3e5ae645u>(p, r);          // 1/11!
  p = gnx_fma_sc<0xb7789f5cu, 0x3e927e4fu>(p, r);          // 1/10!
  p = gnx_fma_sc<0xa556c734u, 0x3ec71de3u>(p, r);          // 1/9!
  p = gnx_fma_sc<0x1a01a01au, 0x3efa01a0u>(p, r);          // 1/8!
  p = gnx_fma_sc<0x1a01a01au, 0x3f2a01a0u>(p, r);          // 1/7!
  p = gnx_fma_sc<0x16c16c17u, 0x3f56c16cu>(p, r);          // 1/6!
  p = gnx_fma_sc<0x11111111u, 0x3f811111u>(p, r);          // 1/5!
  p = gnx_fma_sc<0x55555555u, 0x3fa55555u>(p, r);          // 1/4!
  p = gnx_fma_sc<0x55555555u, 0x3fc55555u>(p, r);          // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)n);
}
#else   // host pass: never called
__device__ inline double gnx_exp_sc(double x) { return x; }
#endif

// xgboost's Softmax exponentiates a float32 margin difference with expf; glibc's expf is correctly rounded in all but a vanishing
// share of cases, which a float64 evaluation rounded to float32 reproduces (the oracle does the same).  ONE function for every softmax
// of the tree smoother on the device — the initial smoother pass, Gnofix's candidates and its re-evaluation — so that equal margins
// give equal probabilities whichever kernel evaluates them.
__device__ __forceinline__ float gnx_softmax_exp(float d) { return (float)gnx_exp_sc((double)d); }
