// k_smooth_xgb_rk.hip — the sliding-window tree smoother on 16-bit RANKS (default variant of k_smooth_xgb).
//
// Same contract and same arithmetic as k_smooth_xgb.hip (slide_window + XGBClassifier.predict_proba + argmax,
// reference src/Smooth/utils.py:4-29, src/Smooth/smooth.py:40-65, src/Smooth/models.py:8-24).  The float kernel is
// bound by the LDS pipe (12 bytes fetched per lane per node: an 8-byte node and a 4-byte probability) with the VALU
// close behind.  A tree only ever asks "p < threshold", and there are at most a few ten thousand distinct thresholds
// in an ensemble, so the model loader sorts them once and
//   * a base probability p becomes r(p) = #{thresholds <= p}  (16 bits, computed while the strip is staged: a
//     1024-bucket table narrows the search to a handful of candidates, then a fixed-length branchless bisection),
//   * a node becomes ONE 32-bit word: (k+1) << 16 | byte offset of its feature in the strip, k = index of its threshold;
//     p < U[k]  <=>  r(p) < k+1, so every walk takes exactly the branch the float compare takes and lands on the same
//     leaf; the leaves stay float32 and are summed in the same order -> margins bit-identical to the float kernel.
// Per node a lane now fetches 4 + 2 bytes, and the level costs 4 VALU ops (address, address, compare, add-with-carry).
// The strip is class-major ([class][padded window] u16) so that the lanes of a wave sitting on the same node read
// consecutive halfwords.
#include <cstdio>
#include <cstdlib>

#include "gnx_internal.h"
#include "gnx_exp.h"
#include "gnx_rank.h"

namespace {

constexpr int WS = 64;  // windows per segment = one wave width

// One level for R segments at once: j = 2j + (r >= rank field of the node), as v_cmp (SDWA picks the field out of the
// node word) + v_addc (the compare's mask is the carry-in).  hipcc's own lowering of the same C++ spends ~7 VALU ops
// per level on bit bookkeeping; this is 2, which is what makes the LDS pipe the bound again.  gfx950 wants two wait
// states between a VALU write of an SGPR pair and a VALU read of it: with R >= 3 the other segments' instructions
// provide them, smaller R pads with s_nop.
template <int R>
__device__ __forceinline__ void step_rk(uint32_t* j, const uint32_t* nd, const uint32_t* r) {
  if constexpr (R == 1) {
    uint64_t c0;
    asm("v_cmp_le_u32_sdwa %[c0], %[n0], %[r0] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %[j0], %[c0], %[j0], %[j0], %[c0]"
        : [j0] "+v"(j[0]), [c0] "=&s"(c0)
        : [n0] "v"(nd[0]), [r0] "v"(r[0]));
  } else if constexpr (R == 2) {
    uint64_t c0, c1;
    asm("v_cmp_le_u32_sdwa %[c0], %[n0], %[r0] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_cmp_le_u32_sdwa %[c1], %[n1], %[r1] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 %[j0], %[c0], %[j0], %[j0], %[c0]\n\t"
        "v_addc_co_u32 %[j1], %[c1], %[j1], %[j1], %[c1]"
        : [j0] "+v"(j[0]), [j1] "+v"(j[1]), [c0] "=&s"(c0), [c1] "=&s"(c1)
        : [n0] "v"(nd[0]), [r0] "v"(r[0]), [n1] "v"(nd[1]), [r1] "v"(r[1]));
  } else if constexpr (R == 3) {
    uint64_t c0, c1, c2;
    asm("v_cmp_le_u32_sdwa %[c0], %[n0], %[r0] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_cmp_le_u32_sdwa %[c1], %[n1], %[r1] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_cmp_le_u32_sdwa %[c2], %[n2], %[r2] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_addc_co_u32 %[j0], %[c0], %[j0], %[j0], %[c0]\n\t"
        "v_addc_co_u32 %[j1], %[c1], %[j1], %[j1], %[c1]\n\t"
        "v_addc_co_u32 %[j2], %[c2], %[j2], %[j2], %[c2]"
        : [j0] "+v"(j[0]), [j1] "+v"(j[1]), [j2] "+v"(j[2]), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2)
        : [n0] "v"(nd[0]), [r0] "v"(r[0]), [n1] "v"(nd[1]), [r1] "v"(r[1]), [n2] "v"(nd[2]), [r2] "v"(r[2]));
  } else if constexpr (R == 4) {
    step_rk<2>(j, nd, r);
    step_rk<2>(j + 2, nd + 2, r + 2);
  } else if constexpr (R == 5) {
    step_rk<3>(j, nd, r);
    step_rk<2>(j + 3, nd + 3, r + 3);
  } else {
    static_assert(R == 6, "segments per strip");
    step_rk<3>(j, nd, r);
    step_rk<3>(j + 3, nd + 3, r + 3);
  }
}

// Level 0 with the level-1 node picked on the way: the root and its two children are words 1..3 of the tree, i.e. ONE
// wave-uniform ds_read_b128 of the tree's first 16 bytes; the child is selected by the compare's own mask (v_cndmask),
// which replaces an address computation + an LDS node read per segment (VALU-neutral, 3 LDS instructions fewer per tree).
template <int R>
__device__ __forceinline__ void step_sel_rk(uint32_t* j, uint32_t root, const uint32_t* r, uint32_t lo, uint32_t hi, uint32_t* nxt) {
  if constexpr (R == 1) {
    uint64_t c0;
    asm("v_cmp_le_u32_sdwa %[c0], %[n], %[r0] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %[x0], %[lo], %[hi], %[c0]\n\t"  /* before the addc: its carry-out overwrites the mask */
        "v_addc_co_u32 %[j0], %[c0], %[j0], %[j0], %[c0]"
        : [j0] "+v"(j[0]), [x0] "=&v"(nxt[0]), [c0] "=&s"(c0)
        : [n] "v"(root), [r0] "v"(r[0]), [lo] "v"(lo), [hi] "v"(hi));
  } else if constexpr (R == 2) {
    uint64_t c0, c1;
    asm("v_cmp_le_u32_sdwa %[c0], %[n], %[r0] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_cmp_le_u32_sdwa %[c1], %[n], %[r1] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32 %[x0], %[lo], %[hi], %[c0]\n\t"
        "v_cndmask_b32 %[x1], %[lo], %[hi], %[c1]\n\t"
        "v_addc_co_u32 %[j0], %[c0], %[j0], %[j0], %[c0]\n\t"
        "v_addc_co_u32 %[j1], %[c1], %[j1], %[j1], %[c1]"
        : [j0] "+v"(j[0]), [j1] "+v"(j[1]), [x0] "=&v"(nxt[0]), [x1] "=&v"(nxt[1]), [c0] "=&s"(c0), [c1] "=&s"(c1)
        : [n] "v"(root), [r0] "v"(r[0]), [r1] "v"(r[1]), [lo] "v"(lo), [hi] "v"(hi));
  } else if constexpr (R == 3) {
    uint64_t c0, c1, c2;
    asm("v_cmp_le_u32_sdwa %[c0], %[n], %[r0] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_cmp_le_u32_sdwa %[c1], %[n], %[r1] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_cmp_le_u32_sdwa %[c2], %[n], %[r2] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_cndmask_b32 %[x0], %[lo], %[hi], %[c0]\n\t"
        "v_cndmask_b32 %[x1], %[lo], %[hi], %[c1]\n\t"
        "v_cndmask_b32 %[x2], %[lo], %[hi], %[c2]\n\t"
        "v_addc_co_u32 %[j0], %[c0], %[j0], %[j0], %[c0]\n\t"
        "v_addc_co_u32 %[j1], %[c1], %[j1], %[j1], %[c1]\n\t"
        "v_addc_co_u32 %[j2], %[c2], %[j2], %[j2], %[c2]"
        : [j0] "+v"(j[0]), [j1] "+v"(j[1]), [j2] "+v"(j[2]), [x0] "=&v"(nxt[0]), [x1] "=&v"(nxt[1]), [x2] "=&v"(nxt[2]),
          [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2)
        : [n] "v"(root), [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [lo] "v"(lo), [hi] "v"(hi));
  } else if constexpr (R == 4) {
    step_sel_rk<2>(j, root, r, lo, hi, nxt);
    step_sel_rk<2>(j + 2, root, r + 2, lo, hi, nxt + 2);
  } else if constexpr (R == 5) {
    step_sel_rk<3>(j, root, r, lo, hi, nxt);
    step_sel_rk<2>(j + 3, root, r + 3, lo, hi, nxt + 3);
  } else {
    static_assert(R == 6, "segments per strip");
    step_sel_rk<3>(j, root, r, lo, hi, nxt);
    step_sel_rk<3>(j + 3, root, r + 3, lo, hi, nxt + 3);
  }
}

// Level 0 for a fresh walk: j = 2 + (r >= rank field) straight from inline constants (v_addc_co_u32 VOP3: 1 + 1 + carry) —
// no "j = 1" initialisation per tree and segment — and the level-1 node picked by the same mask.
template <int R>
__device__ __forceinline__ void step_first_rk(uint32_t* j, uint32_t root, const uint32_t* r, uint32_t lo, uint32_t hi, uint32_t* nxt) {
  if constexpr (R == 1) {
    uint64_t c0;
    asm("v_cmp_le_u32_sdwa %[c0], %[n], %[r0] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %[x0], %[lo], %[hi], %[c0]\n\t"
        "v_addc_co_u32_e64 %[j0], %[c0], 1, 1, %[c0]"
        : [j0] "=&v"(j[0]), [x0] "=&v"(nxt[0]), [c0] "=&s"(c0)
        : [n] "v"(root), [r0] "v"(r[0]), [lo] "v"(lo), [hi] "v"(hi));
  } else if constexpr (R == 2) {
    uint64_t c0, c1;
    asm("v_cmp_le_u32_sdwa %[c0], %[n], %[r0] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_cmp_le_u32_sdwa %[c1], %[n], %[r1] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32 %[x0], %[lo], %[hi], %[c0]\n\t"
        "v_cndmask_b32 %[x1], %[lo], %[hi], %[c1]\n\t"
        "v_addc_co_u32_e64 %[j0], %[c0], 1, 1, %[c0]\n\t"
        "v_addc_co_u32_e64 %[j1], %[c1], 1, 1, %[c1]"
        : [j0] "=&v"(j[0]), [j1] "=&v"(j[1]), [x0] "=&v"(nxt[0]), [x1] "=&v"(nxt[1]), [c0] "=&s"(c0), [c1] "=&s"(c1)
        : [n] "v"(root), [r0] "v"(r[0]), [r1] "v"(r[1]), [lo] "v"(lo), [hi] "v"(hi));
  } else if constexpr (R == 3) {
    uint64_t c0, c1, c2;
    asm("v_cmp_le_u32_sdwa %[c0], %[n], %[r0] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_cmp_le_u32_sdwa %[c1], %[n], %[r1] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_cmp_le_u32_sdwa %[c2], %[n], %[r2] src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_cndmask_b32 %[x0], %[lo], %[hi], %[c0]\n\t"
        "v_cndmask_b32 %[x1], %[lo], %[hi], %[c1]\n\t"
        "v_cndmask_b32 %[x2], %[lo], %[hi], %[c2]\n\t"
        "v_addc_co_u32_e64 %[j0], %[c0], 1, 1, %[c0]\n\t"
        "v_addc_co_u32_e64 %[j1], %[c1], 1, 1, %[c1]\n\t"
        "v_addc_co_u32_e64 %[j2], %[c2], 1, 1, %[c2]"
        : [j0] "=&v"(j[0]), [j1] "=&v"(j[1]), [j2] "=&v"(j[2]), [x0] "=&v"(nxt[0]), [x1] "=&v"(nxt[1]), [x2] "=&v"(nxt[2]),
          [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2)
        : [n] "v"(root), [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [lo] "v"(lo), [hi] "v"(hi));
  } else if constexpr (R == 4) {
    step_first_rk<2>(j, root, r, lo, hi, nxt);
    step_first_rk<2>(j + 2, root, r + 2, lo, hi, nxt + 2);
  } else if constexpr (R == 5) {
    step_first_rk<3>(j, root, r, lo, hi, nxt);
    step_first_rk<2>(j + 3, root, r + 3, lo, hi, nxt + 3);
  } else {
    static_assert(R == 6, "segments per strip");
    step_first_rk<3>(j, root, r, lo, hi, nxt);
    step_first_rk<3>(j + 3, root, r + 3, lo, hi, nxt + 3);
  }
}

// TWO trees side by side for the R segments of a lane (D >= 2): 2R independent chains keep more LDS reads in flight per
// wave, so the LDS pipe and the VALU overlap better; the leaves are still added in tree order (tb0 before tb1).
template <int D, int R>
__device__ __forceinline__ void walk_rk_pair(const uint8_t* tb0, const uint8_t* tb1, const uint8_t* rb, float* psum) {
  static_assert(D >= 2, "pair walk needs the 16-byte tree top");
  uint32_t j[2 * R], nd[2 * R], r[2 * R];
  const uint4 top0 = *reinterpret_cast<const uint4*>(tb0);  // words 0..3: (unused, root, left child, right child)
  const uint4 top1 = *reinterpret_cast<const uint4*>(tb1);
#pragma unroll
  for (int k = 0; k < R; ++k) r[k] = *reinterpret_cast<const uint16_t*>(rb + k * (WS * 2) + (top0.y & 0xffffu));
#pragma unroll
  for (int k = 0; k < R; ++k) r[R + k] = *reinterpret_cast<const uint16_t*>(rb + k * (WS * 2) + (top1.y & 0xffffu));
  step_first_rk<R>(j, top0.y, r, top0.z, top0.w, nd);
  step_first_rk<R>(j + R, top1.y, r + R, top1.z, top1.w, nd + R);
#pragma unroll
  for (int k = 0; k < 2 * R; ++k) r[k] = *reinterpret_cast<const uint16_t*>(rb + (k % R) * (WS * 2) + (nd[k] & 0xffffu));
  step_rk<R>(j, nd, r);
  step_rk<R>(j + R, nd + R, r + R);
#pragma unroll
  for (int d = 2; d < D; ++d) {
#pragma unroll
    for (int k = 0; k < R; ++k) nd[k] = reinterpret_cast<const uint32_t*>(tb0)[j[k]];
#pragma unroll
    for (int k = 0; k < R; ++k) nd[R + k] = reinterpret_cast<const uint32_t*>(tb1)[j[R + k]];
#pragma unroll
    for (int k = 0; k < 2 * R; ++k) r[k] = *reinterpret_cast<const uint16_t*>(rb + (k % R) * (WS * 2) + (nd[k] & 0xffffu));
    step_rk<R>(j, nd, r);
    step_rk<R>(j + R, nd + R, r + R);
  }
  float l0[R], l1[R];
#pragma unroll
  for (int k = 0; k < R; ++k) l0[k] = reinterpret_cast<const float*>(tb0)[j[k]];
#pragma unroll
  for (int k = 0; k < R; ++k) l1[k] = reinterpret_cast<const float*>(tb1)[j[R + k]];
#pragma unroll
  for (int k = 0; k < R; ++k) psum[k] += l0[k];
#pragma unroll
  for (int k = 0; k < R; ++k) psum[k] += l1[k];
}

// D levels for the R segments of a lane: tb = the tree's 2^D node words + 2^D leaves (LDS), rb = the lane's origin in
// its strip (LDS); segment k sits 64 windows = 128 bytes further
template <int D, int R>
__device__ __forceinline__ void walk_rk(const uint8_t* tb, const uint8_t* rb, float* psum) {
  uint32_t j[R];
  uint32_t nd[R], r[R];
  if constexpr (D < 2) {
#pragma unroll
    for (int k = 0; k < R; ++k) j[k] = 1;
  }
  if constexpr (D >= 2) {
    const uint4 top = *reinterpret_cast<const uint4*>(tb);  // words 0..3: (unused, root, left child, right child)
#pragma unroll
    for (int k = 0; k < R; ++k) r[k] = *reinterpret_cast<const uint16_t*>(rb + k * (WS * 2) + (top.y & 0xffffu));
    step_first_rk<R>(j, top.y, r, top.z, top.w, nd);
#pragma unroll
    for (int k = 0; k < R; ++k) r[k] = *reinterpret_cast<const uint16_t*>(rb + k * (WS * 2) + (nd[k] & 0xffffu));
    step_rk<R>(j, nd, r);
  }
#pragma unroll
  for (int d = (D >= 2 ? 2 : 0); d < D; ++d) {
#pragma unroll
    for (int k = 0; k < R; ++k) nd[k] = reinterpret_cast<const uint32_t*>(tb)[j[k]];
#pragma unroll
    for (int k = 0; k < R; ++k) r[k] = *reinterpret_cast<const uint16_t*>(rb + k * (WS * 2) + (nd[k] & 0xffffu));
    step_rk<R>(j, nd, r);
  }
#pragma unroll
  for (int k = 0; k < R; ++k) psum[k] += reinterpret_cast<const float*>(tb)[j[k]];
}

// ---- pointer nodes (PTR variant): the walk carries the LDS ADDRESS of its node instead of a heap index -----------------------
// A node is 8 bytes {w0 = rank field << 16 | strip offset, w1 = address of the left child | address of the right child << 16}
// (ds_read_b64 occupies the LDS array for the same 2 cycles as ds_read_b32), so a level is rank address (v_add_u32_sdwa),
// compare (v_cmp_le_u32_sdwa -> vcc) and v_cndmask_b32_sdwa picking a half of w1: 3 VALU instead of 4 — no node address, no
// heap index — and the last level's "children" are the leaves' own addresses (no leaf address either).
// (the low 32 bits of a generic pointer into the LDS are its LDS address; reading through an address_space(3) pointer made from
// the integer gives `ds_read vdst, p` with nothing added)
#if defined(__HIP_DEVICE_COMPILE__)
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(3))) T* lds_at(uint32_t a) {
  return (const __attribute__((address_space(3))) T*)(uintptr_t)a;
}
#else
template <typename T>
__device__ const T* lds_at(uint32_t) { return nullptr; }  // host pass: never called
#endif
__device__ __forceinline__ uint32_t step_ptr(uint32_t w0, uint32_t w1, uint32_t r) {
  uint32_t p;
  asm("v_cmp_le_u32_sdwa vcc, %[n], %[r] src0_sel:WORD_1 src1_sel:DWORD\n\t"
      "v_cndmask_b32_sdwa %[p], %[w], %[w], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
      : [p] "=v"(p)
      : [n] "v"(w0), [r] "v"(r), [w] "v"(w1)
      : "vcc");
  return p;
}
// level 0: the level-1 node {w0, w1} is one of two wave-uniform candidates, picked by the root's compare
__device__ __forceinline__ void first_ptr(uint32_t root, uint32_t r, uint32_t l0, uint32_t r0, uint32_t l1, uint32_t r1, uint32_t& w0,
                                          uint32_t& w1) {
  asm("v_cmp_le_u32_sdwa vcc, %[n], %[r] src0_sel:WORD_1 src1_sel:DWORD\n\t"
      "v_cndmask_b32 %[x], %[l0], %[r0], vcc\n\t"
      "v_cndmask_b32 %[y], %[l1], %[r1], vcc"
      : [x] "=&v"(w0), [y] "=&v"(w1)
      : [n] "v"(root), [r] "v"(r), [l0] "v"(l0), [r0] "v"(r0), [l1] "v"(l1), [r1] "v"(r1)
      : "vcc");
}

// TWO trees side by side for the R segments of a lane (node addresses = LDS addresses).
template <int D, int R>
__device__ __forceinline__ void walk_rp_pair(const uint8_t* tb0, const uint8_t* tb1, const uint8_t* rb, float* psum) {
  static_assert(D >= 2, "the 16-byte tree top holds levels 0 and 1");
  uint32_t w0[2 * R], w1[2 * R], r[2 * R], p[2 * R];
  const uint4 top0 = *reinterpret_cast<const uint4*>(tb0);  // {w0 of node 2, w0 of node 3, w0 of the root, w1 of node 2}
  const uint4 top1 = *reinterpret_cast<const uint4*>(tb1);
  constexpr uint32_t SIB = (D == 2 ? 8u : 16u) * 0x10001u;  // node 3's children sit right behind node 2's
  const uint32_t k0 = top0.w + SIB, k1 = top1.w + SIB;
#pragma unroll
  for (int k = 0; k < R; ++k) r[k] = *reinterpret_cast<const uint16_t*>(rb + k * (WS * 2) + (top0.z & 0xffffu));
#pragma unroll
  for (int k = 0; k < R; ++k) r[R + k] = *reinterpret_cast<const uint16_t*>(rb + k * (WS * 2) + (top1.z & 0xffffu));
#pragma unroll
  for (int k = 0; k < R; ++k) first_ptr(top0.z, r[k], top0.x, top0.y, top0.w, k0, w0[k], w1[k]);
#pragma unroll
  for (int k = 0; k < R; ++k) first_ptr(top1.z, r[R + k], top1.x, top1.y, top1.w, k1, w0[R + k], w1[R + k]);
#pragma unroll
  for (int k = 0; k < 2 * R; ++k) r[k] = *reinterpret_cast<const uint16_t*>(rb + (k % R) * (WS * 2) + (w0[k] & 0xffffu));
#pragma unroll
  for (int k = 0; k < 2 * R; ++k) p[k] = step_ptr(w0[k], w1[k], r[k]);
#pragma unroll
  for (int d = 2; d < D; ++d) {
#pragma unroll
    for (int k = 0; k < 2 * R; ++k) {
      const uint2 nd = *lds_at<uint2>(p[k]);
      w0[k] = nd.x;
      w1[k] = nd.y;
    }
#pragma unroll
    for (int k = 0; k < 2 * R; ++k) r[k] = *reinterpret_cast<const uint16_t*>(rb + (k % R) * (WS * 2) + (w0[k] & 0xffffu));
#pragma unroll
    for (int k = 0; k < 2 * R; ++k) p[k] = step_ptr(w0[k], w1[k], r[k]);
  }
  float lf[2 * R];
#pragma unroll
  for (int k = 0; k < 2 * R; ++k) lf[k] = *lds_at<float>(p[k]);
#pragma unroll
  for (int k = 0; k < R; ++k) psum[k] += lf[k];  // tree order: tb0 before tb1
#pragma unroll
  for (int k = 0; k < R; ++k) psum[k] += lf[R + k];
}

// one tree (the odd tail of a group)
template <int D>
__device__ __forceinline__ float walk_rp_one(const uint8_t* tb, const uint8_t* rb) {
  const uint32_t root = reinterpret_cast<const uint32_t*>(tb)[2];
  uint32_t r = *reinterpret_cast<const uint16_t*>(rb + (root & 0xffffu));
  uint32_t p = (uint32_t)(uintptr_t)tb + ((r < (root >> 16)) ? 16u : 24u);  // nodes 2 and 3 keep their own slots
#pragma unroll
  for (int d = 1; d < D; ++d) {
    const uint2 nd = *lds_at<uint2>(p);
    r = *reinterpret_cast<const uint16_t*>(rb + (nd.x & 0xffffu));
    p = (r < (nd.x >> 16)) ? (nd.y & 0xffffu) : (nd.y >> 16);
  }
  return *lds_at<float>(p);
}

__device__ __forceinline__ float walk_rk_rt(const uint8_t* tb, const uint8_t* rb, int D) {
  uint32_t j = 1;
  for (int d = 0; d < D; ++d) {
    const uint32_t nd = reinterpret_cast<const uint32_t*>(tb)[j];
    const uint32_t r = *reinterpret_cast<const uint16_t*>(rb + (nd & 0xffffu));
    j = 2 * j + ((r < (nd >> 16)) ? 0u : 1u);
  }
  return reinterpret_cast<const float*>(tb)[j];
}

// One wave = one haplotype x RPL consecutive 64-window segments; NWAVE haplotypes per block.  DT = depth (0 = runtime).
template <int RPL, int NWAVE, int DT, bool PAIR = false, bool PTR = false>
__global__ __launch_bounds__(NWAVE * 64) void k_smooth_xgb_rk(SmoothXGBLaunch L) {
  static_assert(!PTR || (PAIR && DT >= 2), "pointer nodes: pair walks of a compile-time depth");
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int THREADS = NWAVE * 64;
  const int A = L.A, W = L.W, S = L.S, pad = (S + 1) / 2;
  const int D = DT ? DT : L.d.D;
  const int tree_bytes = PTR ? L.d.rp_tree_bytes : L.d.rk_tree_bytes;
  const int32_t* const group_tree0 = PTR ? L.d.rp_group_tree0 : L.d.rk_group_tree0;
  const int32_t* const group_class = PTR ? L.d.rp_group_class : L.d.rk_group_class;
  const uint8_t* const packed = PTR ? L.d.rp_packed : L.d.rk_packed;
  const int stride = L.d.rk_stride;              // halfwords per class row (>= RPL*64 + S - 1)
  const int strip_w = RPL * WS + S - 1;          // padded windows held per haplotype
  const int strip_bytes = A * stride * 2;
  const int buf_bytes = (PTR ? L.d.rp_max_group : L.d.rk_max_group) * tree_bytes;  // multiple of 16
  uint8_t* strip = lds;                          // [NWAVE][A][stride] u16
  uint8_t* tbuf0 = lds + (((size_t)NWAVE * strip_bytes + 15) & ~(size_t)15);
  uint8_t* tbuf1 = tbuf0 + buf_bytes;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t h0 = (int64_t)blockIdx.y * NWAVE;
  const int w0 = blockIdx.x * (RPL * WS);

  // ---- stage the reflected base-probability strips as ranks -------------------------------------------------
  {
    constexpr int NV = 4;
    const int per_h = strip_w * A;
    const int total = NWAVE * per_h;
    for (int e0 = tid; e0 < total; e0 += NV * THREADS) {
      float p[NV];
      uint32_t r[NV];
      int dst[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int e = min(e0 + i * THREADS, total - 1);  // clamped: loads stay unconditional
        const int hl = e / per_h, rr = e - hl * per_h;
        const int q = rr / A, a = rr - q * A;
        const int64_t n = min(h0 + hl, L.N - 1);
        const int j = min(w0 + q, W + 2 * pad - 1);
        const size_t idx = ((size_t)n * W + slide_src(j, W, pad)) * A + a;
        p[i] = L.b_is_f64 ? (float)reinterpret_cast<const double*>(L.B)[idx] : reinterpret_cast<const float*>(L.B)[idx];
        dst[i] = (hl * A + a) * stride + q;
      }
      ranks<NV>(L.d.rk_thr, L.d.rk_lut, L.d.rk_K, L.d.rk_steps, p, r);
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (e0 + i * THREADS < total) reinterpret_cast<uint16_t*>(strip)[dst[i]] = (uint16_t)r[i];
    }
  }

  const int64_t n = h0 + wave;
  const uint8_t* rowbase[RPL];
  bool valid[RPL];
  size_t orow[RPL];
  const size_t NW = (size_t)L.N * W;  // class stride of the margin scratch
#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    rowbase[k] = strip + (size_t)wave * strip_bytes + (size_t)(k * WS + lane) * 2;
    const int w = w0 + k * WS + lane;
    valid[k] = (n < L.N) && (w < W);
    orow[k] = (size_t)(valid[k] ? n : 0) * W + (valid[k] ? w : 0);  // row index (n, w)
  }

  // ---- tree groups through the double-buffered LDS window ----
  const int ng = PTR ? L.d.rp_n_groups : L.d.rk_n_groups;
  constexpr int MAXV = (8192 / 16 + THREADS - 1) / THREADS;  // 16-byte staging pieces per thread (buf <= 8 KB)
  uint4 stg[MAXV];
  const int nv = (buf_bytes / 16 + THREADS - 1) / THREADS;
  // unconditional clamped loads: no branch around a load, so hipcc keeps its waits counted
#define GNX_G_LOAD(g)                                                                               \
  {                                                                                                 \
    const int t0_ = group_tree0[g], t1_ = group_tree0[(g) + 1];                                     \
    const int last_ = (t1_ - t0_) * tree_bytes / 16 - 1;                                            \
    const uint4* src_ = reinterpret_cast<const uint4*>(packed + (size_t)t0_ * tree_bytes);          \
    _Pragma("unroll") for (int v = 0; v < MAXV; ++v) if (v < nv) stg[v] = src_[min(v * THREADS + tid, last_)]; \
  }
#define GNX_G_STORE(dst)                                                                            \
  {                                                                                                 \
    _Pragma("unroll") for (int v = 0; v < MAXV; ++v) {                                              \
      const int e_ = v * THREADS + tid;                                                             \
      if constexpr (PTR) { /* child addresses: relative to the group -> relative to the block's LDS origin */ \
        const int pc_ = (e_ * 16 % tree_bytes) >> 4;                                                \
        const uint32_t add_ = pc_ < (1 << (DT - 1)) ? (uint32_t)(uintptr_t)(dst) * 0x10001u : 0u;   \
        stg[v].w += add_;                                                                           \
        stg[v].y += pc_ ? add_ : 0u;                                                                \
      }                                                                                             \
      if (v < nv && e_ * 16 < buf_bytes) *reinterpret_cast<uint4*>((dst) + (size_t)e_ * 16) = stg[v]; \
    }                                                                                               \
  }

  float psum[RPL];
#pragma unroll
  for (int k = 0; k < RPL; ++k) psum[k] = 0.f;

  GNX_G_LOAD(0);
  GNX_G_STORE(tbuf0);
  __syncthreads();

  int cur_class = group_class[0];
  for (int g = 0; g < ng; ++g) {
    uint8_t* cur = (g & 1) ? tbuf1 : tbuf0;
    uint8_t* nxt = (g & 1) ? tbuf0 : tbuf1;
    const int gn = min(g + 1, ng - 1);  // clamped: the last iteration re-fetches its own group
    GNX_G_LOAD(gn);

    const int cls = group_class[g];
    if (cls != cur_class) {  // class finished: park its margin (base_score + psum), class-major = full cache lines
#pragma unroll
      for (int k = 0; k < RPL; ++k) {
        if (valid[k]) L.marg[(size_t)cur_class * NW + orow[k]] = L.d.base_score + psum[k];
        psum[k] = 0.f;
      }
      cur_class = cls;
    }
    const int nt = group_tree0[g + 1] - group_tree0[g];
    int t = 0;
    if constexpr (DT >= 2 && PAIR) {
      for (; t + 1 < nt; t += 2) {
        const uint8_t* tb = cur + (size_t)t * tree_bytes;
        if constexpr (PTR) walk_rp_pair<DT, RPL>(tb, tb + tree_bytes, rowbase[0], psum);
        else walk_rk_pair<DT, RPL>(tb, tb + tree_bytes, rowbase[0], psum);
      }
    }
    for (; t < nt; ++t) {
      const uint8_t* tb = cur + (size_t)t * tree_bytes;
      if constexpr (PTR) {
#pragma unroll
        for (int k = 0; k < RPL; ++k) psum[k] += walk_rp_one<DT>(tb, rowbase[k]);
      } else if constexpr (DT > 0) walk_rk<DT, RPL>(tb, rowbase[0], psum);
      else {
#pragma unroll
        for (int k = 0; k < RPL; ++k) psum[k] += walk_rk_rt(tb, rowbase[k], D);
      }
    }
    GNX_G_STORE(nxt);
    __syncthreads();
  }
#undef GNX_G_LOAD
#undef GNX_G_STORE
#pragma unroll
  for (int k = 0; k < RPL; ++k)
    if (valid[k]) L.marg[(size_t)cur_class * NW + orow[k]] = L.d.base_score + psum[k];

  // ---- softmax (xgboost common/math.h Softmax) + argmax, per row, by the lane that wrote the margins ----
#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    if (!valid[k]) continue;
    const float* mg = L.marg + orow[k];
    float* o = L.proba + orow[k] * A;
    float wmax = mg[0];
    for (int a = 1; a < A; ++a) wmax = fmaxf(mg[(size_t)a * NW], wmax);
    double wsum = 0.0;
    for (int a = 0; a < A; ++a) {
      const float e = gnx_softmax_exp(mg[(size_t)a * NW] - wmax);
      o[a] = e;
      wsum += (double)e;
    }
    const float fs = (float)wsum;
    int best = 0;
    float bv = -1.f;
    for (int a = 0; a < A; ++a) {
      const float p = o[a] / fs;
      o[a] = p;
      if (L.proba64) L.proba64[orow[k] * A + a] = (double)p;
      if (p > bv) { bv = p; best = a; }
    }
    if (L.labels) L.labels[orow[k]] = best;
  }
}

template <int NWAVE>
size_t lds_bytes(const SmoothXGBDev& d, int A, bool ptr = false) {
  const size_t strip = (size_t)NWAVE * A * d.rk_stride * 2;
  return ((strip + 15) & ~(size_t)15) + 2 * (ptr ? (size_t)d.rp_max_group * d.rp_tree_bytes : (size_t)d.rk_max_group * d.rk_tree_bytes);
}

template <int RPL, int NWAVE>
hipError_t launch(const SmoothXGBLaunch& L, bool pair, int lds_pad, hipStream_t s) {
  const dim3 grid((unsigned)((L.W + RPL * WS - 1) / (RPL * WS)), (unsigned)((L.N + NWAVE - 1) / NWAVE));
  // pointer nodes: depth 4, pair walks, and every node address must fit 16 bits
  const bool ptr = L.d.impl == 3 && L.d.rp_packed && L.d.D == 4 && pair && RPL <= 3 && lds_bytes<NWAVE>(L.d, L.A, true) <= 65536;
  size_t lds = lds_bytes<NWAVE>(L.d, L.A, ptr);
  if (lds > (size_t)160 * 1024) return hipErrorInvalidValue;
  lds = std::min(lds + (size_t)std::max(lds_pad, 0), (size_t)160 * 1024);
  if (ptr) {
    if constexpr (RPL <= 3) {
      GNX_LDS_OPTIN(lds, k_smooth_xgb_rk<RPL, NWAVE, 4, true, true>);
      hipLaunchKernelGGL((k_smooth_xgb_rk<RPL, NWAVE, 4, true, true>), grid, dim3(NWAVE * 64), lds, s, L);
    }
  } else if (L.d.D == 4 && pair && RPL <= 3) {
    if constexpr (RPL <= 3) {
      GNX_LDS_OPTIN(lds, k_smooth_xgb_rk<RPL, NWAVE, 4, true>);
      hipLaunchKernelGGL((k_smooth_xgb_rk<RPL, NWAVE, 4, true>), grid, dim3(NWAVE * 64), lds, s, L);
    }
  } else if (L.d.D == 4) {
    GNX_LDS_OPTIN(lds, k_smooth_xgb_rk<RPL, NWAVE, 4>);
    hipLaunchKernelGGL((k_smooth_xgb_rk<RPL, NWAVE, 4>), grid, dim3(NWAVE * 64), lds, s, L);
  } else {
    GNX_LDS_OPTIN(lds, k_smooth_xgb_rk<RPL, NWAVE, 0>);
    hipLaunchKernelGGL((k_smooth_xgb_rk<RPL, NWAVE, 0>), grid, dim3(NWAVE * 64), lds, s, L);
  }
  return hipGetLastError();
}

}  // namespace

hipError_t gnx_launch_smooth_xgb_rk(const SmoothXGBLaunch& L, const gnx_tune& tune, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  const int rpl = L.d.rk_rpl;  // fixed at model load: the node offsets encode the strip stride
  const bool pair = tune.sm_pair != 0;  // two trees side by side per lane
  int nw = tune.sm_nw;
  if (nw != 2 && nw != 4 && nw != 8) {
    // as many waves per CU as the LDS allows: 8-wave blocks unless 4-wave blocks pack the 160 KB better
    const size_t l8 = lds_bytes<8>(L.d, L.A), l4 = lds_bytes<4>(L.d, L.A);
    const size_t w8 = l8 <= (size_t)160 * 1024 ? ((size_t)160 * 1024 / l8) * 8 : 0, w4 = l4 <= (size_t)160 * 1024 ? ((size_t)160 * 1024 / l4) * 4 : 0;
    nw = (w8 > w4) ? 8 : (w4 > 0 ? 4 : 2);  // equal wave counts: the smaller block (A = 12 at chr1: 16 waves per CU either way, 4-wave blocks 3 % faster)
    if (L.N < 8) nw = L.N < 3 ? 2 : 4;
  }
#define GNX_SM_CASE(R_) \
  if (rpl == R_) return nw == 8 ? launch<R_, 8>(L, pair, tune.sm_lds_pad, s) : (nw == 4 ? launch<R_, 4>(L, pair, tune.sm_lds_pad, s) : launch<R_, 2>(L, pair, tune.sm_lds_pad, s));
  GNX_SM_CASE(1) GNX_SM_CASE(2) GNX_SM_CASE(3) GNX_SM_CASE(4) GNX_SM_CASE(5) GNX_SM_CASE(6)
#undef GNX_SM_CASE
  return hipErrorInvalidValue;
}
