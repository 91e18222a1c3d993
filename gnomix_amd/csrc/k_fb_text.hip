// k_fb_text.hip — the text of <prefix>.fb produced on the GPU (include/gnomix_io.h: gnx_write_fb_dev).
//
// Replaces the float-to-text part of write_fb (reference src/postprocess.py:100-126: pandas DataFrame.to_csv of float32 columns, i.e.
// numpy's shortest round-trip text of every probability).  The host writer (gnx_io.cpp: gnx_write_fb) formats on every allowed
// core and funnels 0.8 MB blocks through one writing thread: 75-95 ms for chr22 x 10 000 haplotypes (305 MB of text), flat from 12
// threads up — the bound is the one thread that copies blocks other cores produced into the page cache.  Here the 26 M numbers
// become text in HBM (two launches), the file's body comes back as ONE page-locked buffer and is written with one write():
//   k_fb_len    one thread per value: the length of "\t" + text (NaN: just the tab, as pandas' na_rep ""), and the line totals;
//   k_fb_emit   one block per line (window): prefix text, then the values in chunks of 1024 — length scan inside the block,
//               every thread prints its value at its offset — and the newline.
// The digits are Schubfach's (the binary32 instance in gnx_io.cpp, same constants, same steps, in 64-bit integer arithmetic:
// __umul64hi is the 128-bit product's upper half), the layout numpy's: positional for 1e-4 <= |x| < 1e16, else d.ddde+XX.
// Byte-identical to the host writer (tests/test_gpu_vcf.py), which is byte-identical to the reference's writer (golden G6).
// MEASURED (chr22 x 10 000 haplotypes, tmpfs): 0.078-0.082 s against 0.080 s for the host writer — no gain.  The premise above was
// wrong about the bound: ONE thread putting 305 MB into a FRESH file spends 0.052-0.056 s in the kernel's page cache whoever produced
// the bytes (scripts/dev/tmpfs_write_probe.py: 5.4-5.9 GB/s; 7-9 GB/s into an existing file), and concurrent writers of one tmpfs
// file are slower still (scripts/dev/io_probe.cpp).  Kept as what `write_fb(..., ctx=)` / GNX_FB_DEV=1 select (it leaves the host's
// cores free); the command line's default stays the host writer.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gnx_internal.h"

namespace {

__constant__ uint64_t kG1[77] = {
    0x40e7599625a1fe7aull, 0x51212ffbaf0a7e18ull, 0x65697bfa9acd1d9full, 0x7ec3daf941806506ull,
    0x4f3a68dbc8f03f24ull, 0x63090312bb2c4eedull, 0x7bcb43d769f762a8ull, 0x4d5f0a66a23a9da9ull,
    0x60b6cd004ac94513ull, 0x78e480405d7b9658ull, 0x4b8ed0283a6d3df7ull, 0x5e72843249088d75ull,
    0x760f253edb4ab0d2ull, 0x49c97747490eae83ull, 0x5c3bd5191b525a24ull, 0x734aca5f6226f0adull,
    0x480ebe7b9d58566cull, 0x5a126e1a84ae6c07ull, 0x709709a125da0709ull, 0x465e6604b7a84465ull,
    0x57f5ff85e592557full, 0x6df37f675ef6eadfull, 0x44b82fa09b5a52cbull, 0x55e63b88c230e77eull,
    0x6b5fca6af2bd215eull, 0x431bde82d7b634daull, 0x53e2d6238da3c211ull, 0x68db8bac710cb295ull,
    0x4189374bc6a7ef9dull, 0x51eb851eb851eb85ull, 0x6666666666666666ull, 0x4000000000000000ull,
    0x5000000000000000ull, 0x6400000000000000ull, 0x7d00000000000000ull, 0x4e20000000000000ull,
    0x61a8000000000000ull, 0x7a12000000000000ull, 0x4c4b400000000000ull, 0x5f5e100000000000ull,
    0x7735940000000000ull, 0x4a817c8000000000ull, 0x5d21dba000000000ull, 0x746a528800000000ull,
    0x48c2739500000000ull, 0x5af3107a40000000ull, 0x71afd498d0000000ull, 0x470de4df82000000ull,
    0x58d15e1762800000ull, 0x6f05b59d3b200000ull, 0x4563918244f40000ull, 0x56bc75e2d6310000ull,
    0x6c6b935b8bbd4000ull, 0x43c33c1937564800ull, 0x54b40b1f852bda00ull, 0x69e10de76676d080ull,
    0x422ca8b0a00a4250ull, 0x52b7d2dcc80cd2e4ull, 0x6765c793fa10079dull, 0x409f9cbc7c4a04c2ull,
    0x50c783eb9b5c85f2ull, 0x64f964e68233a76full, 0x7e37be2022c0914bull, 0x4ee2d6d415b85aceull,
    0x629b8c891b267182ull, 0x7b426fab61f00de3ull, 0x4d0985cb1d3608aeull, 0x604be73de4838ad9ull,
    0x785ee10d5da46d90ull, 0x4b3b4ca85a86c47aull, 0x5e0a1fd271287598ull, 0x758ca7c70d7292feull,
    0x4977e8dc68679bdfull, 0x5bd5e313828182d6ull, 0x72cb5bd86321e38cull, 0x47bf19673df52e37ull,
    0x59aedfc10d7279c5ull,
};

__device__ __forceinline__ int64_t sf_flog10pow2(int64_t e) { return (e * 661971961083LL) >> 41; }
__device__ __forceinline__ int64_t sf_flog10_three_quarters_pow2(int64_t e) { return (e * 661971961083LL - 274743187321LL) >> 41; }
__device__ __forceinline__ int64_t sf_flog2pow10(int64_t e) { return (e * 913124641741LL) >> 38; }
__device__ __forceinline__ uint32_t sf_rop(uint64_t g, uint64_t cp) {
  const uint64_t x1 = __umul64hi(g, cp);
  return (uint32_t)((x1 >> 31) | (((x1 & 0xffffffffull) + 0xffffffffull) >> 32));
}

// positive, finite, non-zero float -> digits f (no trailing zeros) and exponent e: the value prints as f x 10^e (gnx_io.cpp: f32_shortest)
__device__ __forceinline__ void f32_shortest(uint32_t bits, uint32_t& f_out, int& e_out) {
  const uint32_t t = bits & 0x7fffffu, bq = bits >> 23;
  const int64_t q = bq ? (int64_t)bq - 150 : -149;
  const uint64_t c = bq ? (0x800000u | t) : t;
  const uint32_t out = (uint32_t)(c & 1);
  const uint64_t cb = c << 2, cbr = cb + 2;
  uint64_t cbl;
  int64_t k;
  if (c != 0x800000u || q == -149) {
    cbl = cb - 2;
    k = sf_flog10pow2(q);
  } else {
    cbl = cb - 1;
    k = sf_flog10_three_quarters_pow2(q);
  }
  const int h = (int)(q + sf_flog2pow10(-k) + 33);
  const uint64_t g = kG1[-k + 31] + 1;
  const uint32_t vb = sf_rop(g, cb << h), vbl = sf_rop(g, cbl << h), vbr = sf_rop(g, cbr << h);
  const uint32_t s = vb >> 2;
  uint32_t f = 0;
  bool done = false;
  if (s >= 100) {
    const uint32_t sp10 = 10 * (s / 10), tp10 = sp10 + 10;
    const bool upin = vbl + out <= (sp10 << 2), wpin = (tp10 << 2) + out <= vbr;
    if (upin != wpin) {
      f = upin ? sp10 : tp10;
      done = true;
    }
  }
  if (!done) {
    const uint32_t t1 = s + 1;
    const bool uin = vbl + out <= (s << 2), win = (t1 << 2) + out <= vbr;
    if (uin != win) f = uin ? s : t1;
    else {
      const int32_t cmp = (int32_t)vb - (int32_t)((s + t1) << 1);
      f = (cmp < 0 || (cmp == 0 && (s & 1) == 0)) ? s : t1;
    }
  }
  int e = (int)k;
  while (f % 10 == 0) {
    f /= 10;
    ++e;
  }
  f_out = f;
  e_out = e;
}

// numpy's str() of a float32 (gnx_io.cpp: put_float<float>) into o (at most 16 characters); returns the length.  NaN -> 0.
__device__ __forceinline__ int put_f32(float v, char* o) {
  if (v != v) return 0;
  char* const o0 = o;
  uint32_t bits = __float_as_uint(v);
  if (bits >> 31) {
    *o++ = '-';
    bits &= 0x7fffffffu;
  }
  if (bits == 0x7f800000u) {
    o[0] = 'i'; o[1] = 'n'; o[2] = 'f';
    return (int)(o - o0) + 3;
  }
  if (bits == 0) {
    o[0] = '0'; o[1] = '.'; o[2] = '0';
    return (int)(o - o0) + 3;
  }
  uint32_t f;
  int e10;
  f32_shortest(bits, f, e10);
  char dig[10];
  const int k = f >= 100000000u ? 9 : f >= 10000000u ? 8 : f >= 1000000u ? 7 : f >= 100000u ? 6 : f >= 10000u ? 5 : f >= 1000u ? 4 : f >= 100u ? 3 : f >= 10u ? 2 : 1;
  for (int i = k - 1; i >= 0; --i) {
    dig[i] = (char)('0' + f % 10);
    f /= 10;
  }
  const int e = e10 + k - 1;  // exponent of the first digit
  const double av = (double)__uint_as_float(bits);
  if (av >= 1e-4 && av < 1e16) {
    if (e >= 0) {
      const int ni = e + 1;  // digits before the point
      if (k <= ni) {
        for (int i = 0; i < k; ++i) *o++ = dig[i];
        for (int i = k; i < ni; ++i) *o++ = '0';
        *o++ = '.';
        *o++ = '0';
      } else {
        for (int i = 0; i < ni; ++i) *o++ = dig[i];
        *o++ = '.';
        for (int i = ni; i < k; ++i) *o++ = dig[i];
      }
    } else {
      *o++ = '0';
      *o++ = '.';
      for (int i = 0; i < -e - 1; ++i) *o++ = '0';
      for (int i = 0; i < k; ++i) *o++ = dig[i];
    }
    return (int)(o - o0);
  }
  *o++ = dig[0];
  if (k > 1) {
    *o++ = '.';
    for (int i = 1; i < k; ++i) *o++ = dig[i];
  }
  *o++ = 'e';
  *o++ = e < 0 ? '-' : '+';
  int ae = e < 0 ? -e : e;
  if (ae >= 100) {
    *o++ = (char)('0' + ae / 100);
    ae %= 100;
  }
  *o++ = (char)('0' + ae / 10);
  *o++ = (char)('0' + ae % 10);
  return (int)(o - o0);
}

// the length put_f32 would write, without writing (no per-thread character buffer: k_fb_len stays free of scratch)
__device__ __forceinline__ int len_f32(float v) {
  if (v != v) return 0;
  uint32_t bits = __float_as_uint(v);
  const int sign = (int)(bits >> 31);
  bits &= 0x7fffffffu;
  if (bits == 0x7f800000u || bits == 0) return sign + 3;
  uint32_t f;
  int e10;
  f32_shortest(bits, f, e10);
  const int k = f >= 100000000u ? 9 : f >= 10000000u ? 8 : f >= 1000000u ? 7 : f >= 100000u ? 6 : f >= 10000u ? 5 : f >= 1000u ? 4 : f >= 100u ? 3 : f >= 10u ? 2 : 1;
  const int e = e10 + k - 1;
  const double av = (double)__uint_as_float(bits);
  if (av >= 1e-4 && av < 1e16) return sign + (e >= 0 ? (k <= e + 1 ? e + 3 : k + 1) : 1 - e + k);
  const int ae = e < 0 ? -e : e;
  return sign + k + (k > 1 ? 1 : 0) + 2 + (ae >= 100 ? 3 : 2);
}

// value (w, i = n * A + a) of proba (N, W, A)
__device__ __forceinline__ float fb_value(const float* __restrict__ proba, int64_t W, int A, int64_t w, int64_t i) {
  const int64_t n = i / A;
  const int a = (int)(i - n * A);
  return proba[((size_t)n * W + w) * A + a];
}

// lengths: len[w][i] = 1 + text length; line_len[w] += ...
__global__ __launch_bounds__(256) void k_fb_len(const float* __restrict__ proba, int64_t N, int64_t W, int A, uint8_t* __restrict__ len,
                                                unsigned long long* __restrict__ line_len) {
  const int64_t w = blockIdx.y, NA = N * A;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int l = 0;
  if (i < NA) {
    l = 1 + len_f32(fb_value(proba, W, A, w, i));
    len[(size_t)w * NA + i] = (uint8_t)l;
  }
  __shared__ int red[256];
  red[threadIdx.x] = l;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&line_len[w], (unsigned long long)red[0]);
}

// one block per line: prefix, values, newline, at line_off[w] of the body
constexpr int EB = 1024;
__global__ __launch_bounds__(EB) void k_fb_emit(const float* __restrict__ proba, int64_t N, int64_t W, int A, const uint8_t* __restrict__ len,
                                                const char* __restrict__ pb, const int64_t* __restrict__ po, const int64_t* __restrict__ line_off,
                                                char* __restrict__ body) {
  const int64_t w = blockIdx.x, NA = N * A;
  char* line = body + line_off[w];
  const int64_t plen = po[w + 1] - po[w];
  for (int64_t e = threadIdx.x; e < plen; e += EB) line[e] = pb[po[w] + e];
  __shared__ int wsum[EB / 64];
  __shared__ int64_t base_s;
  if (threadIdx.x == 0) base_s = plen;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t i0 = 0; i0 < NA; i0 += EB) {
    const int64_t i = i0 + threadIdx.x;
    const int l = i < NA ? (int)len[(size_t)w * NA + i] : 0;
    int incl = l;  // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < EB / 64; ++k) {
      const int s = wsum[k];
      before += k < wave ? s : 0;
      total += s;
    }
    const int64_t base = base_s;
    if (i < NA) {
      char* o = line + base + before + incl - l;
      *o = '\t';
      put_f32(fb_value(proba, W, A, w, i), o + 1);
    }
    __syncthreads();  // everybody has read base_s and wsum
    if (threadIdx.x == 0) base_s = base + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) line[base_s] = '\n';
}

}  // namespace

hipError_t gnx_launch_fb_len(const float* d_proba, int64_t N, int64_t W, int A, uint8_t* d_len, unsigned long long* d_line_len, hipStream_t s) {
  const int64_t NA = N * A;
  if (NA <= 0 || W <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_fb_len, dim3((unsigned)((NA + 255) / 256), (unsigned)W), dim3(256), 0, s, d_proba, N, W, A, d_len, d_line_len);
  return hipGetLastError();
}

hipError_t gnx_launch_fb_emit(const float* d_proba, int64_t N, int64_t W, int A, const uint8_t* d_len, const char* d_pb, const int64_t* d_po,
                              const int64_t* d_line_off, char* d_body, hipStream_t s) {
  if (W <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_fb_emit, dim3((unsigned)W), dim3(EB), 0, s, d_proba, N, W, A, d_len, d_pb, d_po, d_line_off, d_body);
  return hipGetLastError();
}
