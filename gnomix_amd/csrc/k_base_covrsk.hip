// k_base_covrsk.hip — covering-random-string-kernel SVC base classifiers on gfx950.
//
// Replaces CovRSKBase.predict_proba (reference src/Base/models.py:195-215): per window
//   K = CovRSK_DP_triangular_numbers(Xw, Xfit)        src/Base/string_kernel.py:91-110
//   sklearn SVC(kernel=callable, probability=True).predict_proba  -> libsvm predict_values /
//   sigmoid_predict / multiclass_probability (sklearn/svm/src/libsvm/svm.cpp, third-party).
//
// The kernel value is an exact integer: over a maximal run of L equal symbols the reference adds
// g(L) = sum_{m in Ms, m<=L} (L-m+1)  (cov_tri counts the m <= run-so-far at every matched position), so
//   K(x,y) = sum over maximal match runs of g(run length).
// Design:
//  * pass 1 (k_pack_bits): X int8 {0,1,2} -> two bit-planes over the reflect-PADDED coordinate (base.py:41-44),
//    so a window is a bit range and symbol equality is ~((xl^yl)|(xh^yh)): 32 SNP compares per 3 VALU ops;
//  * pass 2 (k_covrsk_svc): one wave = 64 query haplotypes of one window; the query's window bits sit in LDS
//    [word][lane]; the support vector is WAVE-UNIFORM, so its bit-planes and dual coefficients come through
//    the scalar unit; match runs are peeled with ctz and looked up in an LDS copy of g; the pairwise decision
//    values accumulate in float64 in libsvm's own order (class-major, SV order inside a class), then the
//    Platt sigmoids and the Wu-Lin-Weng coupling iteration run per lane on LDS-resident [index][lane] arrays.
// Integer-ALU bound: W * n_sv * width symbol compares per haplotype.
#include "gnx_internal.h"

namespace {

__device__ __forceinline__ int64_t pad_src(int64_t p, int64_t C, int64_t ctx) {
  if (p < ctx) return ctx - 1 - p;
  if (p < ctx + C) return p - ctx;
  return C - 1 - (p - ctx - C);
}

// one thread = one 32-SNP word of the padded bit-planes
__global__ __launch_bounds__(256) void k_pack_bits(const int8_t* X, int64_t N, int64_t ldx, int64_t C, int64_t ctx,
                                                    int64_t nwp, uint32_t* planes) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * nwp) return;
  const int64_t n = idx / nwp, wd = idx - n * nwp;
  const int64_t Cp = C + 2 * ctx;
  const int8_t* x = X + n * ldx;
  uint32_t lo = 0, hi = 0;
  const int64_t p0 = wd * 32;
  if (p0 >= ctx && p0 + 32 <= ctx + C) {
    // interior word: 32 consecutive SNPs = 32 consecutive bytes (any alignment); bit k of every byte of a 64-bit group is
    // gathered into one byte by the multiply (the eight partial products land on distinct bits: no carries)
    const int8_t* src = x + (p0 - ctx);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned long long v;
      __builtin_memcpy(&v, src + 8 * q, 8);
      const unsigned long long b0 = ((v & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56;
      const unsigned long long b1 = (((v >> 1) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56;
      lo |= (uint32_t)b0 << (8 * q);
      hi |= (uint32_t)b1 << (8 * q);
    }
  } else {
    for (int b = 0; b < 32; ++b) {
      const int64_t p = p0 + b;
      if (p < Cp) {
        const uint32_t v = (uint32_t)(uint8_t)x[pad_src(p, C, ctx)];
        lo |= (v & 1u) << b;
        hi |= ((v >> 1) & 1u) << b;
      }
    }
  }
  planes[(n * 2 + 0) * nwp + wd] = lo;
  planes[(n * 2 + 1) * nwp + wd] = hi;
}

__device__ __forceinline__ int pair_index(int i, int j, int A) { return i * (2 * A - i - 1) / 2 + (j - i - 1); }

__device__ __forceinline__ double sigmoid_predict(double dec, double pa, double pb) {
  const double f = dec * pa + pb;
  if (f >= 0) return exp(-f) / (1.0 + exp(-f));
  return 1.0 / (1.0 + exp(f));
}

// pass 2a: kernel values + pairwise decision values + Platt sigmoids -> r_ij (i<j) in global memory.
// Small LDS footprint (query bit-planes, g table, the P accumulators of 64 queries) so that many waves are resident:
// the run-peeling loop is latency-bound (LDS lookups, scalar loads), occupancy is what hides it.
template <bool POLY>
__global__ __launch_bounds__(64) void k_covrsk_dec(CovRSKLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x;
  const int w = L.w_first + blockIdx.y;
  const int A = L.A, P = A * (A - 1) / 2;
  const SvcWinDev win = L.win[w];
  const int NW = win.nw, width = win.width, n_sv = win.n_sv;

  size_t off = 0;
  auto carve = [&](size_t bytes) { uint8_t* p = lds + off; off += (bytes + 15) & ~(size_t)15; return p; };
  uint32_t* xq = reinterpret_cast<uint32_t*>(carve((size_t)2 * L.max_nw * 64 * 4));  // [plane][word][lane]
  uint32_t* gl = reinterpret_cast<uint32_t*>(carve((size_t)(L.max_width + 2) * 8));   // g[0..width] (uint32), or run values (double)
  const double* rv = reinterpret_cast<const double*>(gl);
  double* dec = reinterpret_cast<double*>(carve((size_t)P * 64 * 8));                 // [pair][lane]

  const int64_t n = L.n_first + (int64_t)blockIdx.x * 64 + lane;  // haplotype index within the whole batch
  const int64_t n_end = L.n_first + L.n_count;
  const int64_t nc = n < n_end ? n : n_end - 1;

  // ---- query window bits: funnel-shift the padded planes to the window start (w*M) ----
  {
    const int64_t s = (int64_t)w * L.M;
    const int64_t w0 = s >> 5;
    const int sh = (int)(s & 31);
    for (int pl = 0; pl < 2; ++pl) {
      const uint32_t* src = L.planes + (nc * 2 + pl) * L.nwp + w0;
      for (int i = 0; i < NW; ++i) {
        const uint32_t a = src[i], b = src[i + 1];  // planes are padded with 2 zero words
        uint32_t v = sh ? ((a >> sh) | (b << (32 - sh))) : a;
        if (i == NW - 1 && (width & 31)) v &= (1u << (width & 31)) - 1u;
        xq[((size_t)pl * L.max_nw + i) * 64 + lane] = v;
      }
    }
  }
  if constexpr (POLY) {
    for (int i = lane; i <= width; i += 64) reinterpret_cast<double*>(gl)[i] = L.coef[win.rv_off + i];
  } else {
    for (int i = lane; i <= width; i += 64) gl[i] = L.gtab[win.g_off + i];
  }
  for (int p = 0; p < P; ++p) dec[p * 64 + lane] = 0.0;
  __syncthreads();

  const uint32_t tail_mask = (width & 31) ? ((1u << (width & 31)) - 1u) : 0xffffffffu;
  const double* dual = L.coef + win.coef_off;  // (A-1, n_sv)

  const int32_t* cls_start = L.win[w].cls_start;  // from the table, not from the private copy: a run-time index would put it in scratch
  for (int c = 0; c < A; ++c) {
    for (int sv = cls_start[c]; sv < cls_start[c + 1]; ++sv) {
      const uint32_t* yb = L.svbits + win.sv_off + (size_t)sv * 2 * NW;  // wave-uniform -> scalar loads
      double Kd;
      if constexpr (POLY) {
        // polynomial string kernel (string_kernel.py:40-61): contigs = run lengths, one per mismatch plus the final one;
        // K = int(np.sum(contigs ** p) / p) with numpy's pairwise summation order restated element by element
        auto ebits = [&](int i) -> uint32_t {
          const uint32_t e = ~((xq[(size_t)i * 64 + lane] ^ yb[i]) | (xq[((size_t)L.max_nw + i) * 64 + lane] ^ yb[NW + i]));
          return (i == NW - 1) ? (e & tail_mask) : e;
        };
        int n_el = 1;
        for (int i = 0; i < NW; ++i) {
          const int valid = (i == NW - 1 && (width & 31)) ? (width & 31) : 32;
          n_el += valid - __builtin_popcount(ebits(i));
        }
        int pos = 0;  // generator state: next SNP to look at
        auto next_val = [&]() -> double {
          int Lr = 0;
          while (pos < width) {
            const int wi = pos >> 5, b = pos & 31;
            const uint32_t mm = (~ebits(wi)) >> b;  // mismatches from `pos` on (bits past the window read as mismatches)
            if (mm != 0u) {
              const int z = __builtin_ctz(mm);
              if (pos + z < width) { Lr += z; pos += z + 1; return rv[Lr]; }
              Lr += width - pos; pos = width;
              return rv[Lr];
            }
            const int take = min(32 - b, width - pos);
            Lr += take; pos += take;
          }
          return rv[Lr];  // the run closed by the end of the window
        };
        auto leaf = [&](int m) -> double {  // numpy DOUBLE_pairwise_sum for n <= 128
          if (m < 8) {
            double res = 0.;
            for (int i = 0; i < m; ++i) res += next_val();
            return res;
          }
          double r0 = next_val(), r1 = next_val(), r2 = next_val(), r3 = next_val(), r4 = next_val(), r5 = next_val(),
                 r6 = next_val(), r7 = next_val();
          int i = 8;
          for (; i < m - (m % 8); i += 8) {
            r0 += next_val(); r1 += next_val(); r2 += next_val(); r3 += next_val();
            r4 += next_val(); r5 += next_val(); r6 += next_val(); r7 += next_val();
          }
          double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
          for (; i < m; ++i) res += next_val();
          return res;
        };
        // the recursion  PW(n) = PW(n2) + PW(n - n2), n2 = n/2 rounded down to a multiple of 8, n > 128, made iterative
        int fsz[12], fst[12];
        double flv[12];
        int top = 0;
        fsz[0] = n_el; fst[0] = 0;
        double ret = 0.0;
        bool have = false;
        while (top >= 0) {
          if (!have) {
            if (fsz[top] <= 128) { ret = leaf(fsz[top]); have = true; --top; }
            else {
              int n2 = fsz[top] / 2; n2 -= n2 % 8;
              fst[top] = 0;
              fsz[top + 1] = n2; fst[top + 1] = 0;
              ++top;
            }
          } else if (fst[top] == 0) {
            flv[top] = ret; fst[top] = 1; have = false;
            int n2 = fsz[top] / 2; n2 -= n2 % 8;
            fsz[top + 1] = fsz[top] - n2; fst[top + 1] = 0;
            ++top;
          } else {
            ret = flv[top] + ret;
            --top;
          }
        }
        Kd = (double)(long long)(ret / win.poly_p);  // numpy float -> int assignment truncates toward zero
      } else {
        uint32_t K = 0, run = 0;
        for (int i = 0; i < NW; ++i) {
          const uint32_t yl = yb[i], yh = yb[NW + i];
          const uint32_t xl = xq[(size_t)i * 64 + lane], xh = xq[((size_t)L.max_nw + i) * 64 + lane];
          uint32_t e = ~((xl ^ yl) | (xh ^ yh));
          if (i == NW - 1) e &= tail_mask;
          if (e == 0xffffffffu) { run += 32; continue; }
          // trailing ones continue the carried run
          uint32_t t = (uint32_t)__builtin_ctz(~e);
          run += t;
          K += gl[run];
          run = 0;
          e >>= t;
          uint32_t rem = 32 - t;
          while (e) {
            const uint32_t z = (uint32_t)__builtin_ctz(e);
            e >>= z;
            rem -= z;
            const uint32_t o = (uint32_t)__builtin_ctz(~e);  // e has zeros above bit rem-1, so o <= rem
            if (o == rem) { run = o; break; }                // the run touches the end of the word: carry
            K += gl[o];
            e >>= o;
            rem -= o;
          }
        }
        K += gl[run];
        Kd = (double)K;
      }
      // libsvm predict_values order: every pair (i<j) sums its class-i SVs (coef row j-1) then its class-j SVs (row i)
      for (int o = 0; o < A; ++o) {
        if (o == c) continue;
        const int row = (o > c) ? o - 1 : o;
        const int p = (o > c) ? pair_index(c, o, A) : pair_index(o, c, A);
        dec[p * 64 + lane] += dual[(size_t)row * n_sv + sv] * Kd;
      }
    }
  }

  // ---- Platt sigmoids (svm_predict_probability): r_ij, i<j ----
  const double* icpt = dual + (size_t)(A - 1) * n_sv;
  const double* pA = icpt + P;
  const double* pB = pA + P;
  const double min_prob = 1e-7;
  if (n < n_end) {
    double* out = L.rpair + (((size_t)(n - L.n_first)) * L.W + w) * P;
    for (int p = 0; p < P; ++p) {
      const double d = dec[p * 64 + lane] + icpt[p];  // sklearn _intercept_ = -rho
      double v = sigmoid_predict(d, pA[p], pB[p]);
      out[p] = fmin(fmax(v, min_prob), 1 - min_prob);
    }
  }
}

// ---- fast path of pass 2a: branch-free AND-shift counting for the canonical CovSample lengths ----------------
// #matching substrings of length m = popcount(r_m), r_m[t] = AND_{k<m} e[t+k].  With the lengths the reference's
// CovSample(seed=37) always produces (prefix of 1,4,8,39,42,117,376: string_kernel.py:80-89) r_m comes from a
// doubling chain r_2p = r_p & (r_p >> p) plus r_m = r_a & (r_b >> (m-b)) (a+b >= m), all shifts compile-time, the
// NWT-word bit vectors in registers: ~30 straight-line VALU ops per word, no loop, no divergence, no LDS in the loop.
template <int NWT>
struct BitVec {
  uint32_t w[NWT];
};

template <int NWT, int K>
__device__ __forceinline__ BitVec<NWT> shr(const BitVec<NWT>& a) {  // result[t] = a[t + K]
  constexpr int q = K / 32, s = K % 32;
  BitVec<NWT> r;
#pragma unroll
  for (int i = 0; i < NWT; ++i) {
    const uint32_t lo = (i + q < NWT) ? a.w[i + q] : 0u;
    const uint32_t hi = (i + q + 1 < NWT) ? a.w[i + q + 1] : 0u;
    r.w[i] = s ? __funnelshift_r(lo, hi, s) : lo;
  }
  return r;
}

template <int NWT>
__device__ __forceinline__ BitVec<NWT> band(const BitVec<NWT>& a, const BitVec<NWT>& b) {
  BitVec<NWT> r;
#pragma unroll
  for (int i = 0; i < NWT; ++i) r.w[i] = a.w[i] & b.w[i];
  return r;
}

template <int NWT>
__device__ __forceinline__ uint32_t pop(const BitVec<NWT>& a) {
  // v_bcnt_u32_b32 adds its count to an accumulator operand: NWT instructions (hipcc's own lowering of `c += __popc(w)` is a tree of
  // bcnt(w, 0) and v_add3: NWT + NWT/2)
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < NWT; ++i) asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c) : "v"(a.w[i]));
  return c;
}

template <int NWT>
__device__ __forceinline__ bool any_bit(const BitVec<NWT>& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < NWT; ++i) o |= a.w[i];
  return o != 0;
}

// AT > 0: the class count is a compile-time constant — the class loop is unrolled, so the P pairwise decision sums live in
// REGISTERS with static indices (per support vector the LDS version does 2(A-1) dependent read-modify-writes of [pair][lane]
// doubles: ~2 LDS latencies on every iteration's critical path; VALU busy 57 % on config 3).  AT = 0: any A, sums in LDS.
template <int NWT, int AT>
__global__ __launch_bounds__(64) void k_covrsk_dec_fast(CovRSKLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x;
  const int w = L.w_first + blockIdx.y;
  const int A = L.A, P = A * (A - 1) / 2;
  const SvcWinDev win = L.win[w];
  // the launcher groups windows by their word count and instantiates NWT = that count: a compile-time NW lets the support vector's
  // words arrive as a few wide scalar loads instead of one guarded s_load_dword per word
  constexpr int NW = NWT;
  const int width = win.width, n_sv = win.n_sv, n_ms = win.n_ms;
  double* dec = reinterpret_cast<double*>(lds);  // [pair][lane]

  const int64_t n = L.n_first + (int64_t)blockIdx.x * 64 + lane;
  const int64_t n_end = L.n_first + L.n_count;
  const int64_t nc = n < n_end ? n : n_end - 1;

  BitVec<NWT> xl, xh;
  {
    const int64_t s = (int64_t)w * L.M;
    const int64_t w0 = s >> 5;
    const int sh = (int)(s & 31);
    const uint32_t* s0 = L.planes + (nc * 2 + 0) * L.nwp + w0;
    const uint32_t* s1 = L.planes + (nc * 2 + 1) * L.nwp + w0;
#pragma unroll
    for (int i = 0; i < NWT; ++i) {
      uint32_t a = 0, b = 0;
      if (i < NW) {
        a = sh ? ((s0[i] >> sh) | (s0[i + 1] << (32 - sh))) : s0[i];
        b = sh ? ((s1[i] >> sh) | (s1[i + 1] << (32 - sh))) : s1[i];
        if (i == NW - 1 && (width & 31)) { a &= (1u << (width & 31)) - 1u; b &= (1u << (width & 31)) - 1u; }
      }
      xl.w[i] = a;
      xh.w[i] = b;
    }
  }
  BitVec<NWT> valid;  // bit t set iff t < width
#pragma unroll
  for (int i = 0; i < NWT; ++i) valid.w[i] = (i < NW - 1) ? 0xffffffffu : (i == NW - 1 ? ((width & 31) ? ((1u << (width & 31)) - 1u) : 0xffffffffu) : 0u);
  constexpr int PT = AT > 0 ? AT * (AT - 1) / 2 : 1;
  double decr[PT];
#pragma unroll
  for (int p = 0; p < PT; ++p) decr[p] = 0.0;
  if (AT == 0)
    for (int p = 0; p < P; ++p) dec[p * 64 + lane] = 0.0;

  const double* dual = L.coef + win.coef_off;
  // kernel value of the lane's query against support vector `sv` (wave-uniform): substring counts by AND-shift doubling
  struct SvWords { uint32_t w[2 * NWT]; };  // a support vector's two bit-planes: wave-uniform, i.e. scalar registers
  auto load_sv = [&](int sv) -> SvWords {
    const uint32_t* yb = L.svbits + win.sv_off + (size_t)sv * 2 * NW;
    SvWords y;
#pragma unroll
    for (int i = 0; i < 2 * NWT; ++i) y.w[i] = yb[i];
    return y;
  };
  auto kernel_value_of = [&](const SvWords& y) -> uint32_t {
    BitVec<NWT> e;
#pragma unroll
    for (int i = 0; i < NWT; ++i) e.w[i] = ~((xl.w[i] ^ y.w[i]) | (xh.w[i] ^ y.w[NW + i])) & valid.w[i];
    uint32_t K = pop(e);                                              // m = 1
    const BitVec<NWT> r2 = band(e, shr<NWT, 1>(e));
    const BitVec<NWT> r4 = band(r2, shr<NWT, 2>(r2));
    if (n_ms > 1) K += pop(r4);                                       // m = 4
    const BitVec<NWT> r8 = band(r4, shr<NWT, 4>(r4));
    if (n_ms > 2) K += pop(r8);                                       // m = 8
    if (n_ms > 3) {
      const BitVec<NWT> r16 = band(r8, shr<NWT, 8>(r8));
      const BitVec<NWT> r32 = band(r16, shr<NWT, 16>(r16));
      if (__any(any_bit(r32))) {                                      // wave-uniform: some query has a run >= 32
        K += pop(band(r32, shr<NWT, 31>(r8)));                        // m = 39 = [t,t+32) & [t+31,t+39)
        if (n_ms > 4) K += pop(band(r32, shr<NWT, 26>(r16)));         // m = 42 = [t,t+32) & [t+26,t+42)
        if (n_ms > 5) {
          const BitVec<NWT> r64 = band(r32, shr<NWT, 32>(r32));
          if (__any(any_bit(r64))) {
            K += pop(band(r64, shr<NWT, 53>(r64)));                   // m = 117 = [t,t+64) & [t+53,t+117)
            if (n_ms > 6) {
              const BitVec<NWT> r128 = band(r64, shr<NWT, 64>(r64));
              const BitVec<NWT> r256 = band(r128, shr<NWT, 128>(r128));
              K += pop(band(r256, shr<NWT, 248>(r128)));              // m = 376 = [t,t+256) & [t+248,t+376)
            }
          }
        }
      }
    }
    return K;
  };
  auto kernel_value = [&](int sv) -> uint32_t { return kernel_value_of(load_sv(sv)); };
  if constexpr (AT > 0) {
#pragma unroll
    for (int c = 0; c < AT; ++c) {
      // the NEXT support vector's words are requested before this one is evaluated (index clamped: always a valid address), so the
      // scalar-load latency is not paid at the top of every trip
      const int sv_end = win.cls_start[c + 1];
      auto load_dual = [&](int sv, double (&d)[AT - 1]) {  // the vector's AT-1 dual coefficients (rows of the classes other than c)
#pragma unroll
        for (int r = 0; r < AT - 1; ++r) d[r] = dual[(size_t)r * n_sv + sv];
      };
      SvWords ycur = load_sv(min(win.cls_start[c], n_sv - 1));
      double dcur[AT - 1];
      load_dual(min(win.cls_start[c], n_sv - 1), dcur);
      for (int sv = win.cls_start[c]; sv < sv_end; ++sv) {
        const SvWords ynext = load_sv(min(sv + 1, n_sv - 1));
        double dnext[AT - 1];
        load_dual(min(sv + 1, n_sv - 1), dnext);
        const double Kd = (double)kernel_value_of(ycur);
        ycur = ynext;
#pragma unroll
        for (int o = 0; o < AT; ++o) {
          if (o == c) continue;
          const int row = (o > c) ? o - 1 : o;
          const int p = (o > c) ? (c * (2 * AT - c - 1) / 2 + (o - c - 1)) : (o * (2 * AT - o - 1) / 2 + (c - o - 1));
          decr[p] += dcur[row] * Kd;   // libsvm's order: class-major, SV order inside a class
        }
#pragma unroll
        for (int r = 0; r < AT - 1; ++r) dcur[r] = dnext[r];
      }
    }
  } else {
    // class bounds straight from the window table (wave-uniform scalar loads): indexing the private COPY of the window with a
    // run-time class put its cls_start[36] in scratch (208 B per lane on every generic-A instance)
    const int32_t* cls_start = L.win[w].cls_start;
    for (int c = 0; c < A; ++c) {
      for (int sv = cls_start[c]; sv < cls_start[c + 1]; ++sv) {
        const double Kd = (double)kernel_value(sv);
        for (int o = 0; o < A; ++o) {
          if (o == c) continue;
          const int row = (o > c) ? o - 1 : o;
          const int p = (o > c) ? pair_index(c, o, A) : pair_index(o, c, A);
          dec[p * 64 + lane] += dual[(size_t)row * n_sv + sv] * Kd;
        }
      }
    }
  }
  const double* icpt = dual + (size_t)(A - 1) * n_sv;
  const double* pA = icpt + P;
  const double* pB = pA + P;
  const double min_prob = 1e-7;
  if (n < n_end) {
    double* out = L.rpair + (((size_t)(n - L.n_first)) * L.W + w) * P;
    if constexpr (AT > 0) {
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        const double v = sigmoid_predict(decr[p] + icpt[p], pA[p], pB[p]);
        out[p] = fmin(fmax(v, min_prob), 1 - min_prob);
      }
    } else {
      for (int p = 0; p < P; ++p) {
        const double d = dec[p * 64 + lane] + icpt[p];
        double v = sigmoid_predict(d, pA[p], pB[p]);
        out[p] = fmin(fmax(v, min_prob), 1 - min_prob);
      }
    }
  }
}

// pass 2b: multiclass_probability (Wu, Lin, Weng 2004) per (haplotype, window); thread-private arrays live in LDS as
// [index][thread] (dynamic indexing without scratch).  Tiny next to pass 2a.
__global__ __launch_bounds__(64) void k_svc_couple(CovRSKLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x;
  const int A = L.A, P = A * (A - 1) / 2;
  double* rr = reinterpret_cast<double*>(lds);          // [P][64]
  double* Q = rr + (size_t)P * 64;                      // [A*A][64]
  double* Qp = Q + (size_t)A * A * 64;                  // [A][64]
  double* pr = Qp + (size_t)A * 64;                     // [A][64]
  const int64_t idx = (int64_t)blockIdx.x * 64 + lane;  // (n_local, w) flattened
  const int64_t total = L.n_count * L.W;
  const int64_t ic = idx < total ? idx : total - 1;
  const double* in = L.rpair + (size_t)ic * P;
  for (int p = 0; p < P; ++p) rr[p * 64 + lane] = in[p];
  auto r = [&](int i, int j) -> double {
    return (i < j) ? rr[pair_index(i, j, A) * 64 + lane] : 1.0 - rr[pair_index(j, i, A) * 64 + lane];
  };
  const int k = A;
  const int max_iter = k > 100 ? k : 100;
  const double eps = 0.005 / k;
#define QQ(t, j) Q[((size_t)(t) * k + (j)) * 64 + lane]
#define QP(t) Qp[(size_t)(t) * 64 + lane]
#define PP(t) pr[(size_t)(t) * 64 + lane]
  for (int t = 0; t < k; ++t) {
    PP(t) = 1.0 / k;
    double qtt = 0.0;
    for (int j = 0; j < t; ++j) { qtt += r(j, t) * r(j, t); QQ(t, j) = QQ(j, t); }
    for (int j = t + 1; j < k; ++j) { qtt += r(j, t) * r(j, t); QQ(t, j) = -r(j, t) * r(t, j); }
    QQ(t, t) = qtt;
  }
  for (int iter = 0; iter < max_iter; ++iter) {
    double pQp = 0.0;
    for (int t = 0; t < k; ++t) {
      double q = 0.0;
      for (int j = 0; j < k; ++j) q += QQ(t, j) * PP(j);
      QP(t) = q;
      pQp += PP(t) * q;
    }
    double max_error = 0.0;
    for (int t = 0; t < k; ++t) { const double e = fabs(QP(t) - pQp); if (e > max_error) max_error = e; }
    if (max_error < eps) break;
    for (int t = 0; t < k; ++t) {
      const double diff = (-QP(t) + pQp) / QQ(t, t);
      PP(t) += diff;
      pQp = (pQp + diff * (diff * QQ(t, t) + 2 * QP(t))) / (1 + diff) / (1 + diff);
      for (int j = 0; j < k; ++j) { QP(j) = (QP(j) + diff * QQ(t, j)) / (1 + diff); PP(j) /= (1 + diff); }
    }
  }
  if (idx < total) {
    const int64_t nl = idx / L.W, w = idx - nl * L.W;
    const size_t o = ((size_t)(L.n_first + nl) * L.W + w) * A;
    for (int a = 0; a < A; ++a) {
      const double v = PP(a);
      if (L.b64) L.b64[o + a] = v;
      if (L.b32) L.b32[o + a] = (float)v;
    }
  }
#undef QQ
#undef QP
#undef PP
}

// The same iteration with a compile-time class count: r, Q, Qp and p live in REGISTERS with static indices (the generic kernel
// keeps them in LDS as [index][thread]: every one of the ~k^2 operands of an iteration is an LDS round trip on the lane's
// critical path).  Identical operations in identical order -> identical results.
template <int AT>
__global__ __launch_bounds__(64) void k_svc_couple_reg(CovRSKLaunch L) {
  constexpr int k = AT, P = AT * (AT - 1) / 2;
  const int lane = threadIdx.x;
  const int64_t idx = (int64_t)blockIdx.x * 64 + lane;
  const int64_t total = L.n_count * L.W;
  const int64_t ic = idx < total ? idx : total - 1;
  const double* in = L.rpair + (size_t)ic * P;
  double rr[P];
#pragma unroll
  for (int p = 0; p < P; ++p) rr[p] = in[p];
  const int max_iter = k > 100 ? k : 100;
  const double eps = 0.005 / k;
  // R[i][j] = r_ij (i < j from the pairwise pass, j < i as 1 - r_ji): the full matrix in registers, every index static
  double R[AT][AT], Q[AT][AT], Qp[AT], pr[AT];
#pragma unroll
  for (int i = 0; i < AT; ++i)
#pragma unroll
    for (int j = 0; j < AT; ++j)
      R[i][j] = (i < j) ? rr[i * (2 * AT - i - 1) / 2 + (j - i - 1)] : (i > j) ? 1.0 - rr[j * (2 * AT - j - 1) / 2 + (i - j - 1)] : 0.0;
#pragma unroll
  for (int t = 0; t < k; ++t) {
    pr[t] = 1.0 / k;
    double qtt = 0.0;
#pragma unroll
    for (int j = 0; j < k; ++j)
      if (j != t) qtt += R[j][t] * R[j][t];   // (j < t first, then j > t: libsvm's order)
    Q[t][t] = qtt;
#pragma unroll
    for (int j = 0; j < k; ++j)
      if (j > t) { Q[t][j] = -R[j][t] * R[t][j]; Q[j][t] = Q[t][j]; }
  }
  for (int iter = 0; iter < max_iter; ++iter) {
    double pQp = 0.0;
#pragma unroll
    for (int t = 0; t < k; ++t) {
      double q = 0.0;
#pragma unroll
      for (int j = 0; j < k; ++j) q += Q[t][j] * pr[j];
      Qp[t] = q;
      pQp += pr[t] * q;
    }
    double max_error = 0.0;
#pragma unroll
    for (int t = 0; t < k; ++t) { const double e = fabs(Qp[t] - pQp); if (e > max_error) max_error = e; }
    if (max_error < eps) break;
#pragma unroll
    for (int t = 0; t < k; ++t) {
      const double diff = (-Qp[t] + pQp) / Q[t][t];
      pr[t] += diff;
      pQp = (pQp + diff * (diff * Q[t][t] + 2 * Qp[t])) / (1 + diff) / (1 + diff);
#pragma unroll
      for (int j = 0; j < k; ++j) { Qp[j] = (Qp[j] + diff * Q[t][j]) / (1 + diff); pr[j] /= (1 + diff); }
    }
  }
  if (idx < total) {
    const int64_t nl = idx / L.W, w = idx - nl * L.W;
    const size_t o = ((size_t)(L.n_first + nl) * L.W + w) * AT;
#pragma unroll
    for (int a = 0; a < AT; ++a) {
      const double v = pr[a];
      if (L.b64) L.b64[o + a] = v;
      if (L.b32) L.b32[o + a] = (float)v;
    }
  }
}

}  // namespace

size_t gnx_covrsk_lds_bytes(int A, int max_nw, int max_width) {
  auto r16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
  const int P = A * (A - 1) / 2;
  const size_t dec = r16((size_t)2 * max_nw * 64 * 4) + r16((size_t)(max_width + 2) * 8) + r16((size_t)P * 64 * 8);
  const size_t couple = (size_t)(P + A * A + 2 * A) * 64 * 8;
  return dec > couple ? dec : couple;
}

hipError_t gnx_launch_pack_bits(const int8_t* X, int64_t N, int64_t ldx, int64_t C, int64_t ctx, int64_t nwp,
                                uint32_t* planes, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  const int64_t total = N * nwp;
  hipLaunchKernelGGL(k_pack_bits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, N, ldx, C, ctx, nwp, planes);
  return hipGetLastError();
}

hipError_t gnx_launch_covrsk(const CovRSKLaunch& L0, hipStream_t s0) {
  if (L0.N <= 0) return hipSuccess;
  const int A = L0.A, P = A * (A - 1) / 2;
  auto r16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
  const size_t lds_dec = r16((size_t)2 * L0.max_nw * 64 * 4) + r16((size_t)(L0.max_width + 2) * 8) + r16((size_t)P * 64 * 8);
  const size_t lds_cpl = (size_t)(P + A * A + 2 * A) * 64 * 8;
  GNX_LDS_OPTIN(lds_dec, k_covrsk_dec<false>);
  GNX_LDS_OPTIN(lds_dec, k_covrsk_dec<true>);
  GNX_LDS_OPTIN(lds_cpl, k_svc_couple);
  // the pairwise probabilities travel through a bounded global buffer: haplotypes in chunks of rpair_haps
  for (int64_t n0 = 0; n0 < L0.N; n0 += L0.rpair_haps) {
    CovRSKLaunch L = L0;
    L.n_first = n0;
    L.n_count = (L0.N - n0 < L0.rpair_haps) ? L0.N - n0 : L0.rpair_haps;
    // consecutive windows with the same (fast-path word count) go in one grid.  A grid that cannot fill the chip (the one wider last
    // window: a dozen waves walking 1 400 support vectors, 1.5 ms at 1 % VALU) goes to the side stream, next to the main grid
    const bool fork = L0.aux != nullptr;
    bool forked = false;
    if (fork) (void)hipEventRecord(L0.ev_fork, s0);  // the side stream may start once this chunk's buffers are free
    for (int w0 = 0; w0 < L.W;) {
      const int key = L0.host_fast_nw[w0];
      int w1 = w0 + 1;
      while (w1 < L.W && L0.host_fast_nw[w1] == key) ++w1;
      L.w_first = w0;
      const dim3 grid((unsigned)((L.n_count + 63) / 64), (unsigned)(w1 - w0));
      const bool small = fork && (int64_t)grid.x * grid.y < (int64_t)4 * std::max(L0.n_cu, 1) && w1 - w0 < L.W;
      hipStream_t s = s0;
      if (small) {
        if (!forked) { (void)hipStreamWaitEvent(L0.aux, L0.ev_fork, 0); forked = true; }
        s = L0.aux;
      }
      const size_t lds_fast = (size_t)P * 64 * 8;
      switch (key) {
#define GNX_FAST_CASE(NWT_)                                                                                         \
  case NWT_:                                                                                                        \
    if (A == 7) hipLaunchKernelGGL((k_covrsk_dec_fast<NWT_, 7>), grid, dim3(64), lds_fast, s, L);                    \
    else hipLaunchKernelGGL((k_covrsk_dec_fast<NWT_, 0>), grid, dim3(64), lds_fast, s, L);                          \
    break;
        GNX_FAST_CASE(1) GNX_FAST_CASE(2) GNX_FAST_CASE(3) GNX_FAST_CASE(4) GNX_FAST_CASE(5) GNX_FAST_CASE(6) GNX_FAST_CASE(7)
        GNX_FAST_CASE(8) GNX_FAST_CASE(9) GNX_FAST_CASE(10) GNX_FAST_CASE(11) GNX_FAST_CASE(12) GNX_FAST_CASE(13)
        GNX_FAST_CASE(14) GNX_FAST_CASE(15) GNX_FAST_CASE(16)
#undef GNX_FAST_CASE
        case -1: hipLaunchKernelGGL(k_covrsk_dec<true>, grid, dim3(64), lds_dec, s, L); break;  // polynomial string kernel
        default: hipLaunchKernelGGL(k_covrsk_dec<false>, grid, dim3(64), lds_dec, s, L); break;  // 0: generic run peeling
      }
      w0 = w1;
    }
    const int64_t total = L.n_count * L.W;
    hipStream_t s = s0;
    if (forked) { (void)hipEventRecord(L0.ev_join, L0.aux); (void)hipStreamWaitEvent(s0, L0.ev_join, 0); }
    if (A == 7) hipLaunchKernelGGL(k_svc_couple_reg<7>, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, s, L);
    else if (A == 3) hipLaunchKernelGGL(k_svc_couple_reg<3>, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, s, L);
    else hipLaunchKernelGGL(k_svc_couple, dim3((unsigned)((total + 63) / 64)), dim3(64), lds_cpl, s, L);
  }
  return hipGetLastError();
}
