// gnx_internal.h — shared between the C-ABI translation unit and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/gnomix_hip.h"

// ------------------------------------------------------------------------------------------------
// context / model
// ------------------------------------------------------------------------------------------------
struct gnx_devbuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct gnx_prof_pair {
  hipEvent_t a, b;
  int kid;
};

// Development / test knobs.  Environment variables are read ONCE per context in gnx_init (never on a launch path);
// every field's 0 means "the built-in choice".
struct gnx_tune {
  int lr_bpc = 0, lr_want = 0;          // GNX_LR_BPC, GNX_LR_WANT: window ranges of the logistic pass
  int lr_mt = 0, lr_waves = 0;          // GNX_LR_TUNE="mt,waves": tile shape of the logistic pass
  int lr_flags = 0;                     // GNX_LR_FLAGS: ablation switches
  int lr_dl = -1;                       // GNX_LR_DL: 1 = LDS-direct loads (k_base_logistic_i8_dl) always, 0 = never; default: for 2
                                        // column tiles (A > 8 at the default context), where it measured 2.5 vs 3.1 ms (A = 12,
                                        // chr22); with one column tile both kernels run at the same 1.14-1.16 ms
  int lr_nbuf = 0;                      // GNX_LR_NBUF: ring slots of the LDS-direct kernel
  int p2_mt = 0, p2_cw = 0, p2_ew = 0, p2_xsn = 0, p2_nbuf = 0;  // GNX_P2_TUNE="mt,compute waves,epilogue waves,X stages,plane slots":
                                        // shape of the 2-bit-native logistic pass (development)
  int p2_decline = 0;                   // GNX_P2_DECLINE=1 (tests): the 2-bit launcher reports "no instantiation fits" -> the widening fallback runs
  int lr_p2 = 1;                        // GNX_LR_P2=0: packed (2-bit) input is widened to int8 in HBM and run through the int8 kernels
                                        // instead of k_base_logistic_p2 (A/B runs; the outputs are bit-identical)
  int lr_flat = -1;                     // GNX_LR_FLAT: 1 = flat column tiles (k_base_logistic_i8_fl) wherever built, 0 = never
  int lr_w512 = 0;                      // GNX_LR_W512=1: 512 rows per block, 64-SNP steps (k_base_logistic_i8_w512)
  int lr_ws = 0, lr_ws_pw = 2;          // GNX_LR_WS=1: wave-specialised kernel (k_base_logistic_i8_ws); GNX_LR_WS_PW: producer waves (2, 4)
  int sm_nw = 0;                        // GNX_SM_NW: waves per block of the rank smoother
  int sm_pair = 1;                      // GNX_SM_PAIR=0: one tree at a time per lane in the rank smoother (default: two)
  int smf_rpl = 0, smf_nw = 0;          // GNX_SM_TUNE="rpl,nw": float smoother
  int crf_impl = 0;                     // GNX_CRF_IMPL=scan|row|lanes (default: row for up to 16 labels, lanes above); quad in the EXPERIMENTS build
  int crf_flags = 0;                    // GNX_CRF_FLAGS bit 0: row kernel computes psi itself instead of reading the pre-pass's (slower)
  int forest_threads = 0;               // GNX_FOREST_T
  int forest_wrun = 0;                  // GNX_FOREST_WRUN: windows per block of the forest bases
  int forest_halves = 0;                // GNX_FOREST_H=1: one wave group per tile (default: two for the boosted-tree base)
  int forest_impl = 0;                  // GNX_FOREST_IMPL=1: k_base_forest (256-haplotype tile, register prefetch) for the boosted-tree
                                        // base too; default 2: k_base_forest2 (two blocks per CU) wherever its tile fits
  int forest_flags = 0;                 // GNX_FOREST_FLAGS: ablation (1 = no walks, 2 = no register prefetch, 4 = no incremental staging)
  int64_t host_batch = 0;               // GNX_HOST_BATCH: haplotypes per staging batch of the host-pointer entry points
  int h2d_overlap = 1;                  // GNX_H2D_OVERLAP=0: serial staging (one stream) in the host-pointer entry points
  int lr_lds_pad = 0, sm_lds_pad = 0;   // GNX_LDS_PAD="lr,sm": extra dynamic LDS bytes (occupancy experiments: scripts/dev/overlap_probe.py)
  int gnofix_impl = 0;                  // GNX_GNOFIX_IMPL=f32: the float32-strip kernel (k_gnofix_f32) even where the rank kernel runs
  int gnofix_aux = 1;                   // GNX_GNOFIX_AUX=0: the input-only pre-passes of k_gnofix on the context stream instead of beside the smoother
  int gnofix_threads = 0;               // GNX_GNOFIX_T: threads per individual of k_gnofix (192, 256, 384, 512)
  int debug = 0;                        // GNX_DEBUG
};

// opt a kernel into its dynamic LDS size; a refusal is reported by the launcher instead of surfacing later as an
// opaque launch failure
#define GNX_LDS_OPTIN(bytes, ...)                                                                                  \
  do {                                                                                                             \
    const hipError_t e_optin_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&__VA_ARGS__),                   \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));     \
    if (e_optin_ != hipSuccess) return e_optin_;                                                                   \
  } while (0)

struct gnx_ctx {
  int device = 0;
  gnx_tune tune;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool usable = false;
  std::string err;
  int n_cu = 256;
  // grow-only device workspaces (host-pointer entry points stage through these)
  gnx_devbuf ws_x, ws_b32, ws_b64, ws_p32, ws_p64, ws_lab, ws_misc, ws_scale, ws_bits, ws_lastrow, ws_rpair, ws_y0, ws_cal, ws_marg;
  // host-pointer pipeline (gnx_infer / gnx_infer_packed): copy-in and copy-out streams next to the compute stream, created on
  // first use; buffers alternate between two halves of the staging workspaces
  // side stream of the CovRSK base: window groups too small to fill the chip (the wider last window) run beside the main grid
  hipStream_t s_aux = nullptr;
  hipEvent_t ev_aux[2] = {nullptr, nullptr};
  hipStream_t s_in = nullptr, s_out = nullptr;
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
  gnx_devbuf ws_pk, ws_xu, ws_xu2, ws_psi;  // ws_xu2: int8 rows widened from packed rows that themselves live in ws_xu
  gnx_devbuf ws_rank;  // k_smooth_ranks -> k_smooth_xgb_h64
  gnx_devbuf ws_fb, ws_fb_body;  // gnx_write_fb_dev: probabilities / lengths / tables, and the file's body as text
  gnx_devbuf ws_gt2, ws_src, ws_gt2o;  // file path (gnx_api_vcf.hip): variant-major 2-bit genotypes, column map, phased rows
  // profiling
  bool prof = false;
  std::vector<gnx_prof_pair> prof_pending;
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[GNX_K_COUNT] = {0};
  int64_t prof_n[GNX_K_COUNT] = {0};
};

// ---- logistic base (k_base_logistic.hip) ---------------------------------------------------------
// The padded chromosome is processed in "pieces" that end exactly where a window ends, each piece in
// chunks of 64 SNPs (one 16-byte load per lane).  Weights are stored in MFMA-fragment order.
struct BaseLRDev {
  const double* V = nullptr;        // f64 path: [n_chunks][16 steps][NT][64 lanes]
  const int8_t* V8 = nullptr;       // i8 path: [n_chunks][NT][7 limbs][64 lanes][16 bytes] balanced base-256 digits
  const int8_t* V8F = nullptr;      // i8 path, flat column tiles (k_base_logistic_i8_fl): [n_chunks][NF][64 lanes][16 bytes], column q = slot * 7 + limb
  const double* wscale = nullptr;   // i8 path: [W] 2^-f_w
  const double* icpt = nullptr;     // [W][A]
  const int32_t* chunk_j0 = nullptr;     // [n_chunks] first real SNP of the chunk
  const int32_t* chunk_flush0 = nullptr; // [n_chunks] first window flushed after this chunk (-1 none)
  const int32_t* chunk_nflush = nullptr; // [n_chunks] number of windows flushed after this chunk
  const int32_t* win_chunk0 = nullptr;   // [W] first chunk a block must start from to compute window w
  const int32_t* win_chunk1 = nullptr;   // [W] one past the chunk after which window w is flushed
  int32_t n_chunks = 0;
  int32_t R = 0;   // windows simultaneously active = column slots
  int32_t NT = 0;  // 16-column tiles
  int32_t NF = 0;  // flat column tiles of V8F (0: not built)
  int32_t max_piece_chunks = 0;  // longest piece, in chunks
  // 2-bit-native pass (k_base_logistic_p2.hip): the same pieces walked in RUNS of 256 SNPs (64 packed bytes per haplotype row,
  // dword-aligned: a piece starts at SNP b0 & ~15, the up to 15 SNPs before b0 meet zero weights); one run = 4 MFMA entries
  const int8_t* V2 = nullptr;            // [n_runs][EPR entries][NT2][7 limbs][64 lanes][16 bytes], k order of the in-register unpack
  const int32_t* run_byte = nullptr;     // [n_runs] byte offset of the run within a packed row
  const int32_t* run_flush0 = nullptr;   // [n_runs] first window flushed after this run (-1 none)
  const int32_t* run_nflush = nullptr;   // [n_runs] number of windows flushed after this run
  const int32_t* win_run0 = nullptr;     // [W] first run a block must start from to compute window w
  const int32_t* win_run1 = nullptr;     // [W] one past the run after which window w is flushed
  int32_t n_runs = 0;
  int32_t EPR = 0;   // MFMA entries per run of V2: 4 (runs of 256 SNPs, 64 packed bytes per row) or 8 (512 SNPs, 128 bytes)
  int32_t NT2 = 0;   // column tiles of V2: 1 (column = slot * A + class, R * A <= 16) or R (one tile per slot, column = class; A <= 16)
  // FLAT column tiles of the 2-bit pass for R * A == 24 class columns (A = 12 at the default context): flat column q = 24 * limb + column,
  // ceil(24 * 7 / 16) = 11 tiles instead of 2 x 7 — 21 % fewer plane bytes and MFMAs, and at 88 accumulator registers per 32 rows both
  // slots fit one wave, so X is read ONCE (k_base_logistic_p2f); same runs and tables as V2.  NULL: not built.
  const int8_t* V2F = nullptr;           // [n_runs][EPR entries][11 flat tiles][64 lanes][16 bytes]
};
constexpr int GNX_LR_FLAT_COLS = 24;     // class columns per SNP the flat layout is built for
constexpr int GNX_LR_FLAT_TILES = 11;    // ceil(24 * 7 / 16)

struct BaseLRLaunch {
  const int8_t* X;
  const int8_t* last_row;  // zero-padded (64 B) device copy of haplotype N-1: the only row whose tail reads could leave X
  int64_t N, ldx;
  BaseLRDev d;
  const int32_t* h_win_chunk0;  // HOST copies of d.win_chunk0/1 (launcher only: exact size of the per-block LDS tables)
  const int32_t* h_win_chunk1;
  int32_t W, A, wch;    // wch = windows per block
  int32_t n_htiles;     // i8 path: haplotype tiles (1-D XCD-aware grid)
  int32_t n_rg8;        // 2-bit pass: groups of 8 window ranges (the grid holds NT2 passes of n_rg8 * 8 ranges x n_htiles tiles)
  int32_t flags;        // development ablation switches (0 in production)
  int32_t max_chunks;   // i8 path: upper bound of chunks one block walks (sizes its LDS tables)
  int32_t max_wins;     // i8 path: upper bound of windows one block may flush
  float* b32;
  double* b64;
  unsigned long long* dbg = nullptr;  // development (GNX_DEBUG & 2, k_base_logistic_p2f): per block 16 cycle counters (where the waves' time goes)
};

// ---- xgb smoother (k_smooth_xgb.hip) ----------------------------------------------------------------
// Trees are re-ordered class-major and every tree is expanded to a complete binary tree of depth D >= 1,
// heap order (children of 1-based node j are 2j, 2j+1; go right iff !(f < thr)):
//   [ 2^(D-1) last-level nodes x 16 B {uint32 feature_byte_offset, float threshold, float leaf_left, float leaf_right}
//   | 2^(D-1)-1 upper nodes x 8 B {uint32 feature_byte_offset, float threshold} | pad to 16 B ]
// The last split and its two leaves arrive in ONE 16-byte read (192 B per tree at D=4).
struct SmoothXGBDev {
  const uint8_t* packed = nullptr;   // n_trees * tree_bytes
  const int32_t* group_tree0 = nullptr;  // [n_groups+1] first tree of each staging group
  const int32_t* group_class = nullptr;  // [n_groups]
  int32_t n_groups = 0, n_trees = 0, D = 0, tree_bytes = 0, max_group = 0;
  float base_score = 0.5f;
  // rank-quantised copy for k_smooth_xgb_rk (NULL when the ensemble does not fit 16-bit ranks / offsets):
  // every split threshold is replaced by its index in the sorted list of all thresholds, every base probability by
  // the number of thresholds <= it, so `p < threshold` becomes a 16-bit integer compare with the same outcome.
  // Per tree: 2^D node words (heap slot 0 unused; (rank field << 16) | byte offset into the class-major u16 strip)
  // followed by 2^D float leaves.
  const uint8_t* rk_packed = nullptr;
  const float* rk_thr = nullptr;         // [rk_K] sorted distinct finite thresholds
  const uint32_t* rk_lut = nullptr;      // [1024] first | (last << 16) candidate index per 1/1024-wide bucket
  const int32_t* rk_group_tree0 = nullptr;
  const int32_t* rk_group_class = nullptr;
  int32_t rk_K = 0, rk_steps = 0, rk_stride = 0, rk_tree_bytes = 0, rk_n_groups = 0, rk_max_group = 0;
  int32_t rk_rpl = 0;                    // 64-window segments per strip the node offsets were laid out for
  // lane = haplotype copy for k_smooth_xgb_h64 (same ranks, rk_thr / rk_lut): per tree 2^D nodes of 8 bytes {byte offset of the
  // feature's slot = (s * A + a) * 128, rank field} (heap slot 0 unused) followed by 2^D float leaves, padded to 16 bytes
  const uint8_t* h8_packed = nullptr;
  const int32_t* h8_group_tree0 = nullptr;
  const int32_t* h8_group_class = nullptr;
  int32_t h8_tree_bytes = 0, h8_n_groups = 0, h8_max_group = 0;
  // pointer-node copy for k_smooth_xgb_rk<.., PTR> (same strip, ranks and offsets as rk_*): per tree 2^D slots of 8 bytes
  // {w0 = rank field << 16 | strip offset, w1 = address of the left child | address of the right child << 16} (addresses relative
  // to the staging group's first byte; children of the last level = the leaves) followed by 2^D float leaves; slot 0 carries
  // {w0 of node 2, w0 of node 3}, the root slot {w0 of the root, w1 of node 2}: the tree's first 16 bytes are levels 0 and 1
  const uint8_t* rp_packed = nullptr;
  const int32_t* rp_group_tree0 = nullptr;
  const int32_t* rp_group_class = nullptr;
  int32_t rp_tree_bytes = 0, rp_n_groups = 0, rp_max_group = 0;
  int32_t impl = 0;                      // 1 rk, 2 h64, 3 rk with pointer nodes (GNX_SMOOTH_IMPL at model load)
  // rank copy for k_gnofix: per tree 2^D node words (heap order; rank field << 16 | byte offset (a * gf_pitch + s) * 2 of feature
  // s * A + a in the [class][gf_pitch] u16 tile) followed by 2^D float leaves; class-major tree order (class_tree0)
  const uint32_t* gf_packed = nullptr;
  int32_t gf_pitch = 0, gf_max_class = 0;
  // bit-sliced copy for k_smooth_xgb_bs (ensembles of depth <= 4; NULL otherwise), trees in the same class-major order, every tree
  // a complete depth-4 heap.  The kernel never walks: per chunk of bs_wc windows it sorts each class's base probabilities once
  // (per-class ranks -> counting sort), keeps for every prefix length n of that order the bitmap "window is NOT among the n
  // smallest" (rows of the table G), and a node (class a, threshold k, window offset s) is the row n = #{p < threshold} shifted by
  // s: 32 windows per 32-bit operation.  Node j (heap 1..15) of a tree = words 2j, 2j+1 of its 32-word block:
  //   w0 = (s & 31) | (LDS byte address of the node's counter) << 16   (byte 1: zero)
  //   w1 = LDS byte address of word s >> 5 of class a's row 0 (rows are GnxBsLayout::rb bytes apart)
  const uint32_t* bs_nodes = nullptr;    // [n_trees][32]
  const float* bs_leaves = nullptr;      // [n_trees][16]
  const float* bs_thr = nullptr;         // per-class sorted distinct finite thresholds: class c = [bs_uoff[c], bs_uoff[c+1])
  const uint32_t* bs_lut = nullptr;      // [A][1024] first | (last << 16) candidate per 1/1024-wide bucket, relative to the class
  const int32_t* bs_uoff = nullptr;      // [A+1]
  const int32_t* bs_binoff = nullptr;    // [A+1] first counter of class c (ranks 0..K_c, then the NaN / outside-the-chromosome bin)
  const int32_t* bs_class_tree0 = nullptr;  // [A+1]
  int32_t bs_steps = 0, bs_nbins = 0, bs_wc = 0, bs_nthr = 0, bs_maxbins = 0;
};

// LDS map of k_smooth_xgb_bs (bytes from the block's first LDS byte, which the kernel checks to be address 0: the node words carry
// absolute addresses).  wp = padded windows per chunk (<= 255: counters are bytes), nr = wp + 1 rows per class, rw words per row
// (one beyond the last real one: a shifted read takes two), nseg segments of sl prefix lengths the rows are built in.
struct GnxBsLayout {
  int32_t wp, nr, rw, rb, nbins, hist_bytes, off_cnt, off_hist, off_P, off_bin, off_pi, off_seg, off_wtot, nseg, sl, total;
  uint32_t inv_sl;  // ceil(2^32 / sl): n / sl = umulhi(n, inv_sl) for n < 256
};
GnxBsLayout gnx_bs_layout(int A, int S, int wc, int nbins);  // k_smooth_xgb_bs.hip

// 32-bit words per tree of SmoothXGBDev::gf_packed (2^D node words in heap order, slot 0 unused, then 2^D float leaves: the leaf of
// heap index j is word j) and the zero trees that follow the last one (k_gnofix walks past a lane's range without clamping)
__host__ __device__ inline int gnx_gf_tree_words(int D) { return 2 << D; }
constexpr int GNX_GF_PAD_TREES = 24;

constexpr int GNX_RK_RPL_MAX = 6;  // most 64-window segments per strip the rank kernel is instantiated for

static inline int gnx_tree_bytes(int D) {
  const int half = 1 << (D - 1);
  return ((half * 16 + (half - 1) * 8) + 15) & ~15;
}

#if defined(__HIPCC__)
// generic-pointer walker (global or LDS tree, global or LDS feature row); row is indexed in BYTES by the node
__device__ __forceinline__ float gnx_walk(const uint8_t* tb, const uint8_t* row, int D) {
  const uint32_t half = 1u << (D - 1);
  uint32_t j = 1;
  for (int d = 0; d < D - 1; ++d) {
    uint2 nd;
    __builtin_memcpy(&nd, tb + half * 16 + (j - 1) * 8, 8);
    float fv;
    __builtin_memcpy(&fv, row + nd.x, 4);
    j = 2 * j + ((fv < __uint_as_float(nd.y)) ? 0u : 1u);
  }
  uint4 n4;
  __builtin_memcpy(&n4, tb + (j - half) * 16, 16);
  float fv;
  __builtin_memcpy(&fv, row + n4.x, 4);
  return (fv < __uint_as_float(n4.y)) ? __uint_as_float(n4.z) : __uint_as_float(n4.w);
}
#endif

struct SmoothXGBLaunch {
  const void* B;     // (N, W, A) float32 or float64
  int32_t b_is_f64;
  int64_t N;
  int32_t W, A, S;
  SmoothXGBDev d;
  float* proba;      // (N, W, A): the float kernel parks its margins here between class passes
  float* marg;       // (A, N*W) class-major margin scratch of the rank kernel (coalesced parking)
  double* proba64;   // optional widened copy
  int32_t* labels;   // optional
};


// ---- crf smoother (k_smooth_crf.hip) ------------------------------------------------------------------
struct SmoothCRFLaunch {
  const void* B;          // (N, W, A) float64 or float32
  int32_t b_is_f64;
  int64_t N;
  int32_t W, A;
  const double* state;    // device (A, A) theta[a][y]
  const double* etrans;   // device (A, A) exp(tau)[y'][y]
  double* psi;            // (N, W, A) scratch: exp(theta'B) per window and label, computed once for both directions
  double* alpha;          // (N, W, A) scratch for the scaled forward variables (may alias proba64)
  double* scale;          // (N, W, 2) scratch (the 9..16-label kernel parks (c_t, 1/c_t); the others use (N, W) of it)
  double* proba64;        // optional
  float* proba32;         // optional
  int32_t* labels;        // optional
  int32_t norm_mask;      // k_smooth_crf_ck takes the forward scale at windows t with (t & norm_mask) == norm_mask (0, 1, 3 or 7: gnx_build_crf)
};

#ifdef GNX_EXPERIMENTS
hipError_t gnx_launch_smooth_crf_mm(const SmoothCRFLaunch& L, hipStream_t s);   // scripts/dev/rejected/k_smooth_crf_mm.hip: products on the float64 MFMA, 16 haplotypes per wave, A <= 16
hipError_t gnx_launch_smooth_crf_quad(const SmoothCRFLaunch& L, hipStream_t s);  // scripts/dev/rejected/k_smooth_crf_quad.hip: four lanes per haplotype, A <= 12
#endif

// ---- cnn smoother (k_smooth_cnn.hip) -----------------------------------------------------------------
struct SmoothCNNLaunch {
  const void* B;          // (N, W, A) float32 or float64 (cast to float32 as torch.tensor(B, dtype=torch.float) does)
  int32_t b_is_f64;
  int64_t N;
  int32_t W, A, S, reserved;
  const float* weight;    // device [A_in][S][AP], AP = gnx_cnn_ap(A)
  const float* bias;      // device (AP,)
  float* proba32;         // optional
  double* proba64;        // optional
  int32_t* labels;        // optional
};

// ---- CovRSK / SVC base (k_base_covrsk.hip) ------------------------------------------------------------
struct SvcWinDev {
  int32_t width, nw, n_sv, g_off;  // window width in SNPs, 32-bit words per plane, support vectors, offset into gtab
  int32_t n_ms, poly;              // number of substring lengths (fast path: prefix of the canonical list); poly = 1:
                                   // polynomial string kernel, run values at coef[rv_off + L], exponent poly_p
  int64_t sv_off;                  // offset (uint32 units) of this window's SV bit-planes: [sv][plane][nw]
  int64_t coef_off;                // offset (doubles): dual (A-1, n_sv) | intercept P | probA P | probB P
  int32_t cls_start[36];           // SV index range per class (prefix sums of n_support)
  int64_t rv_off;
  double poly_p;
};

struct CovRSKDev {
  const SvcWinDev* win = nullptr;
  const uint32_t* svbits = nullptr;
  const double* coef = nullptr;
  const uint32_t* gtab = nullptr;
  int32_t max_nw = 0, max_width = 0;
  std::vector<int32_t> fast_nw;  // host: per window, words of the AND-shift fast path (0 = generic kernel)
};

struct CovRSKLaunch {
  const uint32_t* planes;  // (N, 2, nwp) bit-planes of padded X
  int64_t N, nwp, M;
  int32_t W, A;
  const SvcWinDev* win;
  const uint32_t* svbits;
  const double* coef;
  const uint32_t* gtab;
  int32_t max_nw, max_width;
  float* b32;
  double* b64;
  double* rpair;            // (rpair_haps, W, A(A-1)/2) clipped pairwise probabilities between the two passes
  int64_t rpair_haps;       // haplotypes per chunk of the pairwise buffer
  int64_t n_first, n_count; // chunk being processed (set by the launcher)
  int32_t w_first;          // first window of the grid (set by the launcher)
  const int32_t* host_fast_nw;  // HOST pointer: per-window fast-path word count (launcher only)
  hipStream_t aux;              // optional side stream + two events (launcher only): small window groups overlap the main grid
  hipEvent_t ev_fork, ev_join;
  int32_t n_cu;
};

// ---- gnofix (k_gnofix.hip) ----------------------------------------------------------------------------
struct GnofixLaunch {
  int8_t* X;               // (2*n_ind, ldx) re-phased in place; x_packed: 2-bit rows (gnx_pack_x layout, 4-byte aligned), ldx bytes apart
  int64_t ldx, C;
  int32_t x_packed;
  const double* B;         // (2*n_ind, W, A) base probabilities
  const int32_t* Y0;       // (2*n_ind, W) initial smoother labels
  int32_t* Yout;           // (2*n_ind, W)
  int32_t* n_switches;     // (n_ind,) optional
  int32_t W, A, S, max_it;
  SmoothXGBDev d;
  const int32_t* class_tree0;  // [A+1] tree ranges per class in the packed (class-major) order
  int32_t bp_in_lds;
  float* bp_scratch;       // [n_ind][2][W+2pad][A] when the strips do not fit LDS (k_gnofix_f32 only)
  uint32_t* hist;          // [n_ind][max_it][ceil(W/32)] convergence signatures
  // k_gnofix (rank strips): scratch of the pre- and post-passes and the tile's tree copy
  const uint16_t* R;       // [2*n_ind][W][A] ranks of float32(B)
  const uint32_t* dif;     // [n_ind][ceil(W/32)] "SNP block differs between the two haplotypes"
  uint32_t* par;           // [n_ind][ceil(W/32)] final switch parity per window
  const uint32_t* gf;      // SmoothXGBDev::gf_packed
  const float* proba0;     // (2*n_ind, W, A) probabilities of the initial smoother pass (the labels Y0 come from)
  const float* pmax0;      // [2*n_ind][W] scratch: their row maxima
  const int32_t* order;    // scratch: [n_ind] dispatch order | [W+1] histogram | [n_ind] change counts | [W+1] bucket starts
  int32_t gf_pitch, gf_cap;
};

// ---- calibrator (k_calibrate.hip) -----------------------------------------------------------------------
struct CalibLaunch {
  const void* in;        // (R, A) probabilities, float32 or float64
  int32_t in_is_f64;
  int64_t R;
  int32_t A;
  const int32_t* off;    // device (A+1)
  const double* x;       // device thresholds
  const double* y;
  int32_t thr_f32;       // thresholds were fitted in float32
  double* out64;         // optional (may alias `in` when in_is_f64)
  float* out32;          // optional
  int32_t* labels;       // optional
};

// ---- forest base (k_base_forest.hip) ------------------------------------------------------------------
// Per tree: 2^D node words (heap slot 0 unused; (position << 4) | left-mask over the SNP values 0..3, position = SNP index
// within the window + (window start mod 16): the tile's words are anchored on the padded chromosome coordinate)
// followed by 2^D float leaves.
struct ForestDev {
  const uint8_t* packed = nullptr;          // fb_n_trees * tree_bytes, per window class-major
  const int32_t* win_tree0 = nullptr;       // [W+1]
  const int32_t* win_class_tree0 = nullptr; // [W][A+1] class ranges relative to the window's first tree
  int32_t D = 0, tree_bytes = 0, max_trees = 0, max_words = 0, missing = 2;
  float base_score = 0.5f;
  const double* rf_leafval = nullptr;       // random forest: [tree][2^D heap leaf][A] class-probability rows
  const uint32_t* nodes2 = nullptr;         // k_base_forest2: [tree][2^D] node words baked per window (ring slot byte offset | field | right-mask)
};

struct ForestLaunch {
  const int8_t* X;    // (N, ldx)
  int64_t N, ldx, C, ctx, M, width, width_last;
  int32_t W, A, D, tree_bytes, max_trees, missing;
  int32_t flags;  // development ablation switches (0 in production)
  int32_t w_first, n_windows, wrun, ring;  // set by the launcher: window range of the grid, windows per block, ring slots
  float base_score;
  const uint8_t* packed;
  const int32_t* win_tree0;
  const int32_t* win_class_tree0;
  const double* rf_leafval;  // non-NULL selects the random-forest kernel
  float* b32;
  double* b64;
};

struct gnx_model {
  gnx_ctx* ctx = nullptr;
  gnx_model_info info{};
  std::vector<void*> dev_allocs;
  BaseLRDev lr;
  std::vector<int32_t> lr_h_win_chunk0, lr_h_win_chunk1;
  std::vector<int32_t> lr_h_win_run0, lr_h_win_run1;   // host copies of lr.win_run0/1 (launcher of the 2-bit pass)
  bool lr_i8 = true;
  // what gnx_model_export_prepared needs of the logistic base: the key of the coefficients the planes were made from and their sizes
  uint64_t lr_key = 0;
  int64_t lr_v8_bytes = 0, lr_v2_bytes = 0;   // V8 / V2 (or V2F) as uploaded, without padding
  SmoothXGBDev xgb;
  CovRSKDev svc;
  ForestDev forest;
  // class-major xgboost-schema copy for the rows kernel
  const int32_t* class_tree0 = nullptr;  // device [A+1]
  // CRF
  const double* crf_state = nullptr;  // device (A,A)
  const double* crf_etrans = nullptr; // device (A,A) exp(trans)
  int32_t crf_norm_mask = 0;          // windows between two forward scales - 1, from the weights' range (gnx_build_crf)
  const float* cnn_weight = nullptr;  // device [A_in][S][AP] (transposed from torch's (out, in, k) at load), AP = gnx_cnn_ap(A)
  const float* cnn_bias = nullptr;    // device (AP,)
  // calibrator
  const int32_t* calib_off = nullptr;
  const double* calib_x = nullptr;
  const double* calib_y = nullptr;
  bool calibrate_on = false;
  bool calib_f32 = false;
};

// shared with gnx_api_vcf.hip (defined in gnx_api.hip)
int gnx_fail(gnx_ctx* ctx, int code, const std::string& msg);

// The HIP "current device" is per host THREAD (a new thread starts on device 0) and hipMalloc / kernel launches / event creation use
// it, not the stream's device.  Every extern "C" entry that touches the GPU therefore binds its context's device for the duration of
// the call and puts the caller's device back on return (one gnx_ctx per device, different contexts on different threads —
// or, with torch tensors, several contexts driven from ONE thread).
struct gnx_device_scope {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit gnx_device_scope(const gnx_ctx* ctx) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != ctx->device) {
      err = hipSetDevice(ctx->device);
      switched = err == hipSuccess;
    }
  }
  ~gnx_device_scope() {
    if (switched) (void)hipSetDevice(prev);
  }
  gnx_device_scope(const gnx_device_scope&) = delete;
  gnx_device_scope& operator=(const gnx_device_scope&) = delete;
};
#define GNX_BIND_DEVICE(ctx)                                                                                       \
  gnx_device_scope dev_scope__(ctx);                                                                               \
  if (dev_scope__.err != hipSuccess)                                                                               \
  return gnx_fail((ctx), GNX_EHIP, std::string("binding device ") + std::to_string((ctx)->device) + ": " + hipGetErrorString(dev_scope__.err))
int gnx_ws_reserve(gnx_ctx* ctx, gnx_devbuf& b, size_t bytes);
int gnx_pipe_init(gnx_ctx* ctx);
bool gnx_gnofix_packed_ok(const gnx_model* m);  // gnx_gnofix_packed_dev can run (rank-strip Gnofix kernel)
bool gnx_lr_p2_usable(const gnx_model* m);  // packed (2-bit) rows go straight into the logistic pass (k_base_logistic_p2)
void* gnx_pin_alloc(size_t bytes);  // page-locked host memory through the process-wide reuse list (gnx_api.hip); NULL on failure
void gnx_pin_free(void* p);

// model preparation (gnx_model_build.hip), called by gnx_model_load
int gnx_build_lr(gnx_model* m, const gnx_model_desc* d);
int gnx_build_covrsk(gnx_model* m, const gnx_model_desc* d);
int gnx_build_forest(gnx_model* m, const gnx_model_desc* d);
int gnx_build_rforest(gnx_model* m, const gnx_model_desc* d);
int gnx_build_xgb(gnx_model* m, const gnx_model_desc* d);
int gnx_build_crf(gnx_model* m, const gnx_model_desc* d);

// a host vector as a device array owned by the model (freed with it), followed by pad_bytes + 16 zero bytes
template <typename T>
inline int gnx_dev_upload(gnx_model* m, const std::vector<T>& h, const T** out, size_t pad_bytes = 0) {
  gnx_ctx* ctx = m->ctx;
  void* p = nullptr;
  const size_t bytes = h.size() * sizeof(T);
  hipError_t e = hipMalloc(&p, bytes + pad_bytes + 16);
  if (e != hipSuccess) return gnx_fail(ctx, GNX_ENOMEM, std::string("hipMalloc model: ") + hipGetErrorString(e));
  m->dev_allocs.push_back(p);
  m->info.device_bytes += (int64_t)(bytes + pad_bytes + 16);
  if (bytes && (e = hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice)) != hipSuccess) return gnx_fail(ctx, GNX_EHIP, std::string("hipMemcpy model: ") + hipGetErrorString(e));
  if ((e = hipMemset((char*)p + bytes, 0, pad_bytes + 16)) != hipSuccess) return gnx_fail(ctx, GNX_EHIP, std::string("hipMemset model: ") + hipGetErrorString(e));
  *out = (const T*)p;
  return GNX_OK;
}


// kernel launchers (defined in the .hip files)
hipError_t gnx_launch_gt2_to_x(const uint8_t* G, int64_t V, int64_t ldg, int64_t n0, int64_t N, const int32_t* src, int64_t C,
                               int8_t* X, int64_t ldx, hipStream_t s);
hipError_t gnx_launch_gt2_to_p2(const uint8_t* G, int64_t V, int64_t ldg, int64_t n0, int64_t N, const int32_t* src, int64_t C,
                                uint8_t* P, int64_t ldp, hipStream_t s);
hipError_t gnx_launch_p2_to_gt2(const uint8_t* P, int64_t N, int64_t ldp, int64_t n0, const int32_t* cols, int64_t V, uint8_t* G,
                                int64_t ldg, hipStream_t s);
hipError_t gnx_launch_x_to_gt2(const int8_t* X, int64_t N, int64_t ldx, int64_t n0, const int32_t* cols, int64_t V, uint8_t* G,
                               int64_t ldg, hipStream_t s);
hipError_t gnx_launch_base_logistic(const BaseLRLaunch& L, int n_cu, hipStream_t s);
hipError_t gnx_launch_base_logistic_i8(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s);
hipError_t gnx_launch_base_logistic_i8_dl(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s);
// 2-bit rows: L.X = packed matrix (gnx_pack_x layout), L.ldx = its row stride in bytes, L.last_row = zero-padded copy of packed row
// N-1, L.h_win_chunk0/1 = HOST copies of d.win_run0/1; hipErrorNotSupported when no instantiation fits
hipError_t gnx_launch_base_logistic_p2(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s);
#ifdef GNX_EXPERIMENTS  // scripts/dev/rejected/ (make EXPERIMENTS=1)
hipError_t gnx_launch_base_logistic_i8_w512(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s);
hipError_t gnx_launch_base_logistic_i8_ws(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s);
hipError_t gnx_launch_base_logistic_i8_fl(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s);
hipError_t gnx_launch_smooth_xgb_h64(const SmoothXGBLaunch& L, uint16_t* Rk, const gnx_tune& tune, hipStream_t s);
int gnx_smooth_h64_waves(const SmoothXGBDev& d, int A, int S);            // 0: the strip does not fit the LDS
size_t gnx_smooth_h64_rank_bytes(int64_t N, int W, int A, int S);
#endif
hipError_t gnx_launch_fb_len(const float* d_proba, int64_t N, int64_t W, int A, uint8_t* d_len, unsigned long long* d_line_len, hipStream_t s);
hipError_t gnx_launch_fb_emit(const float* d_proba, int64_t N, int64_t W, int A, const uint8_t* d_len, const char* d_pb, const int64_t* d_po,
                              const int64_t* d_line_off, char* d_body, hipStream_t s);
hipError_t gnx_launch_smooth_xgb(const SmoothXGBLaunch& L, const gnx_tune& tune, hipStream_t s);
hipError_t gnx_launch_smooth_xgb_rk(const SmoothXGBLaunch& L, const gnx_tune& tune, hipStream_t s);
#ifdef GNX_EXPERIMENTS
// bit-sliced tree smoother (scripts/dev/rejected/k_smooth_xgb_bs.hip, GNX_SMOOTH_IMPL=bs); bins = gnx_smooth_bs_scratch_bytes() of
// device scratch (the rank pre-pass's output)
hipError_t gnx_launch_smooth_xgb_bs(const SmoothXGBLaunch& L, uint16_t* bins, int n_cu, const gnx_tune& tune, hipStream_t s);
size_t gnx_smooth_bs_scratch_bytes(int64_t N, int W, int A);
bool gnx_smooth_bs_fits(const SmoothXGBDev& d, int A, int S);
#endif
hipError_t gnx_launch_smooth_rows(const SmoothXGBDev& d, const float* rows, int64_t R, int32_t F, int32_t A,
                                  float* proba, hipStream_t s);
hipError_t gnx_launch_pack_bits(const int8_t* X, int64_t N, int64_t ldx, int64_t C, int64_t ctx, int64_t nwp,
                                uint32_t* planes, hipStream_t s);
hipError_t gnx_launch_covrsk(const CovRSKLaunch& L, hipStream_t s);
size_t gnx_covrsk_lds_bytes(int A, int max_nw, int max_width);
hipError_t gnx_launch_gnofix_prep(const GnofixLaunch& L, int64_t n_ind, hipStream_t s);
hipError_t gnx_launch_gnofix(const GnofixLaunch& L, int64_t n_ind, int threads, hipStream_t s);
size_t gnx_gnofix_lds_bytes(int W, int A, int S, int pitch, int cap, int D, int threads, int n_trees);
int gnx_gnofix_cap(int max_class_trees, int D, int S, int threads);
hipError_t gnx_launch_gnofix_f32(const GnofixLaunch& L, int64_t n_ind, hipStream_t s);
size_t gnx_gnofix_f32_lds_bytes(int W, int A, int S, int n_trees, int tree_bytes, bool bp_in_lds);
hipError_t gnx_launch_calibrate(const CalibLaunch& L, hipStream_t s);
hipError_t gnx_train_lr_run(const int8_t* dX, int64_t N, int64_t ldx, const int32_t* dY, int64_t C, int64_t M, int64_t ctx, int A,
                            double Creg, double tol, int max_newton, int max_cg, double* h_coef, int64_t ldc, double* h_icpt,
                            gnx_train_info* info, hipStream_t st);
hipError_t gnx_launch_unpack2(const uint8_t* P, int64_t N, int64_t ldp, int64_t C, int8_t* X, int64_t ldx, hipStream_t s);
hipError_t gnx_launch_smooth_crf(const SmoothCRFLaunch& L, const gnx_tune& tune, hipStream_t s);
hipError_t gnx_train_gbt_run(const void* dB, int b_is_f64, const int32_t* dy, int64_t N, int32_t W, int32_t A, int32_t S,
                             const gnx_gbt_params& P, int32_t* tree_off, int32_t* tree_class, int32_t* left, int32_t* right,
                             int32_t* feat, float* cond, int64_t* n_nodes_out, double* loss_out, int n_cu, hipStream_t s);
inline int gnx_cnn_ap(int A) { return A <= 8 ? 8 : A <= 16 ? 16 : 32; }  // output channels padded to the kernel's template width
hipError_t gnx_launch_smooth_cnn(const SmoothCNNLaunch& L, hipStream_t s);
hipError_t gnx_launch_base_forest(const ForestLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s);
size_t gnx_forest_lds_bytes(int A, int ring_words, int max_trees, int tree_bytes, int threads);
hipError_t gnx_launch_base_forest2(const ForestLaunch& L, const uint32_t* nodes2, int n_cu, const gnx_tune& tune, hipStream_t s,
                                   hipStream_t aux, hipEvent_t ev_fork, hipEvent_t ev_join);
size_t gnx_forest2_lds_bytes(int A, int ring_words, int max_trees, int D, bool rf = false);
uint32_t gnx_forest2_node(uint32_t loader_word, uint32_t g0, uint32_t ring);
int gnx_forest_ring_words(int64_t width);
size_t gnx_smooth_xgb_lds_bytes(const SmoothXGBDev& d, int A, int S);
