// gnx_api_vcf.hip — C ABI of the file path (include/gnomix_io.h) where it needs a context: page-locked genotype matrices,
// gt2 <-> X on the device, and the host pipelines  parsed VCF -> outputs  that replace gnomix.py:48-72
// (read_vcf / vcf_to_npy / base.predict_proba / smooth.predict_proba, or model.phase + model.predict_proba).
#include <cstdlib>
#include <cstring>

#include "gnx_internal.h"
#include "gnx_io.h"

#define HIPCHK(ctx, expr)                                                                          \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return gnx_fail((ctx), GNX_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__));        \
  } while (0)

// ---- reader ------------------------------------------------------------------------------------------------------------
static void* alloc_plain(void*, size_t bytes) { return malloc(bytes); }
static void free_plain(void*, void* p) { free(p); }
static void* alloc_pinned(void* user, size_t bytes) {
  gnx_ctx* ctx = (gnx_ctx*)user;
  gnx_device_scope bind(ctx);  // (the reader's parsing threads call this: each binds, allocates, unbinds)
  if (bind.err != hipSuccess) return nullptr;
  return gnx_pin_alloc(bytes);  // (released buffers of the same size are reused: gnx_api.hip)
}
static void free_pinned(void*, void* p) { gnx_pin_free(p); }

int gnx_vcf_read(gnx_ctx* ctx, const char* path, const char* region, int n_threads, gnx_vcf** out) {
  if (ctx && ctx->usable) {
    const int rc = gnx_io_vcf_read(path, region, n_threads, alloc_pinned, free_pinned, ctx, 1, out);
    if (rc != GNX_OK) ctx->err = gnx_io_last_error();
    return rc;
  }
  return gnx_io_vcf_read(path, region, n_threads, alloc_plain, free_plain, nullptr, 0, out);
}

// ---- device passes -----------------------------------------------------------------------------------------------------
int gnx_gt2_to_x_dev(gnx_ctx* ctx, const uint8_t* dG, int64_t V, int64_t ldg, int64_t n0, int64_t N, const int32_t* d_src, int64_t C,
                     int8_t* dX, int64_t ldx) {
  if (!ctx) return GNX_EINVAL;
  if (N < 0 || C < 0 || V < 0 || n0 < 0 || (n0 & 3) || ldx < C || ldg < (n0 + N + 3) / 4 || (N > 0 && C > 0 && (!dG || !d_src || !dX)))
    return gnx_fail(ctx, GNX_EINVAL, "gt2_to_x: bad arguments (n0 must be a multiple of 4, ldg >= ceil((n0 + N) / 4), ldx >= C)");
  GNX_BIND_DEVICE(ctx);
  HIPCHK(ctx, gnx_launch_gt2_to_x(dG, V, ldg, n0, N, d_src, C, dX, ldx, ctx->stream));
  return GNX_OK;
}

int gnx_gt2_to_p2_dev(gnx_ctx* ctx, const uint8_t* dG, int64_t V, int64_t ldg, int64_t n0, int64_t N, const int32_t* d_src, int64_t C,
                      uint8_t* dP, int64_t ldp) {
  if (!ctx) return GNX_EINVAL;
  if (N < 0 || C < 0 || V < 0 || n0 < 0 || (n0 & 3) || ldp < (C + 3) / 4 || ldg < (n0 + N + 3) / 4 || (N > 0 && C > 0 && (!dG || !d_src || !dP)))
    return gnx_fail(ctx, GNX_EINVAL, "gt2_to_p2: bad arguments (n0 must be a multiple of 4, ldg >= ceil((n0 + N) / 4), ldp >= ceil(C / 4))");
  GNX_BIND_DEVICE(ctx);
  HIPCHK(ctx, gnx_launch_gt2_to_p2(dG, V, ldg, n0, N, d_src, C, dP, ldp, ctx->stream));
  return GNX_OK;
}

int gnx_x_to_gt2_dev(gnx_ctx* ctx, const int8_t* dX, int64_t N, int64_t ldx, int64_t n0, const int32_t* d_cols, int64_t V, uint8_t* dG,
                     int64_t ldg) {
  if (!ctx) return GNX_EINVAL;
  if (N < 0 || V < 0 || n0 < 0 || (n0 & 3) || ldg < (n0 + N + 3) / 4 || (N > 0 && V > 0 && (!dX || !d_cols || !dG)))
    return gnx_fail(ctx, GNX_EINVAL, "x_to_gt2: bad arguments (n0 must be a multiple of 4, ldg >= ceil((n0 + N) / 4))");
  GNX_BIND_DEVICE(ctx);
  HIPCHK(ctx, gnx_launch_x_to_gt2(dX, N, ldx, n0, d_cols, V, dG, ldg, ctx->stream));
  return GNX_OK;
}

// ---- host pipelines ------------------------------------------------------------------------------------------------------
namespace {
struct Gt2Job {
  int64_t C, ldx, nb;       // model SNPs, device row pitch of X, haplotypes per batch
  int64_t ldg_d;            // row pitch of the device copy of G (only the bytes of this call's haplotype range)
  const uint8_t* dG;
  const int32_t* dsrc;
  int8_t* dX;
};

int check_src(gnx_ctx* ctx, const int32_t* src, int64_t C, int64_t V) {
  for (int64_t c = 0; c < C; ++c) {
    const int32_t s = src[c];
    if (s == -1) continue;
    if (s < 0 || (int64_t)(s & 0x3FFFFFFF) >= V) return gnx_fail(ctx, GNX_EINVAL, "gt2: column map entry outside the variant rows");
  }
  return GNX_OK;
}

// uploads the haplotypes [h0, h0 + N) of G (h0 a multiple of 4: whole bytes of every variant row, one strided copy) and the column
// map, sizes X for one batch; everything on the context stream
int gt2_stage(gnx_model* m, const uint8_t* G, int64_t V, int64_t ldg, int64_t h0, int64_t N, const int32_t* src, const int32_t* extra_cols,
              int64_t n_extra, Gt2Job* J) {
  gnx_ctx* ctx = m->ctx;
  const int64_t C = m->info.C;
  int rc;
  if ((rc = check_src(ctx, src, C, V)) != GNX_OK) return rc;
  J->C = C;
  J->ldx = (C + 63) / 64 * 64;
  // a batch: <= ~4 GiB of X (HBM is 288 GB; fewer, larger launches), whole 1024-haplotype tiles of the transpose
  int64_t nb = (((int64_t)4 << 30) / J->ldx) / 1024 * 1024;
  if (ctx->tune.host_batch > 0) nb = (ctx->tune.host_batch + 3) / 4 * 4;  // tests: several batches on small inputs
  nb = std::max<int64_t>(4, nb);
  J->nb = std::min(nb, (N + 3) / 4 * 4);
  const int64_t wbytes = (N + 3) / 4;
  J->ldg_d = (h0 == 0 && wbytes == ldg) ? ldg : (wbytes + 63) / 64 * 64;
  if ((rc = gnx_ws_reserve(ctx, ctx->ws_gt2, (size_t)V * J->ldg_d + 64)) != GNX_OK) return rc;
  if ((rc = gnx_ws_reserve(ctx, ctx->ws_src, (size_t)(C + n_extra) * 4 + 64)) != GNX_OK) return rc;
  if ((rc = gnx_ws_reserve(ctx, ctx->ws_xu, (size_t)J->nb * J->ldx + 256)) != GNX_OK) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ctx->ws_src.p, src, (size_t)C * 4, hipMemcpyHostToDevice, ctx->stream));
  if (n_extra > 0)
    HIPCHK(ctx, hipMemcpyAsync((int32_t*)ctx->ws_src.p + C, extra_cols, (size_t)n_extra * 4, hipMemcpyHostToDevice, ctx->stream));
  if (V > 0) {
    if (J->ldg_d == ldg) HIPCHK(ctx, hipMemcpyAsync(ctx->ws_gt2.p, G, (size_t)V * ldg, hipMemcpyHostToDevice, ctx->stream));
    else HIPCHK(ctx, hipMemcpy2DAsync(ctx->ws_gt2.p, (size_t)J->ldg_d, G + h0 / 4, (size_t)ldg, (size_t)wbytes, (size_t)V, hipMemcpyHostToDevice, ctx->stream));
  }
  J->dG = (const uint8_t*)ctx->ws_gt2.p;
  J->dsrc = (const int32_t*)ctx->ws_src.p;
  J->dX = (int8_t*)ctx->ws_xu.p;
  return GNX_OK;
}
}  // namespace

int gnx_infer_gt2(gnx_model* m, const uint8_t* G, int64_t V, int64_t ldg, int64_t N, const int32_t* src, float* p32, double* p64,
                  int32_t* lab) {
  return gnx_infer_gt2_range(m, G, V, ldg, 0, N, src, p32, p64, lab);
}

int gnx_infer_gt2_range(gnx_model* m, const uint8_t* G, int64_t V, int64_t ldg, int64_t h0, int64_t N, const int32_t* src, float* p32,
                        double* p64, int32_t* lab) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  if (N < 0 || V < 0 || h0 < 0 || (h0 & 3) || ldg < (h0 + N + 3) / 4 || !src || (N > 0 && V > 0 && !G))
    return gnx_fail(ctx, GNX_EINVAL, "infer_gt2: bad G / V / ldg / first haplotype (a multiple of 4) / N / src");
  if (N == 0) return GNX_OK;
  GNX_BIND_DEVICE(ctx);
  Gt2Job J{};
  int rc = gt2_stage(m, G, V, ldg, h0, N, src, nullptr, 0, &J);
  if (rc != GNX_OK) return rc;
  const size_t WA = (size_t)m->info.W * m->info.A, Wn = (size_t)m->info.W;
  const int64_t nb = J.nb, n_batches = (N + nb - 1) / nb;
  const int nbuf = (ctx->tune.h2d_overlap != 0 && n_batches > 1) ? 2 : 1;
  if (nbuf == 2 && (rc = gnx_pipe_init(ctx)) != GNX_OK) return rc;
  const size_t p32_b = (nb * WA * 4 + 255) & ~(size_t)255, p64_b = (nb * WA * 8 + 255) & ~(size_t)255, lab_b = (nb * Wn * 4 + 255) & ~(size_t)255;
  if ((rc = gnx_ws_reserve(ctx, ctx->ws_p32, p32_b * nbuf)) != GNX_OK) return rc;
  if (p64 && (rc = gnx_ws_reserve(ctx, ctx->ws_p64, p64_b * nbuf)) != GNX_OK) return rc;
  if (lab && (rc = gnx_ws_reserve(ctx, ctx->ws_lab, lab_b * nbuf)) != GNX_OK) return rc;
  {
    const bool f64 = (m->info.smooth_kind == GNX_SMOOTH_CRF);
    if ((rc = gnx_ws_reserve(ctx, f64 ? ctx->ws_b64 : ctx->ws_b32, nb * WA * (f64 ? 8 : 4))) != GNX_OK) return rc;
  }
  hipStream_t sc = ctx->stream, so = nbuf == 2 ? ctx->s_out : ctx->stream;
  const bool p2 = gnx_lr_p2_usable(m);
  const int64_t ldp2 = (J.C + 255) / 256 * 64;  // 64-byte aligned packed rows (<= J.ldx / 4: the batch fits the int8 workspace)
  for (int64_t i = 0; i < n_batches; ++i) {
    const int b = (int)(i % nbuf);
    const int64_t n0 = i * nb, n = std::min(nb, N - n0);
    if (nbuf == 2 && i >= 2) HIPCHK(ctx, hipStreamWaitEvent(sc, ctx->ev_out[b], 0));  // outputs of batch i-2 have left this half
    float* dp32 = (float*)((char*)ctx->ws_p32.p + (size_t)b * p32_b);
    double* dp64 = p64 ? (double*)((char*)ctx->ws_p64.p + (size_t)b * p64_b) : nullptr;
    int32_t* dlab = lab ? (int32_t*)((char*)ctx->ws_lab.p + (size_t)b * lab_b) : nullptr;
    if (p2) {  // the haplotype rows stay 2-bit: transposed as they are, read packed by the logistic pass
      HIPCHK(ctx, gnx_launch_gt2_to_p2(J.dG, V, J.ldg_d, n0, n, J.dsrc, J.C, (uint8_t*)J.dX, ldp2, sc));
      if ((rc = gnx_infer_packed_dev(m, (const uint8_t*)J.dX, n, ldp2, dp32, dp64, dlab)) != GNX_OK) return rc;
    } else {
      HIPCHK(ctx, gnx_launch_gt2_to_x(J.dG, V, J.ldg_d, n0, n, J.dsrc, J.C, J.dX, J.ldx, sc));
      if ((rc = gnx_infer_dev(m, J.dX, n, J.ldx, dp32, dp64, dlab)) != GNX_OK) return rc;
    }
    if (nbuf == 2) {
      HIPCHK(ctx, hipEventRecord(ctx->ev_done[b], sc));
      HIPCHK(ctx, hipStreamWaitEvent(so, ctx->ev_done[b], 0));
    }
    if (p32) HIPCHK(ctx, hipMemcpyAsync(p32 + n0 * WA, dp32, n * WA * 4, hipMemcpyDeviceToHost, so));
    if (p64) HIPCHK(ctx, hipMemcpyAsync(p64 + n0 * WA, dp64, n * WA * 8, hipMemcpyDeviceToHost, so));
    if (lab) HIPCHK(ctx, hipMemcpyAsync(lab + n0 * Wn, dlab, n * Wn * 4, hipMemcpyDeviceToHost, so));
    if (nbuf == 2) HIPCHK(ctx, hipEventRecord(ctx->ev_out[b], so));
  }
  if (nbuf == 2) HIPCHK(ctx, hipStreamSynchronize(so));
  HIPCHK(ctx, hipStreamSynchronize(sc));
  return GNX_OK;
}

int gnx_phase_gt2(gnx_model* m, const uint8_t* G, int64_t V, int64_t ldg, int64_t N, const int32_t* src, int32_t max_it,
                  const int32_t* out_cols, int64_t n_out, uint8_t* G_out, int64_t ldg_out, float* p32, double* p64, int32_t* lab,
                  int32_t* n_switches) {
  return gnx_phase_gt2_range(m, G, V, ldg, 0, N, src, max_it, out_cols, n_out, G_out, ldg_out, p32, p64, lab, n_switches);
}

int gnx_phase_gt2_range(gnx_model* m, const uint8_t* G, int64_t V, int64_t ldg, int64_t h0, int64_t N, const int32_t* src, int32_t max_it,
                        const int32_t* out_cols, int64_t n_out, uint8_t* G_out, int64_t ldg_out, float* p32, double* p64, int32_t* lab,
                        int32_t* n_switches) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  if (N < 0 || (N & 1) || V < 0 || h0 < 0 || (h0 & 3) || ldg < (h0 + N + 3) / 4 || !src || (N > 0 && V > 0 && !G))
    return gnx_fail(ctx, GNX_EINVAL, "phase_gt2: bad G / V / ldg / first haplotype (a multiple of 4) / N / src (N = 2 * individuals)");
  if (n_out < 0 || (G_out && (ldg_out < (h0 + N + 3) / 4 || (n_out > 0 && !out_cols)))) return gnx_fail(ctx, GNX_EINVAL, "phase_gt2: bad output rows");
  if (m->info.smooth_kind != GNX_SMOOTH_XGB) return gnx_fail(ctx, GNX_ESTATE, "Type of Smoother does not currently support re-phasing");
  if (N == 0) return GNX_OK;
  if (!G_out) n_out = 0;
  for (int64_t r = 0; r < n_out; ++r)
    if (out_cols[r] < 0 || out_cols[r] >= m->info.C) return gnx_fail(ctx, GNX_EINVAL, "phase_gt2: output column outside the model's SNPs");
  GNX_BIND_DEVICE(ctx);
  Gt2Job J{};
  int rc = gt2_stage(m, G, V, ldg, h0, N, src, out_cols, n_out, &J);
  if (rc != GNX_OK) return rc;
  const size_t WA = (size_t)m->info.W * m->info.A, Wn = (size_t)m->info.W;
  // one workgroup per individual: batches of whole individuals; smaller than the inference batch (B is float64 here)
  int64_t nb = std::min<int64_t>(J.nb, 65536);
  nb = std::max<int64_t>(4, nb / 4 * 4);
  const int64_t n_batches = (N + nb - 1) / nb;
  if ((rc = gnx_ws_reserve(ctx, ctx->ws_b64, nb * WA * 8)) != GNX_OK) return rc;
  if ((rc = gnx_ws_reserve(ctx, ctx->ws_p32, nb * WA * 4)) != GNX_OK) return rc;
  if (p64 && (rc = gnx_ws_reserve(ctx, ctx->ws_p64, nb * WA * 8)) != GNX_OK) return rc;
  if ((rc = gnx_ws_reserve(ctx, ctx->ws_lab, nb * Wn * 4 + (size_t)nb * 2 + 64)) != GNX_OK) return rc;
  if ((rc = gnx_ws_reserve(ctx, ctx->ws_b32, nb * WA * 4)) != GNX_OK) return rc;
  // the phased rows of THIS range: a device matrix of its own pitch, copied into the caller's rows at byte h0 / 4 (the other bytes of
  // G_out belong to other ranges — other contexts may be writing them — and are not touched)
  const int64_t obytes = (N + 3) / 4, ldo_d = (obytes + 63) / 64 * 64;
  if (n_out > 0) {
    if ((rc = gnx_ws_reserve(ctx, ctx->ws_gt2o, (size_t)n_out * ldo_d + 64)) != GNX_OK) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->ws_gt2o.p, 0, (size_t)n_out * ldo_d, ctx->stream));
  }
  hipStream_t sc = ctx->stream;
  // the haplotype rows stay 2-bit end to end where the model has the 2-bit logistic pass and the rank-strip Gnofix kernel:
  // gt2 -> packed rows -> base -> Gnofix (swaps 2-bit SNP blocks) -> gt2 rows of the phased VCF, and the final predict_proba
  const bool p2 = gnx_lr_p2_usable(m) && gnx_gnofix_packed_ok(m);
  const int64_t ldp2 = (J.C + 255) / 256 * 64;
  for (int64_t i = 0; i < n_batches; ++i) {
    const int64_t n0 = i * nb, n = std::min(nb, N - n0);
    int32_t* dY = (int32_t*)ctx->ws_lab.p;
    int32_t* dNs = dY + (size_t)nb * Wn;
    if (p2) {
      uint8_t* dP = (uint8_t*)J.dX;
      HIPCHK(ctx, gnx_launch_gt2_to_p2(J.dG, V, J.ldg_d, n0, n, J.dsrc, J.C, dP, ldp2, sc));
      if ((rc = gnx_base_predict_packed_dev(m, dP, n, ldp2, nullptr, (double*)ctx->ws_b64.p)) != GNX_OK) return rc;
      if ((rc = gnx_gnofix_packed_dev(m, dP, ldp2, (const double*)ctx->ws_b64.p, n / 2, max_it, dY, dNs)) != GNX_OK) return rc;
      if (n_out > 0)
        HIPCHK(ctx, gnx_launch_p2_to_gt2(dP, n, ldp2, n0, J.dsrc + J.C, n_out, (uint8_t*)ctx->ws_gt2o.p, ldo_d, sc));
    } else {
      HIPCHK(ctx, gnx_launch_gt2_to_x(J.dG, V, J.ldg_d, n0, n, J.dsrc, J.C, J.dX, J.ldx, sc));
      if ((rc = gnx_base_predict_dev(m, J.dX, n, J.ldx, nullptr, (double*)ctx->ws_b64.p)) != GNX_OK) return rc;
      if ((rc = gnx_gnofix_dev(m, J.dX, J.ldx, (const double*)ctx->ws_b64.p, n / 2, max_it, dY, dNs)) != GNX_OK) return rc;
      if (n_out > 0)
        HIPCHK(ctx, gnx_launch_x_to_gt2(J.dX, n, J.ldx, n0, J.dsrc + J.C, n_out, (uint8_t*)ctx->ws_gt2o.p, ldo_d, sc));
    }
    if (lab) HIPCHK(ctx, hipMemcpyAsync(lab + n0 * Wn, dY, n * Wn * 4, hipMemcpyDeviceToHost, sc));
    if (n_switches) HIPCHK(ctx, hipMemcpyAsync(n_switches + n0 / 2, dNs, (size_t)(n / 2) * 4, hipMemcpyDeviceToHost, sc));
    if (p32 || p64) {
      // model.predict_proba(X_phased) (gnomix.py:72): base + smoother again on the re-phased haplotypes
      float* q32 = (float*)ctx->ws_p32.p;
      double* q64 = p64 ? (double*)ctx->ws_p64.p : nullptr;
      rc = p2 ? gnx_infer_packed_dev(m, (const uint8_t*)J.dX, n, ldp2, q32, q64, nullptr) : gnx_infer_dev(m, J.dX, n, J.ldx, q32, q64, nullptr);
      if (rc != GNX_OK) return rc;
      if (p32) HIPCHK(ctx, hipMemcpyAsync(p32 + n0 * WA, ctx->ws_p32.p, n * WA * 4, hipMemcpyDeviceToHost, sc));
      if (p64) HIPCHK(ctx, hipMemcpyAsync(p64 + n0 * WA, ctx->ws_p64.p, n * WA * 8, hipMemcpyDeviceToHost, sc));
    }
  }
  if (n_out > 0)
    HIPCHK(ctx, hipMemcpy2DAsync(G_out + h0 / 4, (size_t)ldg_out, ctx->ws_gt2o.p, (size_t)ldo_d, (size_t)obytes, (size_t)n_out, hipMemcpyDeviceToHost, sc));
  HIPCHK(ctx, hipStreamSynchronize(sc));
  return GNX_OK;
}

// ---- <prefix>.fb with the number text produced on the GPU (k_fb_text.hip) ------------------------------------------------------------
// proba (N, W, A) float32 on the HOST (where gnx_infer_gt2 put it): back to HBM (2 ms for chr22 x 10 000 haplotypes), lengths, line
// offsets (a scan over W numbers on the host), text, one page-locked buffer back, one write().  The body is produced exactly — the same
// bytes gnx_write_fb writes.
extern "C" int gnx_write_fb_dev(gnx_ctx* ctx, const char* path, const char* head, int64_t head_len, const char* pb, const int64_t* po,
                                const float* proba, int64_t N, int64_t W, int64_t A) {
  if (!ctx) return GNX_EINVAL;
  if (!ctx->usable) return gnx_fail(ctx, GNX_ESTATE, "write_fb_dev: context has no device");
  if (!path || head_len < 0 || (head_len > 0 && !head) || N < 0 || W < 0 || A < 0 || (W > 0 && (!pb || !po)) || (N > 0 && W > 0 && A > 0 && !proba))
    return gnx_fail(ctx, GNX_EINVAL, "write_fb_dev: bad arguments");
  if (N == 0 || W == 0 || A == 0 || A > 4096) return gnx_write_fb(path, head, head_len, pb, po, proba, 0, N, W, A, 0);
  GNX_BIND_DEVICE(ctx);
  hipStream_t s = ctx->stream;
  const int64_t NA = N * A, plen_all = po[W] - po[0];
  if (po[0] != 0 || plen_all < 0) return gnx_fail(ctx, GNX_EINVAL, "write_fb_dev: prefix offsets must start at 0 and not decrease");
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_p = take((size_t)N * W * A * 4), o_len = take((size_t)W * NA), o_ll = take((size_t)W * 8), o_pb = take((size_t)plen_all + 1),
               o_po = take((size_t)(W + 1) * 8), o_lo = take((size_t)W * 8);
  int rc = gnx_ws_reserve(ctx, ctx->ws_fb, off);
  if (rc != GNX_OK) return rc;
  uint8_t* base = (uint8_t*)ctx->ws_fb.p;
  HIPCHK(ctx, hipMemcpyAsync(base + o_p, proba, (size_t)N * W * A * 4, hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemcpyAsync(base + o_pb, pb, (size_t)plen_all, hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemcpyAsync(base + o_po, po, (size_t)(W + 1) * 8, hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemsetAsync(base + o_ll, 0, (size_t)W * 8, s));
  HIPCHK(ctx, gnx_launch_fb_len((const float*)(base + o_p), N, W, (int)A, base + o_len, (unsigned long long*)(base + o_ll), s));
  std::vector<unsigned long long> line_len((size_t)W);
  HIPCHK(ctx, hipMemcpyAsync(line_len.data(), base + o_ll, (size_t)W * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(ctx, hipStreamSynchronize(s));
  std::vector<int64_t> line_off((size_t)W);
  int64_t total = 0;
  for (int64_t w = 0; w < W; ++w) {
    line_off[(size_t)w] = total;
    total += (po[w + 1] - po[w]) + (int64_t)line_len[(size_t)w] + 1;
  }
  HIPCHK(ctx, hipMemcpyAsync(base + o_lo, line_off.data(), (size_t)W * 8, hipMemcpyHostToDevice, s));
  if ((rc = gnx_ws_reserve(ctx, ctx->ws_fb_body, (size_t)total + 64)) != GNX_OK) return rc;
  HIPCHK(ctx, gnx_launch_fb_emit((const float*)(base + o_p), N, W, (int)A, base + o_len, (const char*)(base + o_pb), (const int64_t*)(base + o_po),
                                 (const int64_t*)(base + o_lo), (char*)ctx->ws_fb_body.p, s));
  char* host = (char*)gnx_pin_alloc((size_t)total + 64);
  if (!host) return gnx_fail(ctx, GNX_ENOMEM, "write_fb_dev: page-locked buffer for the text");
  hipError_t e = hipMemcpyAsync(host, ctx->ws_fb_body.p, (size_t)total, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    gnx_pin_free(host);
    return gnx_fail(ctx, GNX_EHIP, std::string("write_fb_dev: ") + hipGetErrorString(e));
  }
  rc = gnx_io_write_file(path, head, (size_t)head_len, host, (size_t)total);
  gnx_pin_free(host);
  if (rc != GNX_OK) return gnx_fail(ctx, rc, gnx_io_last_error());
  return GNX_OK;
}
