/* abi_smoke.c — libgnomix_hip.so from plain C: no Python, no torch, only include/gnomix_hip.h.
 *
 *   gcc -Iinclude examples/abi_smoke.c -Lgnomix_amd -lgnomix_hip -Wl,-rpath,$PWD/gnomix_amd -lm -o abi_smoke && ./abi_smoke
 *
 * Builds a small logistic-base + CRF-smoother model from a fixed linear congruential generator (tests/test_c_abi.py builds
 * the identical model through the Python layer and compares), runs gnx_infer on host buffers and prints one checksum line:
 *     labels <sum of (i+1)*label_i mod 2^31> proba <sum of probabilities, 6 decimals>
 * Exit status: 0 on success, 2 when the library reports no usable GPU (the message goes to stderr), 1 on any other error. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gnomix_hip.h"

static uint32_t lcg_state = 12345u;
static double lcg(void) { /* uniform in [0, 1) */
  lcg_state = lcg_state * 1664525u + 1013904223u;
  return (double)(lcg_state >> 8) / 16777216.0;
}

int main(void) {
  enum { C = 1237, M = 50, A = 4, CTX = 25, N = 24 };
  const int W = C / M, rem = C - M * W, ldc = M + 2 * CTX + rem;
  double* coef = (double*)malloc(sizeof(double) * W * A * ldc);
  double* icpt = (double*)malloc(sizeof(double) * W * A);
  double state[A * A], trans[A * A];
  int8_t* X = (int8_t*)malloc((size_t)N * C);
  for (int i = 0; i < W * A * ldc; ++i) coef[i] = (lcg() - 0.5) * 0.2;
  for (int i = 0; i < W * A; ++i) icpt[i] = lcg() - 0.5;
  for (int i = 0; i < A * A; ++i) state[i] = (lcg() - 0.5) * 2.0 + ((i / A == i % A) ? 4.0 : 0.0);
  for (int i = 0; i < A * A; ++i) trans[i] = (lcg() - 0.5) + ((i / A == i % A) ? 3.0 : 0.0);
  for (int i = 0; i < N * C; ++i) { const double u = lcg(); X[i] = (int8_t)(u < 0.02 ? 2 : (u < 0.45 ? 1 : 0)); }

  gnx_ctx* ctx = NULL;
  int rc = gnx_init(0, &ctx);
  if (rc != GNX_OK) {
    fprintf(stderr, "gnx_init failed (%d): %s\n", rc, ctx ? gnx_last_error(ctx) : "no context");
    if (ctx) gnx_ctx_free(ctx);
    return rc == GNX_EHIP ? 2 : 1;
  }
  gnx_model_desc d;
  for (size_t i = 0; i < sizeof d; ++i) ((char*)&d)[i] = 0;
  d.abi_version = GNX_ABI_VERSION;
  d.A = A; d.C = C; d.M = M; d.ctx = CTX; d.S = 75;
  d.base_kind = GNX_BASE_LOGISTIC; d.smooth_kind = GNX_SMOOTH_CRF;
  d.lr_coef = coef; d.lr_ldc = ldc; d.lr_intercept = icpt;
  d.crf_state = state; d.crf_trans = trans;
  gnx_model* m = NULL;
  rc = gnx_model_load(ctx, &d, &m);
  if (rc != GNX_OK) { fprintf(stderr, "gnx_model_load failed (%d): %s\n", rc, gnx_last_error(ctx)); gnx_ctx_free(ctx); return 1; }
  double* proba = (double*)malloc(sizeof(double) * N * W * A);
  int32_t* labels = (int32_t*)malloc(sizeof(int32_t) * N * W);
  rc = gnx_infer(m, X, N, C, NULL, proba, labels);
  if (rc != GNX_OK) { fprintf(stderr, "gnx_infer failed (%d): %s\n", rc, gnx_last_error(ctx)); return 1; }
  uint32_t ls = 0;
  double ps = 0.0;
  for (int i = 0; i < N * W; ++i) ls = (ls + (uint32_t)(i + 1) * (uint32_t)labels[i]) & 0x7fffffffu;
  for (int i = 0; i < N * W * A; ++i) ps += proba[i];
  printf("labels %u proba %.6f\n", ls, ps);
  gnx_model_free(m);
  gnx_ctx_free(ctx);
  free(coef); free(icpt); free(X); free(proba); free(labels);
  return 0;
}
