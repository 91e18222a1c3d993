/* gnomix_io.h — the file side of the hot path, C ABI (part of libgnomix_hip.so, ABI version as gnomix_hip.h).
 *
 * The reference's command line spends its time HERE, not in the models (demo.ipynb cell 10: "loading the query file,
 * second biggest writing to disk"): it reads the query VCF through scikit-allel's C parser and writes .msp / .fb through
 * numpy / pandas string conversion.  These entry points replace, one for one:
 *
 *   gnx_vcf_read            <- read_vcf (allel.read_vcf, gzip.open)                    src/utils.py:55-81, gnomix.py:48
 *   gnx_vcf_gt_int8         <- vcf_data["calldata/GT"] (n_var, n_samples, 2) int8       src/utils.py:121-123
 *   gnx_infer_gt2           <- vcf_to_npy + Base.predict_proba + Smoother.predict_proba src/utils.py:104-159, gnomix.py:49-58
 *   gnx_phase_gt2           <- vcf_to_npy + Gnomix.phase + Gnomix.predict_proba         gnomix.py:49-72, src/model.py:188-214
 *   gnx_gt2_to_x_dev        <- the matrix vcf_to_npy returns, built in HBM              src/utils.py:118-153
 *   gnx_gt2_to_p2_dev       <- the same matrix, four SNPs per byte (k_base_logistic_p2 reads it)   src/utils.py:118-153
 *   gnx_x_to_gt2_dev        <- X_query_phased[:, fmt_idx] per variant                   gnomix.py:69, src/utils.py:299-308
 *   gnx_write_msp           <- write_msp                                                src/postprocess.py:84-98
 *   gnx_write_fb            <- write_fb (pandas to_csv of float columns)                src/postprocess.py:100-126
 *   gnx_write_vcf_gt2       <- npy_to_vcf (pandas to_csv of "a|b" columns)              src/utils.py:247-329
 *   gnx_format_floats       <- numpy's shortest round-trip float text (what `.astype(str)` / to_csv print)
 *
 * Genotype layout ("gt2"): VARIANT-MAJOR 2-bit fields, the order the text arrives in.  Row v = variant v, ldg bytes apart
 * (a multiple of 4); haplotype h = 2*sample + {0: left allele, 1: right allele} lives in bits 2*(h%4)..2*(h%4)+1 of byte
 * h/4 — the field convention of gnx_pack_x.  Codes: 0 = allele 0, 1 = allele 1, 2 = missing ('.', or an absent second
 * allele), 3 = allele >= 2 (the allele number itself is kept in a side list, gnx_vcf_gt_int8 restores it).  vcf_to_npy maps
 * everything that is not 0 / 1 to the missing code 2 AFTER the REF flip (src/utils.py:147-151), so 2 and 3 are the same
 * symbol for inference.
 *
 * Conventions as gnomix_hip.h: 0 / negative GNX_E* codes, no exceptions; the VCF entry points that take no context keep
 * their message per thread (gnx_io_last_error).  All host work runs on a process-wide pool of n_threads threads
 * (<= 0: the cores this process may run on, GNX_IO_THREADS overrides).
 */
#ifndef GNOMIX_IO_H
#define GNOMIX_IO_H

#include "gnomix_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gnx_vcf gnx_vcf; /* a parsed VCF held by the library */

typedef struct gnx_vcf_info {
  int64_t n_variants, n_samples;
  int64_t ldg;            /* bytes between variant rows of the 2-bit genotype matrix */
  int64_t file_bytes;     /* size of the file on disk */
  int64_t text_bytes;     /* bytes of VCF text parsed (after decompression) */
  int64_t n_fast_lines;   /* records taken by the fixed-width "a|b" path */
  int64_t n_general_lines;
  int64_t n_overflow;     /* alleles >= 2 kept in the side list */
  double seconds_load;    /* open / inflate / header */
  double seconds_parse;   /* read + fields + genotypes, all chunks */
  double seconds_alloc;   /* the (page-locked) genotype matrix */
  double seconds_merge;   /* chunk columns -> final arrays */
  int32_t n_threads;
  int32_t compression;    /* 0 plain text, 1 gzip (one stream: serial inflate), 2 BGZF (blocks inflated in parallel) */
  int32_t region_fallback; /* 1: `region` matched no record and the whole file was used (src/utils.py:72-78) */
  int32_t gt2_pinned;     /* 1: the genotype matrix sits in page-locked memory of the context passed to gnx_vcf_read */
} gnx_vcf_info;

enum { GNX_VCF_CHROM = 0, GNX_VCF_ID = 1, GNX_VCF_REF = 2, GNX_VCF_ALT0 = 3, GNX_VCF_ALT1 = 4, GNX_VCF_ALT2 = 5,
       GNX_VCF_SAMPLES = 6, GNX_VCF_META = 7 };

const char* gnx_io_last_error(void);

/* Parse `path` (plain text, gzip or BGZF — recognised by content).  `region`: keep the records whose CHROM equals it
 * (NULL / "": all); when nothing matches, all records are kept and info.region_fallback is set, as the reference does.
 * `ctx` may be NULL (no GPU needed); with a context the genotype matrix is allocated page-locked so that gnx_infer_gt2
 * moves it by DMA. */
int gnx_vcf_read(gnx_ctx* ctx, const char* path, const char* region, int n_threads, gnx_vcf** out);
void gnx_vcf_free(gnx_vcf* vcf);
int gnx_vcf_get_info(const gnx_vcf* vcf, gnx_vcf_info* out);
const uint8_t* gnx_vcf_gt2(const gnx_vcf* vcf);   /* (n_variants, ldg) */
const int64_t* gnx_vcf_pos(const gnx_vcf* vcf);   /* (n_variants,) variants/POS */
const float* gnx_vcf_qual(const gnx_vcf* vcf);    /* (n_variants,) variants/QUAL, NaN for '.' */
/* string columns: entry i = blob[offsets[i] .. offsets[i+1]); CHROM / ID / REF / ALT0..2 have n_variants entries (ALT
 * beyond the record's alternates: empty), SAMPLES n_samples, META one entry = all '##' lines (read_headers, utils.py:232) */
int gnx_vcf_strings(const gnx_vcf* vcf, int field, const char** blob, const int64_t** offsets, int64_t* n);
/* calldata/GT exactly as scikit-allel returns it: (n_variants, n_samples, 2) int8, -1 = missing */
int gnx_vcf_gt_int8(const gnx_vcf* vcf, int8_t* out, int n_threads);

/* raw DEFLATE (RFC 1951) -> exactly out_n bytes at `out` (the payload of one BGZF block and its ISIZE): the word-at-a-time
 * decoder gnx_vcf_read inflates .vcf.gz queries with (csrc/gnx_inflate.cpp; zlib remains the fallback for a block it rejects).
 * 0: ok; -1: corrupt or truncated input, or a size other than out_n.  No context, no GPU. */
int gnx_io_inflate_raw(const uint8_t* in, size_t in_n, uint8_t* out, size_t out_n);
/* CRC-32 of the gzip / BGZF trailer (zlib's crc32(0, data, n)), by carry-less multiplication where the host has PCLMULQDQ (> 10 GB/s
 * per thread: every BGZF block the reader inflates is checked against its trailer) */
uint32_t gnx_io_crc32(const uint8_t* data, size_t n);

/* ---- gt2 <-> the int8 matrix of the models, on the device (context stream; device pointers) --------------------------
 * src (C,) int32 describes vcf_to_npy's column map: src[c] = v | (flip << 30) — model SNP c is variant row v of G, with
 * 0 <-> 1 exchanged when the query's REF differs from the model's (utils.py:136-147) — or -1: absent from the query
 * (filled with the missing code 2, utils.py:131).  X[n, c] for haplotypes n0 <= n < n0 + N, rows ldx bytes apart. */
int gnx_gt2_to_x_dev(gnx_ctx* ctx, const uint8_t* dG, int64_t V, int64_t ldg, int64_t n0, int64_t N, const int32_t* d_src,
                     int64_t C, int8_t* dX, int64_t ldx);
/* the same matrix as 2-bit rows (the gnx_pack_x layout of gnomix_hip.h: SNP c of haplotype n = bits 2 (c % 4).. of byte c / 4 of
 * row n, rows ldp >= ceil(C / 4) bytes apart): what gnx_infer_gt2* hands to the 2-bit-native logistic pass — the haplotype-major
 * matrix never exists as int8 on that route */
int gnx_gt2_to_p2_dev(gnx_ctx* ctx, const uint8_t* dG, int64_t V, int64_t ldg, int64_t n0, int64_t N, const int32_t* d_src,
                      int64_t C, uint8_t* dP, int64_t ldp);
/* the way back for the phased VCF: row r of G_out = column cols[r] of X (values & 3), haplotypes n0 .. n0 + N written
 * into their fields (whole bytes: n0 and N multiples of 4 unless the range ends the row; other bytes untouched) */
int gnx_x_to_gt2_dev(gnx_ctx* ctx, const int8_t* dX, int64_t N, int64_t ldx, int64_t n0, const int32_t* d_cols, int64_t V,
                     uint8_t* dG, int64_t ldg);

/* ---- file-side inference: host gt2 in, host outputs out (batches over haplotypes inside, H2D / kernels / D2H overlapped) ----
 * G (V, ldg) gt2 with N = 2 * n_samples haplotypes (N even); outputs as gnx_infer: proba (N, W, A), labels (N, W). */
int gnx_infer_gt2(gnx_model* model, const uint8_t* G, int64_t V, int64_t ldg, int64_t N, const int32_t* src,
                  float* proba_f32, double* proba_f64, int32_t* labels);
/* The same for the haplotypes h0 .. h0 + N of G only (h0 a multiple of 4; whole samples: N even): what ONE device of a node does
 * with its share of the query (individuals shard contiguously, the model is replicated, no collective).  One parsed
 * query, one page-locked G; every device's context (its own host thread: a context is not thread-safe, different contexts are
 * independent) uploads only its columns of the variant rows (one strided copy) and writes its own row block of the shared outputs:
 * the output pointers address the rows of THIS range (row 0 = haplotype h0). */
int gnx_infer_gt2_range(gnx_model* model, const uint8_t* G, int64_t V, int64_t ldg, int64_t h0, int64_t N, const int32_t* src,
                        float* proba_f32, double* proba_f64, int32_t* labels);
/* gnomix.py:60-72 with phase=True: B = base.predict_proba(X); X_phased, labels = model.phase(X, B);
 * proba = model.predict_proba(X_phased).  G_out (n_out, ldg_out) receives X_phased[:, out_cols[r]] as gt2 rows (may be
 * NULL); n_switches (N/2,) may be NULL. */
int gnx_phase_gt2(gnx_model* model, const uint8_t* G, int64_t V, int64_t ldg, int64_t N, const int32_t* src, int32_t max_it,
                  const int32_t* out_cols, int64_t n_out, uint8_t* G_out, int64_t ldg_out, float* proba_f32,
                  double* proba_f64, int32_t* labels, int32_t* n_switches);
/* ... for the haplotypes h0 .. h0 + N only (see gnx_infer_gt2_range; h0 a multiple of 4 = whole individuals).  Output pointers
 * address the rows of this range; G_out is the WHOLE (n_out, ldg_out) matrix: the call writes bytes h0/4 .. of every row. */
int gnx_phase_gt2_range(gnx_model* model, const uint8_t* G, int64_t V, int64_t ldg, int64_t h0, int64_t N, const int32_t* src,
                        int32_t max_it, const int32_t* out_cols, int64_t n_out, uint8_t* G_out, int64_t ldg_out,
                        float* proba_f32, double* proba_f64, int32_t* labels, int32_t* n_switches);

/* ---- writers ----------------------------------------------------------------------------------------------------------
 * Every file = `head` (head_len bytes, written as is) + one text row per window / variant: the caller's row prefix
 * (prefix_blob[prefix_off[r] .. prefix_off[r+1]), the metadata columns already joined by tabs) followed by the values. */
/* .msp row w: prefix, then "\t<labels[n, w]>" for every haplotype n (labels (N, ldl) int32, src/postprocess.py:84-98) */
int gnx_write_msp(const char* path, const char* head, int64_t head_len, const char* prefix_blob, const int64_t* prefix_off,
                  const int32_t* labels, int64_t N, int64_t ldl, int64_t W, int n_threads);
/* .fb row w: prefix, then "\t<proba[n, w, a]>" for n, then a (proba (N, W, A) float32 or float64); numbers are printed as
 * pandas' to_csv prints a float column: numpy's shortest round-trip text, empty for NaN (src/postprocess.py:100-126) */
int gnx_write_fb(const char* path, const char* head, int64_t head_len, const char* prefix_blob, const int64_t* prefix_off,
                 const void* proba, int proba_is_f64, int64_t N, int64_t W, int64_t A, int n_threads);

/* the same file with the number text produced on the GPU (k_fb_text.hip): proba (N, W, A) float32 on the host goes back to HBM, the
 * body returns as one page-locked buffer and is written with one write() — byte-identical to gnx_write_fb.  Measured no faster for
 * chr22 x 10 000 haplotypes into tmpfs (0.08 s either way: the write of a fresh 305 MB file is the bound), but it leaves the host's
 * cores free. */
int gnx_write_fb_dev(gnx_ctx* ctx, const char* path, const char* head, int64_t head_len, const char* prefix_blob, const int64_t* prefix_off,
                     const float* proba, int64_t N, int64_t W, int64_t A);
/* VCF row v: prefix (CHROM .. FORMAT joined by tabs), then "\t<a>|<b>" per sample from gt2 row v (codes printed as the
 * digits 0..3: the reference prints the int8 matrix, missing = 2, src/utils.py:299-308; missing_as_dot != 0 prints code 2
 * as '.', the VCF spelling of a missing allele — what a query file carries) */
int gnx_write_vcf_gt2(const char* path, const char* head, int64_t head_len, const char* prefix_blob, const int64_t* prefix_off,
                      const uint8_t* G, int64_t V, int64_t ldg, int64_t n_samples, int missing_as_dot, int n_threads);
/* the same with the prefixes taken from a parsed VCF: variant rows[r] of `src` supplies CHROM, POS, ID, QUAL; REF / ALT come
 * from the override blobs when given (the model's alleles, gnomix.py:62-66) else from `src`; FILTER "PASS", INFO ".",
 * FORMAT "GT" (src/utils.py:283-291) */
int gnx_write_phased_vcf(const char* path, const char* head, int64_t head_len, const gnx_vcf* src, const int64_t* rows, int64_t V,
                         const char* ref_blob, const int64_t* ref_off, const char* alt_blob, const int64_t* alt_off,
                         const uint8_t* G, int64_t ldg, int64_t n_samples, int n_threads);

/* numpy's text of n floats (float32, or float64 when is_f64): out = the strings back to back, off (n + 1,) their offsets;
 * out must hold 32 n bytes.  What str(np.float32(x)) / arr.astype(str) give: shortest digits that round-trip, positional
 * for 1e-4 <= |x| < 1e16, scientific with a two-digit exponent otherwise, "nan", "inf". */
int gnx_format_floats(const void* values, int is_f64, int64_t n, char* out, int64_t* off);

#ifdef __cplusplus
}
#endif
#endif /* GNOMIX_IO_H */
