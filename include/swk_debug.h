/*
 * swk_debug.h -- A/B switches, profiling hooks and diagnostics of libswk.so.
 *
 * Nothing in here belongs to the drop-in boundary (swk.h): a maintainer who binds the reference's call surface never calls these.
 * They exist for the measurements (bench.py, tools/), for the cross-checks of the GPU tests (every IALM pass variant and both
 * small-matrix solvers against the CPU restatement of the reference) and for the counters the bench line reports.  Results never depend on a switch: each one
 * selects between implementations that the tests hold to the same outputs.
 */
#ifndef SWK_DEBUG_H
#define SWK_DEBUG_H
#include "swk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement hooks -----------------------------------------------------------
 * With profiling on, every kernel launch of swk_batch_run is bracketed by HIP events on
 * the context's stream; swk_prof_get returns accumulated device time and launch count per
 * kernel family since the last swk_prof_reset.  Family ids: */
enum {
    SWK_K_GRAY = 0, SWK_K_IALM_STATS = 1, SWK_K_IALM_PASS = 2, SWK_K_IALM_SMALL = 3,
    SWK_K_FILTER = 4, SWK_K_CCL = 5, SWK_K_PROPS = 6, SWK_K_COPY = 7, SWK_K_COUNT = 8
};
int32_t swk_prof_enable(swk_ctx *ctx, int32_t on);
int32_t swk_prof_reset(swk_ctx *ctx);
int32_t swk_prof_get(swk_ctx *ctx, int32_t family, double *ms_total, int64_t *launches);
/* Total IALM pass launches x windows still active, i.e. window-iterations streamed. */
int32_t swk_prof_window_iters(swk_ctx *ctx, int64_t *window_iters);
/* Select the IALM pass kernel: 0 = auto (4, or 2 when A / E are requested, 6 above 64 frames), 6 = the plain float64 kernels of long
 * windows (accepted for any n: the tests compare them with 1), 1 = LDS/VALU kernel,
 * 2 = MFMA f64 kernel carrying A and Y, 4 = MFMA f64 kernel carrying M alone (21-22 instead of 34 B per element and
 * iteration; produces the sparse u8 image and the iteration count, not A / E), instantiated per 4-frame k-step with a
 * software-pipelined tile loop, 5 = 4 without the pipeline (its cross-check).  3 (round 1's M-state kernel, one instantiation
 * per 16-frame block) is gone: SWK_ERR_ARG.  For A/B measurements and cross-checks only. */
int32_t swk_set_ialm_variant(swk_ctx *ctx, int32_t variant);
/* k-step-templated M-state pass (variants 4 / 5): bit 0 = the wave in the odd hardware slot of each SIMD runs at raised
 * priority (breaks the lockstep of the two co-resident waves), bit 1 = it also starts late.  A/B knob; results never
 * depend on it. */
int32_t swk_set_pass_tuning(swk_ctx *ctx, int32_t flags);
/* M-state pass only: the per-iteration stores of the sparse u8 image start once ||Z||_F < factor * tol * ||X||_F
 * (default 16; <= 0 = every pass).  A window that stops although the pass before its last iteration skipped the
 * stores makes the library run the batch again without the speculation, so results never depend on the factor;
 * swk_prof_redo_batches counts those reruns. */
int32_t swk_set_sparse_speculation(swk_ctx *ctx, double factor);
/* M-state pass only: while the last formed ||Z||_F is >= factor * tol * ||X||_F (default 256; <= 0 = never) the
 * stopping norm is formed every other iteration only (the f16 copy of Y/mu is neither written nor read in between).
 * In between, the norm over frames 0..3 is still formed: a lower bound that proves the skipped iteration did not
 * stop; a window where it cannot is rerun like above.  After a rerun the guess that failed stays off for the next
 * 64 batches of the context (the windows of one video behave alike). */
int32_t swk_set_norm_speculation(swk_ctx *ctx, double factor);
/* M-state pass only: the stopping test ||Z||_F < tol ||X||_F (image_filtering.py:297) is made on a float32 sum over a binary16 copy
 * of Y/mu (relative error about 1e-6).  A window whose ratio ||Z|| / (tol ||X||) comes within `rel` of 1 (default 1e-3; 0 = off) is
 * not decided on that number: it is run again, alone, by the A/Y-state pass, whose norm is formed in float64 like the reference's.
 * swk_prof_guard_windows counts those windows. */
int32_t swk_set_norm_guard(swk_ctx *ctx, double rel);
int32_t swk_prof_guard_windows(swk_ctx *ctx, int64_t *windows);
/* Diagnostic of the same decision: for every window of the last swk_batch_run / swk_ialm (up to cap) the ratio ||Z||_F / ||X||_F of its
 * LAST stopping test (image_filtering.py:297) and, for the M-state pass, the bound on that number's relative error the band is held
 * against (float32 sum over a binary16 copy of Y/mu: csrc/ialm_small_dev.h; the effective band is max(rel, 4 x bound) per window).
 * Returns the number of windows of that batch (negative: error).  The A/Y-state pass forms the norm in float64: bound 0. */
int32_t swk_last_stopping_norms(swk_ctx *ctx, double *ratio, double *err_bound, int32_t cap);
/* M-state pass only: 1 (default) = statistics and the first iteration's Gram matrix come from one read of X on the
 * integer matrix cores wherever the first shrinkage provably removes nothing; 0 = always the f64 start pass. */
int32_t swk_set_integer_start(swk_ctx *ctx, int32_t on);
/* Windows of the last swk_batch_run / swk_ialm whose start came from the integer kernel (the others ran the f64 start pass). */
int32_t swk_last_integer_start_windows(swk_ctx *ctx, int32_t *windows);
/* Diagnostic: iterations the G^(-1/2) solver took in the LAST small-matrix step of the last batch, maximum over its
 * windows: Newton-Schulz iterations, or 100 + Jacobi sweeps where that solver ran. */
int32_t swk_last_eig_sweeps(swk_ctx *ctx, int32_t *sweeps);
int32_t swk_prof_redo_batches(swk_ctx *ctx, int64_t *batches);
/* ... and the windows that were run again for it (only the windows whose guess failed run again, together, with the guesses off). */
int32_t swk_prof_redo_windows(swk_ctx *ctx, int64_t *windows);
/* M-state pass: algorithmic bytes per matrix element moved by all its launches since swk_prof_reset, summed over
 * windows (each window-iteration counts X 1 + M 8 (+8 read) + U 2 or 1/8 each way + 1 when the sparse image is
 * stored); multiply by n*P for bytes.  Meaningful for the M-state pass only. */
int32_t swk_prof_pass_bytes_per_element(swk_ctx *ctx, double *bytes);
/* G^(-1/2) of the n x n Gram matrix: 0 = coupled Newton-Schulz on the f64 matrix cores (default; falls back
 * to Jacobi by itself if it does not converge), 1 = cyclic Jacobi eigen-solve. */
int32_t swk_set_eig_method(swk_ctx *ctx, int32_t method);

/* Accurate first iteration of ill-conditioned windows (csrc/ialm_refine.hip).  The Gram-matrix route squares cond(M); in iteration 1
 * (M_1 = c X, 1/mu largest) that costs about eps * cond(G_1) / mu_0 in A, and windows of few pixels and many frames carry that error
 * to the end.  A window whose estimate eps * ||G_1||_F sum_i 1/lambda_i / mu_0 exceeds `tau` (default 1e-5; <= 0 = never) gets B_1
 * from a double-double Cholesky factor of the exact integer X^T X (of a double-double M_1^T M_1 where the first shrinkage clips)
 * instead.  swk_prof_refined_windows: windows refined / wanted but given up (rank deficient, or too large for one workgroup's
 * double-double Gram matrix) since the context was made. */
int32_t swk_set_start_refine(swk_ctx *ctx, double tau);
int32_t swk_prof_refined_windows(swk_ctx *ctx, int64_t *refined, int64_t *unrefined);

/* ---- classifier kernels: A/B switches ---- */
/* Measurement knobs of the classifier kernels (A/B runs).  knob 0: workgroup layout of the 1 x 1 kernel (0 = 16-wave workgroups, the
 * default; 1 = 8 waves with the deepest activation ring that fits; results do not depend on it).  knob 1: 1 = the Fire modules' expand1x1
 * shapes run on the split-bf16 kernel (float32 products as six bf16 MFMA products of three-way split operands: float32-accurate, another
 * summation order); 0 = the default, the float32 kernel. */
int32_t swk_set_cnn_tuning(int32_t knob, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* SWK_DEBUG_H */
