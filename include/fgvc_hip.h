/*
 * fgvc_hip.h -- C ABI of libfgvc_hip.so: FGVC's label-propagation inference hot path
 * as hand-written HIP kernels for MI355X (gfx950 / CDNA4).
 *
 * The reference (qianduoduolr/FGVC, "mmpt") is pure Python; its FFI for this path is
 * "call these torch functions".  Each entry point below replaces the body of one reference
 * function (cited as file:line relative to the reference tree); fgvc_amd/mmpt_api/ rebuilds the
 * reference's Python signatures on top of them and INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked "host"; buffers are caller-owned
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     enqueued asynchronously on it; nothing here allocates, frees or synchronises
 *     (the functions are hipGraph-capturable)
 *   - return value: FGVC_OK or an FGVC_ERR_* code; fgvc_last_error() gives the text
 *     (thread-local).  No exceptions cross the ABI.
 *   - feature maps are CHANNELS-LAST ("hwc"): feat[frame][pixel][channel], pixel = y*W + x
 *   - label maps are PIXEL-MAJOR ("hwp"):     lab[frame][pixel][label]
 *   - top-k lists are in canonical order: score descending, index ascending among equals
 *     (torch.topk leaves tie order unspecified; see DESIGN.md "tie policy")
 */
#ifndef FGVC_HIP_H
#define FGVC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FGVC_OK 0
#define FGVC_ERR_INVALID_ARG 1
#define FGVC_ERR_UNSUPPORTED 2
#define FGVC_ERR_LAUNCH 3

/* mask predicate on the integer offset (dy,dx) = key pixel - query pixel
 *   keep  <=>  dy*dy+dx*dx <= r2max  &&  |dy| <= ry  &&  |dx| <= rx
 * circle (affinity_utils.py:98-109, local_attention.py:463-467): r2max = largest d2 with
 *   sqrtf(d2) < radius (fgvc_r2max_for_radius), ry = rx = FGVC_NO_LIMIT
 * square (affinity_utils.py:86-96): r2max = FGVC_NO_LIMIT, ry = nr_h/2, rx = nr_w/2
 * none: all three FGVC_NO_LIMIT */
#define FGVC_NO_LIMIT 0x3fffffff

/* pairs[i] = {query_frame, key_frame, flags, reserved} */
#define FGVC_PAIR_MASKED 1      /* apply the mask predicate to this pair (else full frame)   */

/* weight modes of fgvc_merge_topk_f32 (local_attention.py:368-373) */
#define FGVC_WEIGHT_SOFTMAX 0
#define FGVC_WEIGHT_COSINE 1
#define FGVC_WEIGHT_RAW 2     /* fgvc_dense_attend_f32 only: the affinity itself is the weight, no normalisation (local_square_attention, topk=None) */

const char* fgvc_version(void);
const char* fgvc_last_error(void);

/* Process-wide tuning knobs (host only, not thread-safe against concurrent launches).
 *   "conv_cot_cap"       fgvc_conv_split_f32: at most this many output channels per workgroup (0 = widest, 64, 128).
 *   "conv_narrow"        fgvc_conv_split_f32: 4-row tiles with two workgroups per CU for the 64-channel layers (bit 0, default) /
 *                        the 128-channel 3x3 layers (bit 1).
 *   "readout_prune"      fgvc_softargmax_top5_f32: 1 (default) = pruned read-out, full scan only for the maps it hands back;
 *                        0 = full scan of every map.  Identical results.
 *   "conv64_variant"     fgvc_conv64_split_fmt_f32 (f16 + fp8 operands): 0 (default) = conv64p_kernel, round 5's one-stream-per-tile kernel, for the
 *                        forms the encoder launches; 16 = conv64_kernel<1> (round 3's: the kernel every other form runs on), 32 = the K-split
 *                        experiment, 128 = conv64p_kernel in raster tile order; 8 = s_memtime probe of one workgroup (fgvc_conv64_probe).
 *                        16 and 128 give identical results (A/B switches); the tests hold 0 against 16 bit for bit.
 *   "conv_debug" 1024    fgvc_conv_split_* with f16 + FP6 operands, 3 x 3, 128 / 256-channel tiles: conv_split_kernel (rounds 2-4) instead of
 *                        conv256p_kernel (round 5: one wave per SIMD, main loop one assembly statement).  Identical results (A/B switch).
 *   "pair_f16_debug" 4194304   fgvc_pair_topk_f16f6[x]: pair_topk_kernel_v8 (round 6: one kind of wave, a key block staged once for eight query
 *                        blocks) instead of pair_topk_kernel_v7.  Identical scores, lists equal up to exact score ties across the K-th place.
 *   "pair_debug", "pair_f16_debug", "corr_debug", "corr6_debug", "corr8_debug", "conv_debug", "conv_s2_debug": besides such A/B bits and
 *                        the s_memtime probes these words carry PROFILING ABLATIONS (a kernel without its stores / matrix chain / selection /
 *                        staging): results are WRONG.  libfgvc_hip.so REFUSES them (FGVC_ERR_UNSUPPORTED); libfgvc_hip_ablations.so -- the
 *                        same objects, this one function compiled with -DFGVC_ABLATIONS -- accepts them (fgvc_amd._lib.ablations(),
 *                        FGVC_HIP_LIB; tools/experiments/ablate_*.py, time_*.py). */
int fgvc_set_option(const char* name, int value);

/* Largest integer d2 such that sqrtf((float)d2) < radius, -1 if none (host helper). */
int fgvc_r2max_for_radius(float radius);

/* ---- A5 step 0: F.normalize(dim=1) + NCHW -> channels-last -------------------------------
 * replaces local_attention.py:308-313 (F.normalize(query/key, p=2, dim=1) and the .view()s).
 * in [n][C][HW] f32 (the encoder's NCHW output)  ->  out [n][HW][c_out] f32, c_out >= C,
 * out = in / max(||in||_2 over C, 1e-12) when normalize != 0, plain transpose otherwise;
 * channels C..c_out-1 are written as zeros (zero padding does not change any dot product; it lets
 * callers round C up to a channel count the MFMA kernels support). */
int fgvc_normalize_chw_to_hwc_f32(const float* in, float* out, int n, int C, int HW,
                                  int normalize, int c_out, void* stream);

/* ---- A5 step 1: windowed correlation + running top-k, one (query frame, key frame) pair per
 * grid.y.  Replaces the einsum / masked_fill_ / topk of local_attention.py:321-356 (and the
 * per-chunk mask rebuild of masked_attention_efficient_v2, :452-475) for ONE key frame; the
 * (T*HW x step) slab is never materialised.  f32 MFMA (v_mfma_f32_32x32x2_f32), exact f32.
 *   qfeat [n_qframes][Hq*Wq][C], kfeat [n_kframes][Hk*Wk][C]   (normalised, channels-last)
 *   pairs [n_pairs][4] int32 (device)
 *   idx_out   [n_pairs][Hq*Wq][topk] int32   key pixel index ky*Wk+kx, -1 where fewer than topk
 *                                            candidates exist
 *   score_out [n_pairs][Hq*Wq][topk] f32     raw dot product (NOT yet divided by temperature),
 *                                            descending; -inf where idx = -1
 *   dense_mask: NULL, or an arbitrary [Hk*Wk][Hq*Wq] bool (uint8) mask tensor as accepted by
 *               local_attention.py:329-353; then the analytic predicate must be off (all FGVC_NO_LIMIT)
 *               and FGVC_PAIR_MASKED pairs traverse the full frame consulting the tensor
 * C in {32, 64, 128, 256}; 1 <= topk <= 16.  An analytic mask needs Hq==Hk, Wq==Wk. */
int fgvc_pair_topk_f32(const float* qfeat, const float* kfeat, const int32_t* pairs, int n_pairs,
                       int C, int Hq, int Wq, int Hk, int Wk, int r2max, int ry, int rx, int topk,
                       const uint8_t* dense_mask, int32_t* idx_out, float* score_out, void* stream);

/* Same operator and outputs as fgvc_pair_topk_f32 on the 16-bit matrix pipe, f32-grade (replaces local_attention.py:331-371 like
 * fgvc_pair_topk_f32; the default pair kernel of the engine for C == 256): f16 operands h = f16(2^14 x), l = f16(2^14 x - h) as
 * fgvc_split_f16x2 writes them ([pixel][h C | l C]), three products h h + l h + h l on v_mfma_f32_32x32x16_f16 accumulated in f32
 * (22 significand bits per element; a score differs from the f32 dot product by ~1e-7, the size of f32 summation-order noise);
 * selection runs on fixed-point keys with 24 fractional bits (scores closer than 6e-8 tie and are ordered by pixel index, like
 * equal f32 scores), in waves of their own beside the multiplying ones; key blocks travel through a producer / consumer ring
 * without workgroup barriers.
 *   Precondition: feature rows L2-normalised (|q.k| <= 1), as fgvc_normalize_chw_to_hwc_f32(normalize=1) makes them.
 *   C == 256; 1 <= topk <= 10; analytic mask only (a dense mask tensor needs fgvc_pair_topk_f32).
 *   A query tile walks a list of at most 4096 key blocks (4x8 pixels): the key blocks within the mask's reach for a pair with
 *   FGVC_PAIR_MASKED, the whole key grid for a pair without.  `pairs` lives on the device, so the caller states with
 *   all_masked != 0 that EVERY pair carries FGVC_PAIR_MASKED; then only the reach is checked against the list (a 480x854 grid
 *   with a radius-6 window is fine), otherwise the whole grid must fit (FGVC_ERR_UNSUPPORTED beyond 4096 blocks).  A pair
 *   that breaks the promise on such a grid gets empty lists (-1 / -inf), never truncated ones. */
int fgvc_split_f16x2(const float* feat, uint16_t* h_l /* [n][2][C] f16: h then l */, int64_t n_pixels, int C, void* stream);
int fgvc_pair_topk_f16x3(const uint16_t* qsplit, const uint16_t* ksplit, const int32_t* pairs, int n_pairs, int C, int Hq, int Wq,
                         int Hk, int Wk, int r2max, int ry, int rx, int topk, int all_masked, int32_t* idx_out,
                         float* score_out, void* stream);
/* ... with the pairs taken in RUNS: runs[n_runs][2] (device, int32) = (first pair, count) of consecutive pairs that share the query
 * frame AND the FGVC_PAIR_MASKED flag (the driver's pair list has them that way: a query frame against its <= 6 key frames,
 * vanilla_tracker.py:353-362); every pair belongs to exactly one run.  A workgroup then sets the query block up once per run and
 * streams the key blocks of its pairs through one ring.  Outputs are indexed by pair as before.  The caller vouches for the runs
 * (they live on the device); list longer runs first (the launch does not reorder). */
int fgvc_pair_topk_f16x3_runs(const uint16_t* qsplit, const uint16_t* ksplit, const int32_t* pairs, int n_pairs, int C, int Hq, int Wq,
                              int Hk, int Wk, int r2max, int ry, int rx, int topk, int all_masked, const int32_t* runs, int n_runs,
                              int32_t* idx_out, float* score_out, void* stream);
int fgvc_pair_topk_f16x3_timed_out(void);

/* The same operator (replaces local_attention.py:331-371) in the PRECISION-CONSISTENT arithmetic of the rest of the path -- the
 * encoder computes in f16 + fp8, the dense volume in f16 + FP6: h = f16(256 x), the cross sums h_k l_q + l_k h_q in block-scaled FP6
 * (e2m3, one E8M0 scale per 32 channels) on v_mfma_scale_f32_32x32x64_f8f6f4: 1.5 matrix-pipe units per product where
 * fgvc_pair_topk_f16x3 spends 3, scores within ~6e-5 logit (tau 0.07) of float64 on Gaussian rows (bar: 1e-3), quantised to 2^-20
 * (1.4e-5 logit); indices equal the reference's wherever its scores are further apart than that error.  The engine uses it when the
 * encoder runs f16f8 (whose features are already +-3e-5 logit from the reference's); fgvc_pair_topk_f16x3 remains the 1e-7-grade form.
 *   fgvc_split_f16f6p: feat [n][256] f32 (L2-normalised) -> rows [n][1024] bytes, laid out for the pair kernel's lanes:
 *     [0, 512) h 256 f16 | [512, 640) h6 mains, group v at 32 v + 16 hi | [640, 704) h6 tails (8 B per group and lane half) |
 *     [704, 832) l6 mains | [832, 896) l6 tails | [896, 928) scale bytes, 16 hi + {H v, 4 + L v} | zero.  Group v, lane half hi, element
 *     e = 8 m + i is channel 64 v + 16 m + 8 hi + i (what a lane of the 32 x 32 x 16 f16 shape holds in fragments 4 v .. 4 v + 3);
 *     those 32 channels are one scale block (fgvc_amd/csrc/pair_topk_v7.hpp).  A DIFFERENT layout from fgvc_split_f16f6's.
 *   fgvc_pair_topk_f16f6[_runs]: arguments and outputs of fgvc_pair_topk_f16x3[_runs]; C == 256, topk <= 10, an analytic mask that
 *   every pair carries (all_masked != 0) and that reaches at most 64 key blocks (4 x 8 pixels) per 8 x 16 query tile -- a radius-15
 *   disc reaches 56 -- else FGVC_ERR_UNSUPPORTED.  Same fail-closed protocol; fgvc_pair_topk_f16x3_timed_out() reports for both. */
int fgvc_split_f16f6p(const float* feat, uint8_t* rows, int64_t n_pixels, int C, void* stream);
int fgvc_pair_topk_f16f6(const uint8_t* qsplit, const uint8_t* ksplit, const int32_t* pairs, int n_pairs, int C, int Hq, int Wq, int Hk,
                         int Wk, int r2max, int ry, int rx, int topk, int all_masked, int32_t* idx_out, float* score_out, void* stream);
int fgvc_pair_topk_f16f6_runs(const uint8_t* qsplit, const uint8_t* ksplit, const int32_t* pairs, int n_pairs, int C, int Hq, int Wq,
                              int Hk, int Wk, int r2max, int ry, int rx, int topk, int all_masked, const int32_t* runs, int n_runs,
                              int32_t* idx_out, float* score_out, void* stream);
/* The same rows with the pixel's EXACT channels behind them (round 5: the bank of the default configuration): 2 KiB per pixel =
 * [the 1 KiB row of fgvc_split_f16f6p | 256 f32 = the normalised channels themselves].  fgvc_pair_topk_f16f6x[_runs] = fgvc_pair_topk_f16f6[_runs]
 * on these rows (same arithmetic, same lists, a row stride of 2048 bytes); fgvc_merge_refine_topk_f32 reads the second KiB
 * (row_bytes = 2048, base + 1024) to re-score near-ties exactly.  One tensor, one slice per frame, one message per halo frame. */
int fgvc_split_f16f6x(const float* feat, uint8_t* rows, int64_t n_pixels, int C, void* stream);
int fgvc_pair_topk_f16f6x(const uint8_t* qrows, const uint8_t* krows, const int32_t* pairs, int n_pairs, int C, int Hq, int Wq, int Hk,
                          int Wk, int r2max, int ry, int rx, int topk, int all_masked, int32_t* idx_out, float* score_out, void* stream);
int fgvc_pair_topk_f16f6x_runs(const uint8_t* qrows, const uint8_t* krows, const int32_t* pairs, int n_pairs, int C, int Hq, int Wq,
                               int Hk, int Wk, int r2max, int ry, int rx, int topk, int all_masked, const int32_t* runs, int n_runs,
                               int32_t* idx_out, float* score_out, void* stream);
int fgvc_pair_topk_f16x3_probe(int64_t* out32);   /* debug: s_memtime words of one workgroup (pair_f16_debug = 256) */

/* ---- A5 step 2: merge the per-pair lists of the T key slots of each query frame, divide by the
 * temperature and turn the k logits into weights.  Replaces the global topk over T*HW
 * (local_attention.py:356) and :368-371.
 *   slot_pair [n_out][T] int32 (device): pair id feeding slot t of output frame f, -1 = unused
 *   idx_out [n_out][HWq][topk] = slot*HWk + key pixel   (the reference's flat key index, :312)
 *   logit_out, weight_out [n_out][HWq][topk] f32 */
int fgvc_merge_topk_f32(const int32_t* pair_idx, const float* pair_score, const int32_t* slot_pair,
                        int n_out, int T, int HWq, int HWk, int topk, float temperature,
                        int weight_mode, int32_t* idx_out, float* logit_out, float* weight_out,
                        void* stream);

/* ---- A5 step 2 behind a pair kernel with APPROXIMATE scores (fgvc_pair_topk_f16f6: |score - exact| <= eps): the same merge, made
 * index-exact again.  Replaces the global topk over T*HW (local_attention.py:353-356) like fgvc_merge_topk_f32; the lists it writes are
 * the exact top-k in the exact order wherever the pair kernel's scores are within `eps` (raw dot-product units) of the exact products.
 *   Candidates whose approximate scores lie within 2 eps of a neighbour in the merged order (and can reach the top k) are re-scored
 *   from the EXACT rows -- q_exact / k_exact: 256 f32 per pixel at `base + frame * frame_bytes + pixel * row_bytes` (row_bytes >= 1024:
 *   a plain [frame][pixel][256] f32 bank, or the f32 half of the 2 KiB rows of fgvc_split_f16f6x) -- with products and sums in f64;
 *   a query whose window the pair lists do not close (a slot's own last entry inside it) is recomputed from every candidate under the
 *   mask predicate (r2max / ry / rx as the pair kernel got them; pairs without FGVC_PAIR_MASKED scan the whole frame).
 *   pairs [n_pairs][4] int32 as the pair kernel took them; slot_pair, outputs, weight modes as fgvc_merge_topk_f32; 1 <= topk <= 10 = the
 *   length of the pair lists; C == 256.  Two slots fed by ONE pair (frame 0 twice while idx <= precede_frames,
 *   vanilla_tracker.py:353-362) are exact twins: the lower slot first, as equal scores are ordered everywhere.
 *   workspace: fgvc_merge_refine_workspace_bytes(n_out, HWq) bytes, 16-byte aligned; after the call its first five 32-bit words hold
 *   {queries re-scored, of them recomputed from scratch, candidates re-scored, from-scratch queries beyond the scan queue, the largest
 *   |approximate - exact| score among the re-scored candidates as f32 bits}: the last one is `eps` MEASURED on the call's own data --
 *   a caller that finds it above the eps it passed must not trust the lists (fgvc_amd raises). */
size_t fgvc_merge_refine_workspace_bytes(int n_out, int HWq);
int fgvc_merge_refine_topk_f32(const int32_t* pair_idx, const float* pair_score, const int32_t* slot_pair, const int32_t* pairs,
                               const void* q_exact, int64_t q_frame_bytes, int q_row_bytes, const void* k_exact, int64_t k_frame_bytes,
                               int k_row_bytes, int n_out, int T, int Hq, int Wq, int Hk, int Wk, int C, int topk, float temperature,
                               int weight_mode, float eps, int r2max, int ry, int rx, int32_t* idx_out, float* logit_out,
                               float* weight_out, void* workspace, void* stream);

/* ---- A5 step 3: label propagation  out[i][p] = sum_r weight[i][r] * labels[slot(idx)][pix(idx)][p]
 * replaces the index_select + einsum of local_attention.py:360-375.
 *   labels [n_label_frames][HWk][P]; slot_frame [T] int32 (device): label frame of each key slot
 *   window_L = 0: idx = slot*HWk + pixel.   window_L = L > 0 (A7 local window, vanilla_tracker.py:550-566):
 *   idx = slot*L*L + (dy+R)*L + (dx+R) relative to the query pixel, taps outside the image read 0. */
int fgvc_propagate_topk_f32(const float* labels, const int32_t* slot_frame, int T,
                            const int32_t* idx, const float* weight, int Hq, int Wq, int Hk, int Wk,
                            int P, int topk, int window_L, float* out, void* stream);

/* ---- A5'': dense correlation volume  vol[j][i] = <k_j, q_i> / temperature,  j key, i query
 * replaces affinity_utils.py:6-21 (compute_affinity), local_attention.py:231 / :321-323 and
 * correlation.py:51 for one (query, key) frame pair.  vol is [HWk][HWq] f32, row-major.
 *   _f32    : exact f32 MFMA
 *   _bf16x3 : inputs pre-split by fgvc_split_bf16 into hi+lo bf16, three bf16 MFMA products
 *             (hi*hi + hi*lo + lo*hi), f32 accumulate; |error| <= ~2e-5 in cosine
 *   _bf16   : hi part only (reduced precision, configs[4]) */
int fgvc_corr_volume_f32(const float* qfeat, const float* kfeat, int C, int HWq, int HWk,
                         float temperature, float* vol, void* stream);
int fgvc_split_bf16(const float* feat, uint16_t* hi_lo /* [n][2][C] bf16: hi then lo */, int64_t n_pixels,
                    int C, void* stream);
int fgvc_corr_volume_bf16x3(const uint16_t* q_hi_lo, const uint16_t* k_hi_lo, int C, int HWq, int HWk,
                            float temperature, float* vol, void* stream);
int fgvc_corr_volume_bf16(const uint16_t* q_hi_lo, const uint16_t* k_hi_lo, int C, int HWq, int HWk,
                          float temperature, float* vol, void* stream);

/* The parity-grade volume at two bf16-MFMA times per tile (bf16x3: three): x is stored as h = f16(256 x), h8 = e4m3(h) and
 * l8 = e4m3(256 (256 x - h)); sum h h on v_mfma_f32_32x32x16_f16, the cross sums h8 l8 on the block-scaled fp8 instruction
 * v_mfma_scale_f32_32x32x64_f8f6f4 (twice the K per cycle), l l (2^-22) dropped: every entry within ~1e-4 logit of the f32
 * product on Gaussian features (bound asserted by the tests: 1e-3, the north_star's score bar).  Rows must be L2-normalised
 * (|x| <= 1: 256 x and its residual stay inside the f16 / e4m3 ranges).  C == 256.
 *   fgvc_split_f16f8: feat [n][C] f32 -> out [n][4 C] bytes = [h: C f16 | h8: C bytes | l8: C bytes] per pixel.
 *   fgvc_corr_volume_f16f8: q, k in that format -> vol [HWk][HWq] f32 = <k, q> / temperature.  When HWq * 4 is not a multiple of
 *   128 bytes the launch splits the key rows into classes of equal line phase so that every store still writes whole lines. */
int fgvc_split_f16f8(const float* feat, uint8_t* out, int64_t n_pixels, int C, void* stream);
int fgvc_corr_volume_f16f8(const uint8_t* q_split, const uint8_t* k_split, int C, int HWq, int HWk,
                           float temperature, float* vol, void* stream);

/* The same volume (replaces affinity_utils.py:6-21 `compute_affinity`, local_attention.py:231) at 1.5 bf16-MFMA times per tile: the
 * cross sums in block-scaled FP6 (e2m3, one E8M0 scale per 32 consecutive channels) on v_mfma_scale_f32_16x16x128_f8f6f4, which
 * retires FP6 at four times the f16 rate.  Same accuracy as f16f8 (simulated and measured; asserted bound 1e-3 logit), rows must be
 * L2-normalised, C == 256.
 *   fgvc_split_f16f6: feat [n][256] f32 -> out [n][1024] bytes = [h: 256 f16 | h6: 192 B | l6: 192 B | 16 scale bytes | zero pad]
 *   (exact layout: fgvc_amd/csrc/corr_volume_f6.hip).
 *   fgvc_corr_volume_f16f6: q, k in that format -> vol [HWk][HWq] f32 = <k, q> / temperature. */
int fgvc_split_f16f6(const float* feat, uint8_t* out, int64_t n_pixels, int C, void* stream);
int fgvc_corr_volume_f16f6(const uint8_t* q_split, const uint8_t* k_split, int C, int HWq, int HWk,
                           float temperature, float* vol, void* stream);

/* Measurement helper (no reference counterpart): zeros over `n_floats` floats by a linear sweep of 16-byte stores (plain or non-temporal) --
 * the store ceiling bench.py prices fgvc_corr_volume_f16f6's write stream against; fgvc_set_option("corr6_debug", 1024) replays that
 * kernel's own store sequence without its loads and multiplies (zeros are written). */
int fgvc_debug_store_sweep_f32(float* buf, int64_t n_floats, int nontemporal, void* stream);

/* ---- A5, topk=None branch: weights over EVERY unmasked key instead of the k best
 * replaces local_attention.py:376-383 (`cur_affinity.softmax(dim=1)` / `.clamp(min=0)**2` over the (T*HWk x step) slab and the
 * einsum with value_vec).  Called once per key slot t with that slot's dense volume vol[HWk][HWq] (fgvc_corr_volume_*, already
 * divided by the temperature; -inf entries are skipped) and labels[HWk][P]:
 *   softmax mode: per query an online softmax state {running max, denominator, P weighted label sums} in
 *     state[nsplit][HWq][P+2] f32, `first` != 0 starts it, later calls merge into it;
 *   cosine mode (weight_mode = FGVC_WEIGHT_COSINE): plain sums of max(a,0)^2 * label;
 *   raw mode (FGVC_WEIGHT_RAW): plain sums of a * label over the unmasked keys (reference local_attention.py:38-103 with topk=None).
 * masked != 0 applies the predicate (r2max, ry, rx) on key - query offsets (equal grids required) and visits only the key
 * rows a band of queries can reach.  nsplit = fgvc_dense_attend_splits(HWq, HWk) (host helper) workgroups share a band's key rows.
 * fgvc_dense_attend_finish_f32 merges the splits and normalises: out[HWq][P]. P <= 32. */
int fgvc_dense_attend_splits(int HWq, int HWk);
int fgvc_dense_attend_f32(const float* vol, const float* labels, int Hq, int Wq, int Hk, int Wk, int P, int masked,
                          int r2max, int ry, int rx, int weight_mode, int first, float* state, int nsplit, void* stream);
int fgvc_dense_attend_finish_f32(const float* state, int nsplit, int HWq, int P, int weight_mode, float* out, void* stream);

/* ---- A5'': `propagate` (reference affinity_utils.py:33-50): new_img = img @ affinity for a GIVEN dense affinity aff[HWk][HWq]
 * (e.g. what fgvc_corr_volume_* wrote, softmaxed or not), labels [HWk][P] pixel-major, out [HWq][P], P <= 32 per call:
 *   thr == NULL   out[i][p] = sum_j aff[j][i] * labels[j][p]                                               (:45-49)
 *   thr != NULL   w = max(aff[j][i] - thr[i], 0);  out[i][p] = sum_j w * labels[j][p] / max(sum_j w, 1e-12)  (the `topk` branch, :36-44)
 * One streaming pass over the slab (the kernel of fgvc_dense_attend_f32); state = nsplit * HWq * (P + 2) floats of workspace,
 * nsplit = fgvc_dense_attend_splits(HWq, HWk).
 * fgvc_dense_kth_f32: thr[i] = k-th largest entry of column i of aff (1 <= k <= 64), one more pass; part = workspace of
 * nsplit * HWq * (k <= 16 ? 16 : 64) floats. */
int fgvc_dense_kth_f32(const float* aff, int HWk, int HWq, int k, float* part, int nsplit, float* thr, void* stream);
int fgvc_dense_propagate_f32(const float* aff, const float* labels, int HWk, int HWq, int P, const float* thr, float* state,
                             int nsplit, float* out, void* stream);

/* ---- A7: single-scale local-window correlation + top-k (mmcv.ops.Correlation semantics as used at
 * vanilla_tracker.py:435-443,547-566; torch twin local_attention.py:1190-1240).
 * Same kernel as fgvc_pair_topk_f32 with a square window |dy|,|dx| <= R; additionally the zero-padded
 * taps outside the image are candidates with score exactly 0.
 *   idx_out [HW][topk] = slot*(2R+1)^2 + (dy+R)*(2R+1) + (dx+R);  logit = corr/temperature (divided after
 *   top-k, :563); weight = softmax(logit). */
int fgvc_local_corr_topk_f32(const float* qfeat, const float* kfeat, const int32_t* pairs, int n_slots,
                             int C, int H, int W, int R, int topk, float temperature,
                             int32_t* pair_idx_ws, float* pair_score_ws,
                             int32_t* idx_out, float* logit_out, float* weight_out, void* stream);
/* the same on the f16 matrix pipe: features as written by fgvc_split_f16x2 from L2-NORMALISED rows, C == 256, topk <= 10
 * (fgvc_pair_topk_f16x3 with the square window, then the same merge) -- the default of the HR driver for C == 256 */
int fgvc_local_corr_topk_f16x3(const uint16_t* qsplit, const uint16_t* ksplit, const int32_t* pairs, int n_slots,
                               int C, int H, int W, int R, int topk, float temperature,
                               int32_t* pair_idx_ws, float* pair_score_ws,
                               int32_t* idx_out, float* logit_out, float* weight_out, void* stream);

/* ---- A7 get_coord (vanilla_tracker.py:445-488): expected image coordinate of every query pixel under the top-k
 * window weights of fgvc_local_corr_topk_f32 with ONE key slot (taps outside the grid contribute (0,0), like the
 * zero-padded F.unfold of the coordinate grid).  idx/weight [H*W][topk] -> out [H*W][2] = (x, y) in image pixels
 * (feature coordinate * scale). */
int fgvc_topk_coord_f32(const int32_t* idx, const float* weight, int H, int W, int R, int topk, int scale,
                        float* out, void* stream);

/* ---- A6: coarse-to-fine refine (local_attention.py:721-880), fine stage.
 *   coarse_arg [T][HW] int32: per key slot and query, the coarse cell picked by the coarse stage
 *                             (fgvc_pair_topk_f32 with topk=1 on the coarse features)
 *   qfine [sH*sW][Cf], kfine [T][sH*sW][Cf] normalised channels-last, vfine [T][sH*sW][P]
 *   out [HW][P]; idx_out/logit_out [HW][topk] (idx = t*(2Rf+1)^2 + tap) */
int fgvc_c2f_refine_f32(const int32_t* coarse_arg, const float* qfine, const float* kfine,
                        const float* vfine, int T, int H, int W, int scale, int Cf, int P, int Rf,
                        int topk, float temperature, float* out, int32_t* idx_out, float* logit_out,
                        void* stream);
/* ... with the weights of the reference's `mode`: FGVC_WEIGHT_SOFTMAX, or FGVC_WEIGHT_COSINE = clamp(affinity, 0)^2, not normalised
 * (local_attention.py:858-861) */
int fgvc_c2f_refine_mode_f32(const int32_t* coarse_arg, const float* qfine, const float* kfine, const float* vfine, int T,
                             int H, int W, int scale, int Cf, int P, int Rf, int topk, float temperature, int weight_mode,
                             float* out, int32_t* idx_out, float* logit_out, void* stream);

/* ---- A1 glue: inference BatchNorm2d (+ residual) (+ ReLU) fused over an NCHW activation
 * replaces the BN / `out += identity` / ReLU modules around every convolution of the ResNet
 * (resnet.py:54-116, mmcv ConvModule conv->BN->ReLU).  y = (x-mean[c])*rsqrt(var[c]+eps)*gamma[c]+beta[c]
 * [+ residual] [ReLU].  x, residual (nullable), out: [N][C][HW] f32; out may alias x. */
int fgvc_bn_act_f32(const float* x, const float* residual, const float* mean, const float* var,
                    const float* gamma, const float* beta, float eps, int relu, float* out, int N, int C,
                    int HW, void* stream);

/* ---- A1 on the bf16 matrix pipe: the encoder's stride-1 3x3 / 1x1 convolutions (resnet.py:16-116, mmcv ConvModule =
 * conv -> BN(eval) [-> + identity] [-> ReLU]) as an implicit GEMM over activations and weights stored as (hi, lo) bf16
 * pairs; hi*hi + hi*lo + lo*hi accumulate in f32 (error ~1e-7 relative, the size of f32 rounding noise).
 *   "padded split NHWC" activations: x[n][Hp][Wp][C/32][hi 32 ch | lo 32 ch] bf16, image at (1,1) inside a ZERO border
 *       that the caller provides once (the kernels only ever write the H x W interior);
 *       Hp >= 8*ceil(H/8)+2, Wp >= 32*ceil(W/32)+8.
 *   weights w[KS*KS][Cin/32][Cout][hi 32 ci | lo 32 ci] bf16 with BatchNorm folded in (w * gamma / sqrt(var + eps)),
 *       tap = ky*KS + kx; bias[Cout] = beta - mean * gamma / sqrt(var + eps)   (fgvc_amd/ops.py: prepare_conv_split).
 *   residual: NULL or dense NHWC f32 [n][H][W][Cout] (= a channels_last NCHW tensor, what MIOpen reads/writes without a
 *       layout conversion);  outputs (either may be NULL): y_split (padded split NHWC, the next convolution's input)
 *       and y_f32 (dense NHWC f32: residual of the next block / final features).
 * Cin % 32 == 0, Cout % 64 == 0, KS in {1, 3}, stride 1, zero padding KS/2. */
int fgvc_nchw_to_split_nhwc_f32(const float* in /* [N][C][H][W] */, uint16_t* out_split /* or NULL */,
                                float* out_f32 /* dense NHWC f32, or NULL */, int N, int C, int H, int W,
                                int Hp, int Wp, void* stream);
/* dense NHWC f32 x[n][H][W][C] -> padded split NHWC; relu != 0 applies max(x, 0) first and writes it back to x in place
 * (the tail of a MIOpen convolution with folded BatchNorm run on channels_last tensors) */
int fgvc_nhwc_to_split_f32(float* x, uint16_t* out_split, int N, int C, int H, int W, int Hp, int Wp, int relu,
                           void* stream);
int fgvc_conv_split_f32(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual,
                        uint16_t* y_split, float* y_f32, int N, int H, int W, int Hp, int Wp, int Cin, int Cout,
                        int KS, int relu, void* stream);
/* The same convolution with the operand format spelled out (round 3).  A split activation tensor keeps its shape and its 128 bytes
 * per (pixel, 32-channel chunk) in every format; FGVC_ACT_*:
 *   BF16X2  [hi 32 x bf16 | lo 32 x bf16], x = hi + lo                                      3 bf16 products per f32-grade product
 *   F16F8   [h 32 x f16 | l8 32 x e4m3 | h8 32 x e4m3]: h = f16(s x), l8 = e4m3(8 (s x - h)), h8 = e4m3(h / 128), s a per-tensor
 *           power of two (`*_scale_log2`) that puts the tensor's largest values around 2^8           2 products (f16 + one K-64 fp8 MFMA)
 *   F16X2   [h 32 x f16 | l 32 x f16], l = f16(s x - h)                                     3 f16 products, ~2^-22 per term
 *   F16F6   (round 4) [h 32 x f16 | 4 x 16 B]: the FP6 (e2m3) forms of the residual and of h, 32 elements each under one E8M0 scale:
 *           slot 4 = l6 bytes 0-15, slot 5 = h6 bytes 0-15, slot 6 = [l6 bytes 16-23 | scale byte | 0], slot 7 = the same for h6;
 *           l6 = e2m3(f16(2^11 (s x - h)) / 2^sl) with byte 127 + sl - 11, h6 = e2m3(h / 2^sh) with byte 127 + sh, 2^s the smallest
 *           power of two that keeps the block at or below 7.5; element e = channel e in bits [6 e, 6 e + 6)     1.5 products (f16 + one K-64
 *           FP6 MFMA at twice the fp8 rate).  Weights (ops.prepare_conv_split_f16): h6 in slots 4 / 6, l6 in 5 / 7.
 * `in_fmt` is the format of x AND of w (weights: ops.prepare_conv_split(fmt=...): F16F8 rows are [h | h8 = e4m3(h / 4) | l8 =
 * e4m3(512 l)] with h = f16(s_w w)); in_scale_log2 = log2(s_x s_w); out_fmt / out_scale_log2 describe y_split (what the NEXT
 * layer reads); *overflow (device word, required for an f16-format output) is OR-ed with 1 when |s_out y| exceeds 65504. */
#define FGVC_ACT_BF16X2 0
#define FGVC_ACT_F16F8 1
#define FGVC_ACT_F16X2 2
#define FGVC_ACT_F16F6 3
int fgvc_conv_split_fmt_f32(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual,
                            uint16_t* y_split, float* y_f32, int N, int H, int W, int Hp, int Wp, int Cin, int Cout,
                            int KS, int relu, int in_fmt, int in_scale_log2, int out_fmt, int out_scale_log2, int* overflow,
                            void* stream);
/* A BasicBlock's second convolution with the block's PROJECTION SHORTCUT folded in (round 4; resnet.py:96-106 `identity =
 * self.downsample(x) ... out += identity` for a stride-1 1 x 1 projection): y = conv3x3(x, w) + conv1x1(x2, w2) + bias [+ residual]
 * [ReLU], Cout = 256.  x2 is a second padded split NHWC tensor of the same H x W with Cin2 channels, w2 the [1][Cin2/32][256] rows
 * of the projection (BatchNorm folded; ops.prepare_conv_split*), in the SAME format as x / w and scaled so that s_x2 s_w2 = s_x s_w =
 * 2^in_scale_log2 (both products accumulate in one set of sums: the projection costs Cin2 / 32 extra stages of the 9 Cin / 32, its
 * own launch and the dense f32 copy of the identity disappear); bias = the sum of both folded biases. */
int fgvc_conv_split_proj_fmt_f32(const uint16_t* x, const uint16_t* w, const uint16_t* x2, const uint16_t* w2, const float* bias,
                                 const float* residual, uint16_t* y_split, float* y_f32, int N, int H, int W, int Hp, int Wp, int Cin,
                                 int Cin2, int relu, int in_fmt, int in_scale_log2, int out_fmt, int out_scale_log2, int* overflow,
                                 void* stream);
/* The trunk's LAST convolution writing the pair kernel's feature bank itself (round 4): fgvc_conv_split_fmt_f32 for Cout = 256, KS = 3 whose
 * epilogue -- + bias [+ residual] [ReLU] -- goes on to L2-normalise every pixel's 256 channels (normalize = 1: F.normalize(dim = C),
 * local_attention.py:312-318) and stores them as rows of fgvc_split_f16f6p: bank [N][H*W][1024 B], byte for byte what
 * fgvc_normalize_split_f16f6p_nhwc_f32 makes of the dense f32 output the plain entry point would have written (same association of
 * the sum of squares, same conversions) -- that output and the normalise pass's read of it never touch memory. */
int fgvc_conv_split_bank_f16f6p_f32(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual, void* bank,
                                    int N, int H, int W, int Hp, int Wp, int Cin, int KS, int relu, int in_fmt, int in_scale_log2,
                                    int normalize, void* stream);
/* ... as rows of fgvc_split_f16f6x: bank [N][H*W][2048 B], the normalised f32 channels in the second KiB (what the refining merge reads) */
int fgvc_conv_split_bank_f16f6x_f32(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual, void* bank,
                                    int N, int H, int W, int Hp, int Wp, int Cin, int KS, int relu, int in_fmt, int in_scale_log2,
                                    int normalize, void* stream);
/* fgvc_conv_split_f32 for Cin = Cout = 64, 3x3 (ResNet layer 1: the largest activations of the trunk), as persistent
 * workgroups that keep the folded weights in registers instead of re-streaming them per tile.  Same tensors and epilogue;
 * weights in MFMA-operand order:
 *   w[2 output tiles][9 taps][2 chunks][2 k-steps][hi | lo][lane = 32 * (k >> 3 & 1) + cout % 32][k & 7]   (ops.prepare_conv64). */
int fgvc_conv64_split_f32(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual, uint16_t* y_split,
                          float* y_f32, int N, int H, int W, int Hp, int Wp, int relu, void* stream);
/* ... with the residual (resnet.py:96-106 `identity = x ... out += identity`) given EITHER as dense NHWC f32 (`residual`) OR as a padded split NHWC
 * tensor of x's geometry (`residual_split`: the identity is then hi + lo, the value the block's first convolution multiplied;
 * the producer of the identity need not write an f32 copy of it).  At most one of the two. */
/* ... and with the operand formats of fgvc_conv_split_fmt_f32: in_fmt / out_fmt FGVC_ACT_BF16X2 or FGVC_ACT_F16F8 (the f16 + fp8
 * arithmetic of the wide layers: a third fewer matrix passes; weights from ops.prepare_conv64_f16: per (output tile, tap, chunk)
 * 4 KiB = [f16 k-step 0 | f16 k-step 1][64 lanes][16 B] then [64 lanes][h8 16 B | l8 16 B]); in_scale_log2 = log2(s_x s_w);
 * residual_split only with bf16x2 tensors. */
int fgvc_conv64_split_fmt_f32(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual,
                              const uint16_t* residual_split, uint16_t* y_split, float* y_f32, int N, int H, int W, int Hp, int Wp,
                              int relu, int in_fmt, int in_scale_log2, int out_fmt, int out_scale_log2, int* overflow, void* stream);
int fgvc_conv64_probe(int64_t* out32);   /* debug: s_memtime sums of one workgroup (conv64_variant = 8) */
int fgvc_conv64_split_res_f32(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual,
                              const uint16_t* residual_split, uint16_t* y_split, float* y_f32, int N, int H, int W, int Hp, int Wp,
                              int relu, void* stream);
/* The stride-2 members of the same family: the 3x3 / stride 2 / zero padding 1 convolution that opens a down-sampling
 * stage and its 1x1 / stride 2 projection (resnet.py:54-76 conv1 of the first BasicBlock, :288-296 downsample), BatchNorm
 * folded, + bias (+ ReLU).  x: padded split NHWC of the H x W input; outputs for the Ho x Wo = ((H-1)/2+1) x ((W-1)/2+1)
 * result: y_split (padded split NHWC, Hop x Wop) and / or y_f32 (dense NHWC f32); bias as for fgvc_conv_split_f32;
 * weights: the same folded (hi, lo) values in MFMA-operand order, one contiguous KiB per operand:
 *   w[KS*KS][Cin/32][Cout/32][hi k 0-15 | hi k 16-31 | lo k 0-15 | lo k 16-31][lane = 32 * (k >> 3 & 1) + cout % 32][k & 7]
 *   (k = input channel within the 32-channel chunk; fgvc_amd/ops.py: prepare_conv_s2).
 * Cin % 32 == 0, Cout % 32 == 0, KS in {1, 3}. */
int fgvc_conv_s2_split_f32(const uint16_t* x, const uint16_t* w, const float* bias, uint16_t* y_split, float* y_f32, int N,
                           int H, int W, int Hp, int Wp, int Cin, int Cout, int KS, int Hop, int Wop, int relu,
                           void* stream);
/* ... with y_split in any FGVC_ACT_* format (x and w stay BF16X2: these layers read layer 1's output) */
int fgvc_conv_s2_split_fmt_f32(const uint16_t* x, const uint16_t* w, const float* bias, uint16_t* y_split, float* y_f32, int N,
                               int H, int W, int Hp, int Wp, int Cin, int Cout, int KS, int Hop, int Wop, int relu, int out_fmt,
                               int out_scale_log2, int* overflow, void* stream);
/* The stem: 7x7 / stride 2 / zero padding 3 convolution of 3-channel frames to 64 channels, BatchNorm folded, + bias
 * (+ ReLU) (resnet.py:457-466 with pool_type 'none'), from the f32 NCHW frames x[N][3][H][W] to y_f32 (dense NHWC f32
 * [N][Ho][Wo][64]) and / or y_split (padded split NHWC, Hop x Wop); Ho x Wo = ((H-1)/2+1) x ((W-1)/2+1).
 *   w: folded (hi, lo) bf16 weights in MFMA-operand order, K laid out per kernel row as k = 4 kx + c (c = 3 and kx = 7: zero):
 *      w[7 ky][2 k-steps][2 output tiles][hi | lo][lane = 32 * (k >> 3 & 1) + cout % 32][k & 7]   (ops.prepare_stem7);
 *   bias[64] = beta - mean * gamma / sqrt(var + eps). */
int fgvc_stem7_split_f32(const float* x, const uint16_t* w, const float* bias, uint16_t* y_split, float* y_f32, int N, int H,
                         int W, int Hop, int Wop, int relu, void* stream);
/* ... with the split output in the format the first block's kernel reads (FGVC_ACT_BF16X2 or FGVC_ACT_F16F8 at scale 2^out_scale_log2) */
int fgvc_stem7_split_fmt_f32(const float* x, const uint16_t* w, const float* bias, uint16_t* y_split, float* y_f32, int N, int H,
                             int W, int Hop, int Wop, int relu, int out_fmt, int out_scale_log2, int* overflow, void* stream);
/* dense NHWC f32 -> [n][H*W][C] f32, rows L2-normalised if `normalize` (the output layout of
 * fgvc_normalize_chw_to_hwc_f32) */
int fgvc_normalize_nhwc_f32(const float* in, float* out, int N, int C, int H, int W, int normalize, void* stream);
/* the same rows and / or their (hi, lo) bf16 split [n][H*W][hi C | lo C] (= fgvc_split_bf16 of them, what
 * fgvc_corr_volume_bf16x3 reads) in ONE pass over the trunk output; either output may be NULL */
int fgvc_normalize_split_nhwc_f32(const float* in, float* out_f32, uint16_t* out_split, int N, int C, int H, int W,
                                  int normalize, void* stream);
/* the same with the split in the (h, l) f16 form of fgvc_split_f16x2 (what fgvc_pair_topk_f16x3 reads) */
int fgvc_normalize_split_f16x2_nhwc_f32(const float* in, float* out_f32, uint16_t* out_split, int N, int C, int H, int W,
                                        int normalize, void* stream);
/* ... and straight into the rows of fgvc_split_f16f6p (what fgvc_pair_topk_f16f6 reads): rows [N][H*W][1024] bytes, C == 256 */
int fgvc_normalize_split_f16f6p_nhwc_f32(const float* in, uint8_t* rows, int N, int C, int H, int W, int normalize, void* stream);
/* ... and into the 2 KiB rows of fgvc_split_f16f6x (the same 1 KiB + the normalised f32 channels) */
int fgvc_normalize_split_f16f6x_nhwc_f32(const float* in, uint8_t* rows, int N, int C, int H, int W, int normalize, void* stream);

/* ---- A3: initial labels  g = exp(-((x*s-cx)^2+(y*s-cy)^2)/(2 sigma^2)) on the feature grid
 * replaces vanilla_tracker.py:204-221 ([::stride] subsample of the full-resolution Gaussian).
 *   points [P][2] f32 = (x, y);  out [Hf*Wf][P] */
int fgvc_gaussian_labels_f32(const float* points, int P, int Hf, int Wf, int stride, float sigma,
                             float* out, void* stream);

/* ---- A8 + A9: bilinear upsample (align_corners=False) fused with the top-5 soft-argmax read-out
 * replaces vanilla_tracker.py:396-400 and :172-191 (no (T,P,h,w) tensor, no D2H copy, no argsort).
 *   labels [n_frames][Hf*Wf][P];  gauss_points: if non-NULL, frame 0 is read out from the analytic
 *   full-resolution Gaussian of these points instead (vanilla_tracker.py:329,322).
 *   coords [n_frames][P][2] f64 = (x, y); (-1,-1) where the map is all zero.
 *   workspace: caller-owned device scratch of fgvc_softargmax_workspace_bytes(n_frames, P) bytes.
 *   A bilinear sample never exceeds the largest of its four coarse corners, so only the coarse cells whose corner maximum
 *   reaches the 5th largest value found around the coarse maximum are upsampled (exactly the pixels a full scan would
 *   select from); maps with negative labels or too flat for the work lists are scanned in full, in row bands. */
size_t fgvc_softargmax_workspace_bytes(int n_frames, int P);
int fgvc_softargmax_top5_f32(const float* labels, int n_frames, int Hf, int Wf, int P, int h, int w,
                             const float* gauss_points, float sigma, double* coords, void* workspace,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FGVC_HIP_H */
