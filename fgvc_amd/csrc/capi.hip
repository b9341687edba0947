// extern "C" surface of libfgvc_hip.so (see include/fgvc_hip.h): argument validation, then launch.
// Shapes are validated HERE, on the host, before any hand-written kernel is enqueued: a grid that
// does not match its operands can fault the GPU.
#include <stdarg.h>
#include <string.h>

#include "common.hpp"

namespace fgvc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int pair_topk_launch(const float*, const float*, const int32_t*, int, int, int, int, int, int, int, int, int, int,
                     const uint8_t*, int32_t*, float*, hipStream_t);
int merge_topk_launch(const int32_t*, const float*, const int32_t*, int, int, int, int, int, float, int, int32_t*,
                      float*, float*, hipStream_t);
int normalize_launch(const float*, float*, int, int, int, int, int, hipStream_t);
int propagate_launch(const float*, const int32_t*, int, const int32_t*, const float*, int, int, int, int, int, int,
                     int, float*, hipStream_t);
int bn_act_launch(const float*, const float*, const float*, const float*, const float*, const float*, float, int, float*,
                  int, int, int, hipStream_t);
int gaussian_launch(const float*, int, int, int, int, float, float*, hipStream_t);
int softargmax_launch(const float*, int, int, int, int, int, int, const float*, float, double*, float*, hipStream_t);
int softargmax_bands();
int corr_volume_f32_launch(const float*, const float*, int, int, int, float, float*, hipStream_t);
int split_bf16_launch(const float*, uint16_t*, long long, int, hipStream_t);
int corr_volume_bf16_launch(const uint16_t*, const uint16_t*, int, int, int, float, float*, int, hipStream_t);
int local_merge_launch(const int32_t*, const float*, int, int, int, int, int, float, int32_t*, float*, float*,
                       hipStream_t);
int topk_coord_launch(const int32_t*, const float*, int, int, int, int, int, float*, hipStream_t);
int c2f_refine_launch(const int32_t*, const float*, const float*, const float*, int, int, int, int, int, int, int,
                      int, float, int, float*, int32_t*, float*, hipStream_t);

int conv_split_launch(const uint16_t*, const uint16_t*, const float*, const float*, uint16_t*, float*, int, int, int, int, int,
                      int, int, int, int, int, int, int, int, int*, hipStream_t, unsigned char* y_bank = nullptr, int bank_normalize = 1, const uint16_t* x2 = nullptr,
                      const uint16_t* w2 = nullptr, int Cin2 = 0, int bank_row_bytes = 1024);
int conv_s2_launch(const uint16_t*, const uint16_t*, const float*, uint16_t*, float*, int, int, int, int, int, int, int, int, int,
                   int, int, int, int, int*, hipStream_t);
int conv64_launch(const uint16_t*, const uint16_t*, const float*, const float*, const uint16_t*, uint16_t*, float*, int, int, int, int, int, int,
                  int, int, int, int, int*, hipStream_t);
int stem7_launch(const float*, const uint16_t*, const float*, uint16_t*, float*, int, int, int, int, int, int, int, int, int, int, int*,
                 hipStream_t);
int nchw_to_split_nhwc_launch(const float*, uint16_t*, float*, int, int, int, int, int, int, hipStream_t);
int normalize_nhwc_launch(const float*, float*, uint16_t*, int, int, int, int, int, int, hipStream_t);
int nhwc_to_split_launch(float*, uint16_t*, int, int, int, int, int, int, int, hipStream_t);

void set_conv_cot_cap(int);
void set_conv_narrow(int);
void set_conv64_variant(int);
int conv64_probe_read(long long*);
void set_conv_debug(int);
void set_corr6_skew(int);
void set_corr6_sdma(int);
void set_pair_debug(int);
void set_readout_prune(int);
void set_conv_s2_debug(int);
int split_f16x2_launch(const float*, uint16_t*, long long, int, hipStream_t);
int split_f16f6p_launch(const float*, unsigned char*, long long, int, hipStream_t);
int pair_topk_v7_launch(const uint16_t*, const uint16_t*, const int32_t*, int, int, int, int, int, int, int, int, int, const int32_t*, int,
                        int32_t*, float*, int, hipStream_t);
int pair_topk_v5_launch(const uint16_t*, const uint16_t*, const int32_t*, int, int, int, int, int, int, int, int, int, int, const int32_t*,
                        int, int32_t*, float*, hipStream_t);
void set_pair_v5_debug(int);
int pair_v5_timeout_flag();
int pair_v5_probe_read(long long*);
void set_corr_debug(int);
int split_f16f8_launch(const float*, unsigned char*, long long, int, hipStream_t);
int corr_volume_f16f8_launch(const unsigned char*, const unsigned char*, int, int, float, float*, hipStream_t);
void set_corr8_debug(int);
int split_f16f6_launch(const float*, unsigned char*, long long, hipStream_t);
int corr_volume_f16f6_launch(const unsigned char*, const unsigned char*, int, int, float, float*, hipStream_t);
void set_corr6_debug(int);
int store_sweep_launch(float*, long long, int, hipStream_t);
int dense_attend_splits(int, int);
int dense_attend_launch(const float*, const float*, int, int, int, int, int, int, int, int, int, int, int, float*, int, hipStream_t, const float*);
int dense_kth_launch(const float*, int, int, int, float*, int, float*, hipStream_t);
int dense_attend_finish_launch(const float*, int, int, int, int, float*, hipStream_t);

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace fgvc

using namespace fgvc;

extern "C" {

const char* fgvc_version(void) { return "fgvc_hip 0.1 (gfx950)"; }
const char* fgvc_last_error(void) { return g_err; }

// The profiling ablations (a kernel without its stores, its matrix chain, its selection ...) return WRONG results.  The production
// library refuses to switch them on: the bits below are accepted only by the build with -DFGVC_ABLATIONS (libfgvc_hip_ablations.so,
// `python -m fgvc_amd.build` makes both from the same objects but this file; tools/experiments select it with FGVC_HIP_LIB).  What
// the production library still takes are A/B switches between forms with IDENTICAL results, the s_memtime probes, and the fault
// injection of the fail-closed test (poison lists + the device flag: not a result anybody can mistake for one).
static int ablation_bits(const char* name, int value) {
  if (strcmp(name, "corr_debug") == 0 || strcmp(name, "pair_debug") == 0 || strcmp(name, "conv_s2_debug") == 0) return value;
  if (strcmp(name, "corr8_debug") == 0) return value & (1 | 2);
  if (strcmp(name, "corr6_debug") == 0) return value & (1 | 2 | 256 | 512 | 1024);
  if (strcmp(name, "conv_debug") == 0) return value & (1 | 2 | 4);
  if (strcmp(name, "pair_f16_debug") == 0)
    return value & ((value & 4194304) ? (1 | 2 | 524288 | 1048576) : (1 | 2 | 4 | 8 | 16 | 32 | 64 | 2048 | 524288 | 1048576));
  return 0;
}

int fgvc_set_option(const char* name, int value) {
  FGVC_REQUIRE(name != nullptr, FGVC_ERR_INVALID_ARG, "fgvc_set_option: null name");
#ifndef FGVC_ABLATIONS
  FGVC_REQUIRE(ablation_bits(name, value) == 0, FGVC_ERR_UNSUPPORTED,
               "fgvc_set_option: %s = %d switches a profiling ablation on (bits %d: results would be wrong); this library is built without them -- "
               "use libfgvc_hip_ablations.so (FGVC_HIP_LIB)", name, value, ablation_bits(name, value));
#endif
  if (strcmp(name, "corr_debug") == 0) {   // profiling ablation: 1 = bf16 volume kernels skip their stores
    set_corr_debug(value);
    return FGVC_OK;
  }
  if (strcmp(name, "corr8_debug") == 0) {   // fgvc_corr_volume_f16f8 ablations: 1 = no stores, 2 = no MFMA (results wrong); 4 = no row
    set_corr8_debug(value);                 // classes, 8 = wave stagger, 16 = the 32x32-shape kernel (results right); >> 8 = key blocks per workgroup
    return FGVC_OK;
  }
  if (strcmp(name, "corr6_debug") == 0) {   // fgvc_corr_volume_f16f6 ablations: 1 = no stores, 2 = no MFMA, 4 = no row classes, 8 = no stagger, ...; >> 12 = half-chunks per tile pair
    set_corr6_debug(value);
    return FGVC_OK;
  }
  if (strcmp(name, "pair_debug") == 0) {   // profiling ablations; results are wrong when non-zero
    set_pair_debug(value);
    return FGVC_OK;
  }
  if (strcmp(name, "corr6_sdma") == 0) {   // A/B: staging DMAs of fgvc_corr_volume_f16f6 with a scalar base (1) or 64-bit lane addresses (0)
    set_corr6_sdma(value);
    return FGVC_OK;
  }
  if (strcmp(name, "corr6_skew") == 0) {   // fgvc_corr_volume_f16f6: stages taken from the two-segment piece and given to the others
    set_corr6_skew(value);
    return FGVC_OK;
  }
  if (strcmp(name, "conv_debug") == 0) {   // profiling ablations of fgvc_conv_split_f32; results are wrong when non-zero
    set_conv_debug(value);
    return FGVC_OK;
  }
  if (strcmp(name, "conv64_variant") == 0) {   // A/B bits of fgvc_conv64_split_f32 (csrc/conv64.hip)
    set_conv64_variant(value);
    return FGVC_OK;
  }
  if (strcmp(name, "conv_narrow") == 0) {   // fgvc_conv_split_f32 with 64 output channels per workgroup: 1 (default) = 4-row tiles,
    set_conv_narrow(value);                 // two workgroups per CU;  0 = 8-row tiles, one workgroup per CU;  +2 = the same for 128
    return FGVC_OK;
  }
  if (strcmp(name, "conv_cot_cap") == 0) {   // fgvc_conv_split_f32: at most this many output channels per workgroup (0, 64, 128)
    FGVC_REQUIRE(value == 0 || value == 64 || value == 128, FGVC_ERR_INVALID_ARG, "fgvc_set_option: conv_cot_cap must be 0, 64 or 128");
    set_conv_cot_cap(value);
    return FGVC_OK;
  }
  if (strcmp(name, "pair_f16_debug") == 0) {   // fgvc_pair_topk_f16x3 ablations (results wrong): 1 = no selection, 2 = no MFMA, 4 = no staging,
    set_pair_v5_debug(value);                  // 16 = prologue only, 32 = no epilogue, 64 = no main loop
    return FGVC_OK;
  }
  if (strcmp(name, "conv_s2_debug") == 0) {   // profiling ablations of fgvc_conv_s2_split_f32; results are wrong when non-zero
    set_conv_s2_debug(value);
    return FGVC_OK;
  }
  if (strcmp(name, "readout_prune") == 0) {   // fgvc_softargmax_top5_f32: 1 (default) = pruned read-out first, full scan only for
    set_readout_prune(value != 0);            // the maps it hands back;  0 = full scan of every map (same results)
    return FGVC_OK;
  }
  set_error("fgvc_set_option: unknown option '%s'", name);
  return FGVC_ERR_INVALID_ARG;
}

int fgvc_r2max_for_radius(float radius) {
  if (!(radius > 0.f)) return -1;
  long long d2 = (long long)ceil((double)radius * (double)radius) + 2;
  if (d2 > FGVC_NO_LIMIT) return FGVC_NO_LIMIT;
  while (d2 >= 0 && !(sqrtf((float)d2) < radius)) --d2;
  return (int)d2;
}

int fgvc_normalize_chw_to_hwc_f32(const float* in, float* out, int n, int C, int HW, int normalize, int c_out,
                                  void* stream) {
  FGVC_REQUIRE(in && out, FGVC_ERR_INVALID_ARG, "fgvc_normalize_chw_to_hwc_f32: null pointer");
  FGVC_REQUIRE(n >= 0 && C > 0 && HW > 0, FGVC_ERR_INVALID_ARG, "fgvc_normalize_chw_to_hwc_f32: bad shape n=%d C=%d HW=%d", n, C, HW);
  FGVC_REQUIRE(C <= 1024, FGVC_ERR_UNSUPPORTED, "fgvc_normalize_chw_to_hwc_f32: C=%d > 1024 (LDS tile)", C);
  FGVC_REQUIRE(n <= 65535, FGVC_ERR_UNSUPPORTED, "fgvc_normalize_chw_to_hwc_f32: n=%d > 65535 frames per call", n);
  FGVC_REQUIRE(c_out >= C, FGVC_ERR_INVALID_ARG, "fgvc_normalize_chw_to_hwc_f32: c_out=%d < C=%d", c_out, C);
  if (n == 0) return FGVC_OK;
  return normalize_launch(in, out, n, C, HW, normalize, c_out, (hipStream_t)stream);
}

int fgvc_pair_topk_f32(const float* qfeat, const float* kfeat, const int32_t* pairs, int n_pairs, int C, int Hq,
                       int Wq, int Hk, int Wk, int r2max, int ry, int rx, int topk, const uint8_t* dense_mask,
                       int32_t* idx_out, float* score_out, void* stream) {
  FGVC_REQUIRE(qfeat && kfeat && pairs && idx_out && score_out, FGVC_ERR_INVALID_ARG, "fgvc_pair_topk_f32: null pointer");
  FGVC_REQUIRE(aligned16(qfeat) && aligned16(kfeat) && aligned16(pairs), FGVC_ERR_INVALID_ARG,
               "fgvc_pair_topk_f32: qfeat/kfeat/pairs must be 16-byte aligned");
  FGVC_REQUIRE(Hq > 0 && Wq > 0 && Hk > 0 && Wk > 0 && n_pairs >= 0, FGVC_ERR_INVALID_ARG,
               "fgvc_pair_topk_f32: bad shape Hq=%d Wq=%d Hk=%d Wk=%d n_pairs=%d", Hq, Wq, Hk, Wk, n_pairs);
  FGVC_REQUIRE(topk >= 1 && topk <= 16, FGVC_ERR_UNSUPPORTED, "fgvc_pair_topk_f32: topk=%d outside 1..16", topk);
  FGVC_REQUIRE(r2max >= 0 && ry >= 0 && rx >= 0, FGVC_ERR_INVALID_ARG, "fgvc_pair_topk_f32: negative mask parameter");
  FGVC_REQUIRE(n_pairs <= 65535, FGVC_ERR_UNSUPPORTED, "fgvc_pair_topk_f32: n_pairs=%d > 65535 per call", n_pairs);
  const bool any_limit = r2max < FGVC_NO_LIMIT || ry < FGVC_NO_LIMIT || rx < FGVC_NO_LIMIT;
  FGVC_REQUIRE(!any_limit || (Hq == Hk && Wq == Wk), FGVC_ERR_INVALID_ARG,
               "fgvc_pair_topk_f32: a spatial mask needs equal query/key grids (local_attention.py:331)");
  FGVC_REQUIRE((long long)Hk * Wk < (1ll << 30) && (long long)Hq * Wq < (1ll << 30), FGVC_ERR_UNSUPPORTED,
               "fgvc_pair_topk_f32: grid too large");
  if (n_pairs == 0) return FGVC_OK;
  FGVC_REQUIRE(dense_mask == nullptr || !any_limit, FGVC_ERR_INVALID_ARG,
               "fgvc_pair_topk_f32: give either the analytic predicate or a dense mask, not both");
  return pair_topk_launch(qfeat, kfeat, pairs, n_pairs, C, Hq, Wq, Hk, Wk, r2max, ry, rx, topk, dense_mask, idx_out,
                          score_out, (hipStream_t)stream);
}

int fgvc_split_f16x2(const float* feat, uint16_t* h_l, int64_t n_pixels, int C, void* stream) {
  FGVC_REQUIRE(feat && h_l, FGVC_ERR_INVALID_ARG, "fgvc_split_f16x2: null pointer");
  FGVC_REQUIRE(n_pixels >= 0 && C > 0 && C % 4 == 0, FGVC_ERR_INVALID_ARG, "fgvc_split_f16x2: C must be a multiple of 4");
  FGVC_REQUIRE(aligned16(feat) && aligned16(h_l), FGVC_ERR_INVALID_ARG, "fgvc_split_f16x2: 16-byte alignment required");
  if (n_pixels == 0) return FGVC_OK;
  return split_f16x2_launch(feat, h_l, n_pixels, C, (hipStream_t)stream);
}

static int pair_f16x3_common(const char* what, const uint16_t* qsplit, const uint16_t* ksplit, const int32_t* pairs, int n_pairs, int C,
                             int Hq, int Wq, int Hk, int Wk, int r2max, int ry, int rx, int topk, int all_masked, const int32_t* runs,
                             int n_runs, int32_t* idx_out, float* score_out, void* stream) {
  FGVC_REQUIRE(qsplit && ksplit && pairs && idx_out && score_out, FGVC_ERR_INVALID_ARG, "%s: null pointer", what);
  FGVC_REQUIRE(aligned16(qsplit) && aligned16(ksplit) && aligned16(pairs), FGVC_ERR_INVALID_ARG,
               "%s: qsplit/ksplit/pairs must be 16-byte aligned", what);
  FGVC_REQUIRE(C == 256, FGVC_ERR_UNSUPPORTED, "%s: C=%d unsupported (256 only; use fgvc_pair_topk_f32)", what, C);
  FGVC_REQUIRE(Hq > 0 && Wq > 0 && Hk > 0 && Wk > 0 && n_pairs >= 0, FGVC_ERR_INVALID_ARG,
               "%s: bad shape Hq=%d Wq=%d Hk=%d Wk=%d n_pairs=%d", what, Hq, Wq, Hk, Wk, n_pairs);
  FGVC_REQUIRE(topk >= 1 && topk <= 10, FGVC_ERR_UNSUPPORTED, "%s: topk=%d outside 1..10", what, topk);
  FGVC_REQUIRE(r2max >= 0 && ry >= 0 && rx >= 0, FGVC_ERR_INVALID_ARG, "%s: negative mask parameter", what);
  FGVC_REQUIRE(n_pairs <= 65535, FGVC_ERR_UNSUPPORTED, "%s: n_pairs=%d > 65535 per call", what, n_pairs);
  FGVC_REQUIRE(!runs || (n_runs >= 1 && n_runs <= n_pairs && (reinterpret_cast<uintptr_t>(runs) & 7u) == 0), FGVC_ERR_INVALID_ARG,
               "%s: runs must be 8-byte aligned, 1 <= n_runs <= n_pairs", what);
  const bool any_limit = r2max < FGVC_NO_LIMIT || ry < FGVC_NO_LIMIT || rx < FGVC_NO_LIMIT;
  FGVC_REQUIRE(!any_limit || (Hq == Hk && Wq == Wk), FGVC_ERR_INVALID_ARG,
               "%s: a spatial mask needs equal query/key grids (local_attention.py:331)", what);
  FGVC_REQUIRE(Hk < 32768 && Wk < 32768 && Hq < 32768 && Wq < 32768 && (long long)Hk * Wk < (1ll << 30) &&
                   (long long)Hq * Wq < (1ll << 30),
               FGVC_ERR_UNSUPPORTED, "%s: grid too large", what);
  if (n_pairs == 0) return FGVC_OK;
  return pair_topk_v5_launch(qsplit, ksplit, pairs, n_pairs, Hq, Wq, Hk, Wk, r2max, ry, rx, topk, all_masked != 0 && any_limit, runs,
                             n_runs, idx_out, score_out, (hipStream_t)stream);
}

int fgvc_pair_topk_f16x3(const uint16_t* qsplit, const uint16_t* ksplit, const int32_t* pairs, int n_pairs, int C, int Hq,
                         int Wq, int Hk, int Wk, int r2max, int ry, int rx, int topk, int all_masked, int32_t* idx_out,
                         float* score_out, void* stream) {
  return pair_f16x3_common("fgvc_pair_topk_f16x3", qsplit, ksplit, pairs, n_pairs, C, Hq, Wq, Hk, Wk, r2max, ry, rx, topk, all_masked,
                           nullptr, 0, idx_out, score_out, stream);
}

int fgvc_pair_topk_f16x3_runs(const uint16_t* qsplit, const uint16_t* ksplit, const int32_t* pairs, int n_pairs, int C, int Hq,
                              int Wq, int Hk, int Wk, int r2max, int ry, int rx, int topk, int all_masked, const int32_t* runs,
                              int n_runs, int32_t* idx_out, float* score_out, void* stream) {
  FGVC_REQUIRE(runs, FGVC_ERR_INVALID_ARG, "fgvc_pair_topk_f16x3_runs: null runs");
  return pair_f16x3_common("fgvc_pair_topk_f16x3_runs", qsplit, ksplit, pairs, n_pairs, C, Hq, Wq, Hk, Wk, r2max, ry, rx, topk, all_masked,
                           runs, n_runs, idx_out, score_out, stream);
}

int fgvc_split_f16f6p(const float* feat, uint8_t* rows, int64_t n_pixels, int C, void* stream) {
  FGVC_REQUIRE(feat && rows, FGVC_ERR_INVALID_ARG, "fgvc_split_f16f6p: null pointer");
  FGVC_REQUIRE(C == 256, FGVC_ERR_UNSUPPORTED, "fgvc_split_f16f6p: C=%d unsupported (256 only)", C);
  FGVC_REQUIRE(n_pixels >= 0 && n_pixels < (1ll << 40), FGVC_ERR_INVALID_ARG, "fgvc_split_f16f6p: bad n_pixels");
  FGVC_REQUIRE(aligned16(feat) && aligned16(rows), FGVC_ERR_INVALID_ARG, "fgvc_split_f16f6p: 16-byte alignment required");
  if (n_pixels == 0) return FGVC_OK;
  return split_f16f6p_launch(feat, rows, n_pixels, 1024, (hipStream_t)stream);
}

int fgvc_split_f16f6x(const float* feat, uint8_t* rows, int64_t n_pixels, int C, void* stream) {
  FGVC_REQUIRE(feat && rows, FGVC_ERR_INVALID_ARG, "fgvc_split_f16f6x: null pointer");
  FGVC_REQUIRE(C == 256, FGVC_ERR_UNSUPPORTED, "fgvc_split_f16f6x: C=%d unsupported (256 only)", C);
  FGVC_REQUIRE(n_pixels >= 0 && n_pixels < (1ll << 40), FGVC_ERR_INVALID_ARG, "fgvc_split_f16f6x: bad n_pixels");
  FGVC_REQUIRE(aligned16(feat) && aligned16(rows), FGVC_ERR_INVALID_ARG, "fgvc_split_f16f6x: 16-byte alignment required");
  if (n_pixels == 0) return FGVC_OK;
  return split_f16f6p_launch(feat, rows, n_pixels, 2048, (hipStream_t)stream);
}

static int pair_f16f6_common(const char* what, const uint8_t* qsplit, const uint8_t* ksplit, const int32_t* pairs, int n_pairs, int C,
                             int Hq, int Wq, int Hk, int Wk, int r2max, int ry, int rx, int topk, int all_masked, const int32_t* runs,
                             int n_runs, int32_t* idx_out, float* score_out, void* stream, int row_bytes = 1024) {
  FGVC_REQUIRE(qsplit && ksplit && pairs && idx_out && score_out, FGVC_ERR_INVALID_ARG, "%s: null pointer", what);
  FGVC_REQUIRE(aligned16(qsplit) && aligned16(ksplit) && aligned16(pairs), FGVC_ERR_INVALID_ARG,
               "%s: qsplit/ksplit/pairs must be 16-byte aligned", what);
  FGVC_REQUIRE(C == 256, FGVC_ERR_UNSUPPORTED, "%s: C=%d unsupported (256 only; use fgvc_pair_topk_f32)", what, C);
  FGVC_REQUIRE(Hq > 0 && Wq > 0 && Hk > 0 && Wk > 0 && n_pairs >= 0, FGVC_ERR_INVALID_ARG,
               "%s: bad shape Hq=%d Wq=%d Hk=%d Wk=%d n_pairs=%d", what, Hq, Wq, Hk, Wk, n_pairs);
  FGVC_REQUIRE(topk >= 1 && topk <= 10, FGVC_ERR_UNSUPPORTED, "%s: topk=%d outside 1..10", what, topk);
  FGVC_REQUIRE(r2max >= 0 && ry >= 0 && rx >= 0, FGVC_ERR_INVALID_ARG, "%s: negative mask parameter", what);
  FGVC_REQUIRE(n_pairs <= 65535, FGVC_ERR_UNSUPPORTED, "%s: n_pairs=%d > 65535 per call", what, n_pairs);
  FGVC_REQUIRE(!runs || (n_runs >= 1 && n_runs <= n_pairs && (reinterpret_cast<uintptr_t>(runs) & 7u) == 0), FGVC_ERR_INVALID_ARG,
               "%s: runs must be 8-byte aligned, 1 <= n_runs <= n_pairs", what);
  const bool any_limit = r2max < FGVC_NO_LIMIT || ry < FGVC_NO_LIMIT || rx < FGVC_NO_LIMIT;
  FGVC_REQUIRE(any_limit && all_masked != 0, FGVC_ERR_UNSUPPORTED,
               "%s: every pair must carry FGVC_PAIR_MASKED under an analytic mask (a pair that scans the whole frame needs fgvc_pair_topk_f16x3)", what);
  FGVC_REQUIRE(Hq == Hk && Wq == Wk, FGVC_ERR_INVALID_ARG, "%s: a spatial mask needs equal query/key grids (local_attention.py:331)", what);
  FGVC_REQUIRE(Hk < 16384 && Wk < 32768 && (long long)Hk * Wk < (1ll << 30), FGVC_ERR_UNSUPPORTED, "%s: grid too large", what);
  if (n_pairs == 0) return FGVC_OK;
  return pair_topk_v7_launch(reinterpret_cast<const uint16_t*>(qsplit), reinterpret_cast<const uint16_t*>(ksplit), pairs, n_pairs, Hq, Wq, Hk, Wk,
                             r2max, ry, rx, topk, runs, n_runs, idx_out, score_out, row_bytes, (hipStream_t)stream);
}

int fgvc_pair_topk_f16f6(const uint8_t* qsplit, const uint8_t* ksplit, const int32_t* pairs, int n_pairs, int C, int Hq, int Wq, int Hk,
                         int Wk, int r2max, int ry, int rx, int topk, int all_masked, int32_t* idx_out, float* score_out, void* stream) {
  return pair_f16f6_common("fgvc_pair_topk_f16f6", qsplit, ksplit, pairs, n_pairs, C, Hq, Wq, Hk, Wk, r2max, ry, rx, topk, all_masked,
                           nullptr, 0, idx_out, score_out, stream);
}

int fgvc_pair_topk_f16f6_runs(const uint8_t* qsplit, const uint8_t* ksplit, const int32_t* pairs, int n_pairs, int C, int Hq, int Wq,
                              int Hk, int Wk, int r2max, int ry, int rx, int topk, int all_masked, const int32_t* runs, int n_runs,
                              int32_t* idx_out, float* score_out, void* stream) {
  FGVC_REQUIRE(runs, FGVC_ERR_INVALID_ARG, "fgvc_pair_topk_f16f6_runs: null runs");
  return pair_f16f6_common("fgvc_pair_topk_f16f6_runs", qsplit, ksplit, pairs, n_pairs, C, Hq, Wq, Hk, Wk, r2max, ry, rx, topk, all_masked,
                           runs, n_runs, idx_out, score_out, stream);
}

int fgvc_pair_topk_f16f6x(const uint8_t* qrows, const uint8_t* krows, const int32_t* pairs, int n_pairs, int C, int Hq, int Wq, int Hk,
                          int Wk, int r2max, int ry, int rx, int topk, int all_masked, int32_t* idx_out, float* score_out, void* stream) {
  return pair_f16f6_common("fgvc_pair_topk_f16f6x", qrows, krows, pairs, n_pairs, C, Hq, Wq, Hk, Wk, r2max, ry, rx, topk, all_masked,
                           nullptr, 0, idx_out, score_out, stream, 2048);
}

int fgvc_pair_topk_f16f6x_runs(const uint8_t* qrows, const uint8_t* krows, const int32_t* pairs, int n_pairs, int C, int Hq, int Wq,
                               int Hk, int Wk, int r2max, int ry, int rx, int topk, int all_masked, const int32_t* runs, int n_runs,
                               int32_t* idx_out, float* score_out, void* stream) {
  FGVC_REQUIRE(runs, FGVC_ERR_INVALID_ARG, "fgvc_pair_topk_f16f6x_runs: null runs");
  return pair_f16f6_common("fgvc_pair_topk_f16f6x_runs", qrows, krows, pairs, n_pairs, C, Hq, Wq, Hk, Wk, r2max, ry, rx, topk, all_masked,
                           runs, n_runs, idx_out, score_out, stream, 2048);
}

/* debug: the 32 s_memtime words one workgroup leaves with fgvc_set_option("pair_f16_debug", 256) (tools/experiments/time_pair_v5.py) */
int fgvc_pair_topk_f16x3_probe(int64_t* out32) {
  if (!out32 || hipDeviceSynchronize() != hipSuccess) return FGVC_ERR_INVALID_ARG;
  return pair_v5_probe_read(reinterpret_cast<long long*>(out32)) == 0 ? FGVC_OK : FGVC_ERR_LAUNCH;
}

int fgvc_pair_topk_f16x3_timed_out(void) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return pair_v5_timeout_flag();
}

int fgvc_merge_topk_f32(const int32_t* pair_idx, const float* pair_score, const int32_t* slot_pair, int n_out, int T,
                        int HWq, int HWk, int topk, float temperature, int weight_mode, int32_t* idx_out,
                        float* logit_out, float* weight_out, void* stream) {
  FGVC_REQUIRE(pair_idx && pair_score && slot_pair && idx_out && logit_out && weight_out, FGVC_ERR_INVALID_ARG,
               "fgvc_merge_topk_f32: null pointer");
  FGVC_REQUIRE(n_out >= 0 && T >= 1 && HWq > 0 && HWk > 0, FGVC_ERR_INVALID_ARG, "fgvc_merge_topk_f32: bad shape");
  FGVC_REQUIRE(topk >= 1 && topk <= 16, FGVC_ERR_UNSUPPORTED, "fgvc_merge_topk_f32: topk=%d outside 1..16", topk);
  FGVC_REQUIRE((long long)T * HWk < (1ll << 31), FGVC_ERR_UNSUPPORTED, "fgvc_merge_topk_f32: T*HWk overflows int32");
  FGVC_REQUIRE(temperature > 0.f, FGVC_ERR_INVALID_ARG, "fgvc_merge_topk_f32: temperature must be > 0");
  FGVC_REQUIRE(weight_mode == FGVC_WEIGHT_SOFTMAX || weight_mode == FGVC_WEIGHT_COSINE, FGVC_ERR_INVALID_ARG,
               "fgvc_merge_topk_f32: unknown weight mode %d", weight_mode);
  FGVC_REQUIRE(n_out <= 65535, FGVC_ERR_UNSUPPORTED, "fgvc_merge_topk_f32: n_out=%d > 65535", n_out);
  if (n_out == 0) return FGVC_OK;
  return merge_topk_launch(pair_idx, pair_score, slot_pair, n_out, T, HWq, HWk, topk, temperature, weight_mode,
                           idx_out, logit_out, weight_out, (hipStream_t)stream);
}

int fgvc_propagate_topk_f32(const float* labels, const int32_t* slot_frame, int T, const int32_t* idx,
                            const float* weight, int Hq, int Wq, int Hk, int Wk, int P, int topk, int window_L,
                            float* out, void* stream) {
  FGVC_REQUIRE(labels && slot_frame && idx && weight && out, FGVC_ERR_INVALID_ARG, "fgvc_propagate_topk_f32: null pointer");
  FGVC_REQUIRE(T >= 1 && Hq > 0 && Wq > 0 && Hk > 0 && Wk > 0 && P > 0 && topk >= 1, FGVC_ERR_INVALID_ARG,
               "fgvc_propagate_topk_f32: bad shape");
  FGVC_REQUIRE(window_L == 0 || ((window_L & 1) && Hq == Hk && Wq == Wk), FGVC_ERR_INVALID_ARG,
               "fgvc_propagate_topk_f32: window_L must be odd and the grids equal");
  return propagate_launch(labels, slot_frame, T, idx, weight, Hq, Wq, Hk, Wk, P, topk, window_L, out,
                          (hipStream_t)stream);
}

int fgvc_corr_volume_f32(const float* qfeat, const float* kfeat, int C, int HWq, int HWk, float temperature,
                         float* vol, void* stream) {
  FGVC_REQUIRE(qfeat && kfeat && vol, FGVC_ERR_INVALID_ARG, "fgvc_corr_volume_f32: null pointer");
  FGVC_REQUIRE(aligned16(qfeat) && aligned16(kfeat), FGVC_ERR_INVALID_ARG, "fgvc_corr_volume_f32: features must be 16-byte aligned");
  FGVC_REQUIRE(HWq > 0 && HWk > 0 && temperature > 0.f, FGVC_ERR_INVALID_ARG, "fgvc_corr_volume_f32: bad shape");
  return corr_volume_f32_launch(qfeat, kfeat, C, HWq, HWk, temperature, vol, (hipStream_t)stream);
}

int fgvc_split_bf16(const float* feat, uint16_t* hi_lo, int64_t n_pixels, int C, void* stream) {
  FGVC_REQUIRE(feat && hi_lo, FGVC_ERR_INVALID_ARG, "fgvc_split_bf16: null pointer");
  FGVC_REQUIRE(n_pixels >= 0 && C > 0 && C % 4 == 0, FGVC_ERR_INVALID_ARG, "fgvc_split_bf16: C must be a multiple of 4");
  FGVC_REQUIRE(aligned16(feat) && aligned16(hi_lo), FGVC_ERR_INVALID_ARG, "fgvc_split_bf16: 16-byte alignment required");
  if (n_pixels == 0) return FGVC_OK;
  return split_bf16_launch(feat, hi_lo, n_pixels, C, (hipStream_t)stream);
}

static int corr_bf16_common(const char* who, const uint16_t* q, const uint16_t* k, int C, int HWq, int HWk,
                            float temperature, float* vol, int nseg, void* stream) {
  FGVC_REQUIRE(q && k && vol, FGVC_ERR_INVALID_ARG, "%s: null pointer", who);
  FGVC_REQUIRE(aligned16(q) && aligned16(k), FGVC_ERR_INVALID_ARG, "%s: features must be 16-byte aligned", who);
  FGVC_REQUIRE(C > 0 && C % 64 == 0, FGVC_ERR_UNSUPPORTED, "%s: C=%d must be a multiple of 64", who, C);
  FGVC_REQUIRE(HWq > 0 && HWk > 0 && temperature > 0.f, FGVC_ERR_INVALID_ARG, "%s: bad shape", who);
  return corr_volume_bf16_launch(q, k, C, HWq, HWk, temperature, vol, nseg, (hipStream_t)stream);
}

int fgvc_corr_volume_bf16x3(const uint16_t* q, const uint16_t* k, int C, int HWq, int HWk, float temperature,
                            float* vol, void* stream) {
  return corr_bf16_common("fgvc_corr_volume_bf16x3", q, k, C, HWq, HWk, temperature, vol, 3, stream);
}

int fgvc_corr_volume_bf16(const uint16_t* q, const uint16_t* k, int C, int HWq, int HWk, float temperature,
                          float* vol, void* stream) {
  return corr_bf16_common("fgvc_corr_volume_bf16", q, k, C, HWq, HWk, temperature, vol, 1, stream);
}

int fgvc_split_f16f8(const float* feat, uint8_t* out, int64_t n_pixels, int C, void* stream) {
  FGVC_REQUIRE(feat && out, FGVC_ERR_INVALID_ARG, "fgvc_split_f16f8: null pointer");
  FGVC_REQUIRE(n_pixels >= 0 && C > 0 && C % 4 == 0, FGVC_ERR_INVALID_ARG, "fgvc_split_f16f8: C must be a multiple of 4");
  FGVC_REQUIRE(aligned16(feat) && aligned16(out), FGVC_ERR_INVALID_ARG, "fgvc_split_f16f8: 16-byte alignment required");
  if (n_pixels == 0) return FGVC_OK;
  return split_f16f8_launch(feat, out, n_pixels, C, (hipStream_t)stream);
}

int fgvc_corr_volume_f16f8(const uint8_t* q, const uint8_t* k, int C, int HWq, int HWk, float temperature, float* vol,
                           void* stream) {
  FGVC_REQUIRE(q && k && vol, FGVC_ERR_INVALID_ARG, "fgvc_corr_volume_f16f8: null pointer");
  FGVC_REQUIRE(aligned16(q) && aligned16(k) && aligned16(vol), FGVC_ERR_INVALID_ARG, "fgvc_corr_volume_f16f8: 16-byte alignment required");
  FGVC_REQUIRE(C == 256, FGVC_ERR_UNSUPPORTED, "fgvc_corr_volume_f16f8: C=%d unsupported (256 only; use fgvc_corr_volume_bf16x3)", C);
  FGVC_REQUIRE(HWq > 0 && HWk > 0 && temperature > 0.f, FGVC_ERR_INVALID_ARG, "fgvc_corr_volume_f16f8: bad shape");
  FGVC_REQUIRE((long long)HWq < (1ll << 30) && (long long)HWk < (1ll << 30), FGVC_ERR_UNSUPPORTED, "fgvc_corr_volume_f16f8: grid too large");
  return corr_volume_f16f8_launch(q, k, HWq, HWk, temperature, vol, (hipStream_t)stream);
}

int fgvc_split_f16f6(const float* feat, uint8_t* out, int64_t n_pixels, int C, void* stream) {
  FGVC_REQUIRE(feat && out, FGVC_ERR_INVALID_ARG, "fgvc_split_f16f6: null pointer");
  FGVC_REQUIRE(n_pixels >= 0, FGVC_ERR_INVALID_ARG, "fgvc_split_f16f6: negative pixel count");
  FGVC_REQUIRE(C == 256, FGVC_ERR_UNSUPPORTED, "fgvc_split_f16f6: C=%d unsupported (256 only)", C);
  FGVC_REQUIRE(aligned16(feat) && aligned16(out), FGVC_ERR_INVALID_ARG, "fgvc_split_f16f6: 16-byte alignment required");
  if (n_pixels == 0) return FGVC_OK;
  return split_f16f6_launch(feat, out, n_pixels, (hipStream_t)stream);
}

int fgvc_debug_store_sweep_f32(float* buf, int64_t n_floats, int nontemporal, void* stream) {
  FGVC_REQUIRE(buf && aligned16(buf) && n_floats >= 0 && n_floats % 4 == 0, FGVC_ERR_INVALID_ARG, "fgvc_debug_store_sweep_f32: a 16-byte aligned buffer of a multiple of 4 floats");
  if (n_floats == 0) return FGVC_OK;
  return store_sweep_launch(buf, n_floats, nontemporal, (hipStream_t)stream);
}

int fgvc_corr_volume_f16f6(const uint8_t* q, const uint8_t* k, int C, int HWq, int HWk, float temperature, float* vol,
                           void* stream) {
  FGVC_REQUIRE(q && k && vol, FGVC_ERR_INVALID_ARG, "fgvc_corr_volume_f16f6: null pointer");
  FGVC_REQUIRE(aligned16(q) && aligned16(k) && aligned16(vol), FGVC_ERR_INVALID_ARG, "fgvc_corr_volume_f16f6: 16-byte alignment required");
  FGVC_REQUIRE(C == 256, FGVC_ERR_UNSUPPORTED, "fgvc_corr_volume_f16f6: C=%d unsupported (256 only; use fgvc_corr_volume_bf16x3)", C);
  FGVC_REQUIRE(HWq > 0 && HWk > 0 && temperature > 0.f, FGVC_ERR_INVALID_ARG, "fgvc_corr_volume_f16f6: bad shape");
  FGVC_REQUIRE((long long)HWq < (1ll << 30) && (long long)HWk < (1ll << 30), FGVC_ERR_UNSUPPORTED, "fgvc_corr_volume_f16f6: grid too large");
  return corr_volume_f16f6_launch(q, k, HWq, HWk, temperature, vol, (hipStream_t)stream);
}

int fgvc_dense_attend_splits(int HWq, int HWk) { return (HWq > 0 && HWk > 0) ? dense_attend_splits(HWq, HWk) : 1; }

int fgvc_dense_attend_f32(const float* vol, const float* labels, int Hq, int Wq, int Hk, int Wk, int P, int masked, int r2max,
                          int ry, int rx, int weight_mode, int first, float* state, int nsplit, void* stream) {
  FGVC_REQUIRE(vol && labels && state, FGVC_ERR_INVALID_ARG, "fgvc_dense_attend_f32: null pointer");
  FGVC_REQUIRE(Hq > 0 && Wq > 0 && Hk > 0 && Wk > 0, FGVC_ERR_INVALID_ARG, "fgvc_dense_attend_f32: bad shape");
  FGVC_REQUIRE((long long)Hk * Wk < (1ll << 30) && (long long)Hq * Wq < (1ll << 30), FGVC_ERR_UNSUPPORTED,
               "fgvc_dense_attend_f32: grid too large");
  FGVC_REQUIRE(P >= 1 && P <= 32, FGVC_ERR_UNSUPPORTED, "fgvc_dense_attend_f32: P=%d outside 1..32", P);
  FGVC_REQUIRE(nsplit >= 1 && nsplit <= 65535, FGVC_ERR_INVALID_ARG, "fgvc_dense_attend_f32: nsplit=%d", nsplit);
  FGVC_REQUIRE(weight_mode == FGVC_WEIGHT_SOFTMAX || weight_mode == FGVC_WEIGHT_COSINE || weight_mode == FGVC_WEIGHT_RAW, FGVC_ERR_INVALID_ARG,
               "fgvc_dense_attend_f32: unknown weight mode %d", weight_mode);
  FGVC_REQUIRE(!masked || (r2max >= 0 && ry >= 0 && rx >= 0), FGVC_ERR_INVALID_ARG, "fgvc_dense_attend_f32: negative mask parameter");
  FGVC_REQUIRE(!masked || (Hq == Hk && Wq == Wk), FGVC_ERR_INVALID_ARG,
               "fgvc_dense_attend_f32: a spatial mask needs equal query/key grids (local_attention.py:331)");
  return dense_attend_launch(vol, labels, Hq, Wq, Hk, Wk, P, masked != 0, r2max, ry, rx, weight_mode,
                             first != 0, state, nsplit, (hipStream_t)stream, nullptr);
}

int fgvc_dense_kth_f32(const float* aff, int HWk, int HWq, int k, float* part, int nsplit, float* thr, void* stream) {
  FGVC_REQUIRE(aff && part && thr, FGVC_ERR_INVALID_ARG, "fgvc_dense_kth_f32: null pointer");
  FGVC_REQUIRE(HWk > 0 && HWq > 0 && (long long)HWk < (1ll << 30) && (long long)HWq < (1ll << 30), FGVC_ERR_INVALID_ARG, "fgvc_dense_kth_f32: bad shape");
  FGVC_REQUIRE(k >= 1 && k <= 64 && k <= HWk, FGVC_ERR_UNSUPPORTED, "fgvc_dense_kth_f32: k=%d outside 1..min(64, HWk)", k);
  FGVC_REQUIRE(nsplit >= 1 && nsplit <= 65535, FGVC_ERR_INVALID_ARG, "fgvc_dense_kth_f32: nsplit=%d", nsplit);
  return dense_kth_launch(aff, HWk, HWq, k, part, nsplit, thr, (hipStream_t)stream);
}

int fgvc_dense_propagate_f32(const float* aff, const float* labels, int HWk, int HWq, int P, const float* thr, float* state,
                             int nsplit, float* out, void* stream) {
  FGVC_REQUIRE(aff && labels && state && out, FGVC_ERR_INVALID_ARG, "fgvc_dense_propagate_f32: null pointer");
  FGVC_REQUIRE(HWk > 0 && HWq > 0 && (long long)HWk < (1ll << 30) && (long long)HWq < (1ll << 30), FGVC_ERR_INVALID_ARG, "fgvc_dense_propagate_f32: bad shape");
  FGVC_REQUIRE(P >= 1 && P <= 32, FGVC_ERR_UNSUPPORTED, "fgvc_dense_propagate_f32: P=%d outside 1..32 (call per 32 channels)", P);
  FGVC_REQUIRE(nsplit >= 1 && nsplit <= 65535, FGVC_ERR_INVALID_ARG, "fgvc_dense_propagate_f32: nsplit=%d", nsplit);
  const int mode = thr ? 3 : 2;
  int rc = dense_attend_launch(aff, labels, 1, HWq, 1, HWk, P, 0, FGVC_NO_LIMIT, FGVC_NO_LIMIT, FGVC_NO_LIMIT, mode, 1, state, nsplit,
                               (hipStream_t)stream, thr);
  if (rc != FGVC_OK) return rc;
  return dense_attend_finish_launch(state, nsplit, HWq, P, mode, out, (hipStream_t)stream);
}

int fgvc_dense_attend_finish_f32(const float* state, int nsplit, int HWq, int P, int weight_mode, float* out, void* stream) {
  FGVC_REQUIRE(state && out, FGVC_ERR_INVALID_ARG, "fgvc_dense_attend_finish_f32: null pointer");
  FGVC_REQUIRE(HWq > 0 && P >= 1 && P <= 32 && nsplit >= 1, FGVC_ERR_INVALID_ARG, "fgvc_dense_attend_finish_f32: bad shape");
  FGVC_REQUIRE(weight_mode >= 0 && weight_mode <= 2, FGVC_ERR_INVALID_ARG, "fgvc_dense_attend_finish_f32: unknown weight mode %d", weight_mode);
  return dense_attend_finish_launch(state, nsplit, HWq, P, weight_mode, out, (hipStream_t)stream);
}

int fgvc_local_corr_topk_f32(const float* qfeat, const float* kfeat, const int32_t* pairs, int n_slots, int C, int H,
                             int W, int R, int topk, float temperature, int32_t* pair_idx_ws, float* pair_score_ws,
                             int32_t* idx_out, float* logit_out, float* weight_out, void* stream) {
  FGVC_REQUIRE(pair_idx_ws && pair_score_ws && idx_out && logit_out && weight_out, FGVC_ERR_INVALID_ARG,
               "fgvc_local_corr_topk_f32: null pointer");
  FGVC_REQUIRE(R >= 0 && n_slots >= 1 && temperature > 0.f, FGVC_ERR_INVALID_ARG, "fgvc_local_corr_topk_f32: bad R/n_slots/temperature");
  FGVC_REQUIRE((long long)n_slots * (2 * R + 1) * (2 * R + 1) < (1ll << 31), FGVC_ERR_UNSUPPORTED,
               "fgvc_local_corr_topk_f32: index overflow");
  int rc = fgvc_pair_topk_f32(qfeat, kfeat, pairs, n_slots, C, H, W, H, W, FGVC_NO_LIMIT, R, R, topk, nullptr,
                              pair_idx_ws, pair_score_ws, stream);
  if (rc != FGVC_OK) return rc;
  return local_merge_launch(pair_idx_ws, pair_score_ws, n_slots, H, W, R, topk, temperature, idx_out, logit_out,
                            weight_out, (hipStream_t)stream);
}

int fgvc_local_corr_topk_f16x3(const uint16_t* qsplit, const uint16_t* ksplit, const int32_t* pairs, int n_slots, int C,
                               int H, int W, int R, int topk, float temperature, int32_t* pair_idx_ws,
                               float* pair_score_ws, int32_t* idx_out, float* logit_out, float* weight_out,
                               void* stream) {
  FGVC_REQUIRE(pair_idx_ws && pair_score_ws && idx_out && logit_out && weight_out, FGVC_ERR_INVALID_ARG,
               "fgvc_local_corr_topk_f16x3: null pointer");
  FGVC_REQUIRE(R >= 0 && n_slots >= 1 && temperature > 0.f, FGVC_ERR_INVALID_ARG, "fgvc_local_corr_topk_f16x3: bad R/n_slots/temperature");
  FGVC_REQUIRE((long long)n_slots * (2 * R + 1) * (2 * R + 1) < (1ll << 31), FGVC_ERR_UNSUPPORTED,
               "fgvc_local_corr_topk_f16x3: index overflow");
  // every slot of a local window is a masked pair (the window IS the mask): all_masked = 1
  int rc = fgvc_pair_topk_f16x3(qsplit, ksplit, pairs, n_slots, C, H, W, H, W, FGVC_NO_LIMIT, R, R, topk, 1, pair_idx_ws,
                                pair_score_ws, stream);
  if (rc != FGVC_OK) return rc;
  return local_merge_launch(pair_idx_ws, pair_score_ws, n_slots, H, W, R, topk, temperature, idx_out, logit_out,
                            weight_out, (hipStream_t)stream);
}

int fgvc_topk_coord_f32(const int32_t* idx, const float* weight, int H, int W, int R, int topk, int scale, float* out,
                        void* stream) {
  FGVC_REQUIRE(idx && weight && out, FGVC_ERR_INVALID_ARG, "fgvc_topk_coord_f32: null pointer");
  FGVC_REQUIRE(H > 0 && W > 0 && R >= 0 && topk >= 1 && scale >= 1, FGVC_ERR_INVALID_ARG, "fgvc_topk_coord_f32: bad shape");
  return topk_coord_launch(idx, weight, H, W, R, topk, scale, out, (hipStream_t)stream);
}

int fgvc_c2f_refine_f32(const int32_t* coarse_arg, const float* qfine, const float* kfine, const float* vfine, int T,
                        int H, int W, int scale, int Cf, int P, int Rf, int topk, float temperature, float* out,
                        int32_t* idx_out, float* logit_out, void* stream) {
  return fgvc_c2f_refine_mode_f32(coarse_arg, qfine, kfine, vfine, T, H, W, scale, Cf, P, Rf, topk, temperature, FGVC_WEIGHT_SOFTMAX,
                                  out, idx_out, logit_out, stream);
}

int fgvc_c2f_refine_mode_f32(const int32_t* coarse_arg, const float* qfine, const float* kfine, const float* vfine, int T,
                             int H, int W, int scale, int Cf, int P, int Rf, int topk, float temperature, int weight_mode,
                             float* out, int32_t* idx_out, float* logit_out, void* stream) {
  FGVC_REQUIRE(coarse_arg && qfine && kfine && vfine && out && idx_out && logit_out, FGVC_ERR_INVALID_ARG,
               "fgvc_c2f_refine_f32: null pointer");
  FGVC_REQUIRE(T >= 1 && H > 0 && W > 0 && scale >= 1 && P > 0 && Rf >= 0, FGVC_ERR_INVALID_ARG, "fgvc_c2f_refine_f32: bad shape");
  FGVC_REQUIRE(Cf > 0 && Cf % 4 == 0, FGVC_ERR_UNSUPPORTED, "fgvc_c2f_refine_f32: Cf=%d must be a multiple of 4", Cf);
  FGVC_REQUIRE(aligned16(qfine) && aligned16(kfine), FGVC_ERR_INVALID_ARG, "fgvc_c2f_refine_f32: 16-byte alignment required");
  FGVC_REQUIRE(topk >= 1 && topk <= 16, FGVC_ERR_UNSUPPORTED, "fgvc_c2f_refine_f32: topk=%d outside 1..16", topk);
  FGVC_REQUIRE(temperature > 0.f, FGVC_ERR_INVALID_ARG, "fgvc_c2f_refine_f32: temperature must be > 0");
  FGVC_REQUIRE(weight_mode == FGVC_WEIGHT_SOFTMAX || weight_mode == FGVC_WEIGHT_COSINE, FGVC_ERR_INVALID_ARG,
               "fgvc_c2f_refine_mode_f32: weight_mode %d (FGVC_WEIGHT_SOFTMAX or FGVC_WEIGHT_COSINE)", weight_mode);
  return c2f_refine_launch(coarse_arg, qfine, kfine, vfine, T, H, W, scale, Cf, P, Rf, topk, temperature, weight_mode, out,
                           idx_out, logit_out, (hipStream_t)stream);
}

int fgvc_bn_act_f32(const float* x, const float* residual, const float* mean, const float* var, const float* gamma,
                    const float* beta, float eps, int relu, float* out, int N, int C, int HW, void* stream) {
  FGVC_REQUIRE(x && mean && var && gamma && beta && out, FGVC_ERR_INVALID_ARG, "fgvc_bn_act_f32: null pointer");
  FGVC_REQUIRE(N >= 0 && C > 0 && HW > 0 && eps >= 0.f, FGVC_ERR_INVALID_ARG, "fgvc_bn_act_f32: bad shape");
  if (N == 0) return FGVC_OK;
  return bn_act_launch(x, residual, mean, var, gamma, beta, eps, relu, out, N, C, HW, (hipStream_t)stream);
}

static bool conv_pad_ok(int H, int W, int Hp, int Wp) {
  return Hp >= 8 * cdiv(H, 8) + 2 && Wp >= 32 * cdiv(W, 32) + 8;
}

int fgvc_nchw_to_split_nhwc_f32(const float* in, uint16_t* out, float* out_f32, int N, int C, int H, int W, int Hp, int Wp,
                                void* stream) {
  FGVC_REQUIRE(in && (out || out_f32), FGVC_ERR_INVALID_ARG, "fgvc_nchw_to_split_nhwc_f32: null pointer");
  FGVC_REQUIRE(N >= 0 && C > 0 && C % 32 == 0 && H > 0 && W > 0, FGVC_ERR_INVALID_ARG,
               "fgvc_nchw_to_split_nhwc_f32: bad shape (C must be a multiple of 32)");
  FGVC_REQUIRE(conv_pad_ok(H, W, Hp, Wp), FGVC_ERR_INVALID_ARG,
               "fgvc_nchw_to_split_nhwc_f32: padded size %dx%d too small for %dx%d (need >= 8*ceil(H/8)+2 x 32*ceil(W/32)+8)", Hp, Wp, H, W);
  FGVC_REQUIRE(aligned16(out) && aligned16(out_f32) && (long long)N * (C / 32) <= 65535 && H <= 65535, FGVC_ERR_INVALID_ARG,
               "fgvc_nchw_to_split_nhwc_f32: alignment / grid limits");
  if (N == 0) return FGVC_OK;
  return nchw_to_split_nhwc_launch(in, out, out_f32, N, C, H, W, Hp, Wp, (hipStream_t)stream);
}

int fgvc_conv_split_f32(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual, uint16_t* y_split,
                        float* y_f32, int N, int H, int W, int Hp, int Wp, int Cin, int Cout, int KS, int relu,
                        void* stream) {
  return fgvc_conv_split_fmt_f32(x, w, bias, residual, y_split, y_f32, N, H, W, Hp, Wp, Cin, Cout, KS, relu, FGVC_ACT_BF16X2, 0,
                                 FGVC_ACT_BF16X2, 0, nullptr, stream);
}

int fgvc_conv_split_fmt_f32(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual, uint16_t* y_split,
                            float* y_f32, int N, int H, int W, int Hp, int Wp, int Cin, int Cout, int KS, int relu, int in_fmt,
                            int in_scale_log2, int out_fmt, int out_scale_log2, int* overflow, void* stream) {
  FGVC_REQUIRE(x && w && bias && (y_split || y_f32), FGVC_ERR_INVALID_ARG, "fgvc_conv_split_f32: null pointer");
  FGVC_REQUIRE(in_fmt >= 0 && in_fmt <= 3 && out_fmt >= 0 && out_fmt <= 3, FGVC_ERR_INVALID_ARG, "fgvc_conv_split_fmt_f32: unknown format %d / %d", in_fmt, out_fmt);
  FGVC_REQUIRE(out_fmt == FGVC_ACT_BF16X2 || !y_split || overflow, FGVC_ERR_INVALID_ARG, "fgvc_conv_split_fmt_f32: an f16-format output needs the overflow word");
  FGVC_REQUIRE(in_scale_log2 > -100 && in_scale_log2 < 100 && out_scale_log2 > -100 && out_scale_log2 < 100, FGVC_ERR_INVALID_ARG,
               "fgvc_conv_split_fmt_f32: scale exponent out of range");
  FGVC_REQUIRE(N >= 0 && H > 0 && W > 0, FGVC_ERR_INVALID_ARG, "fgvc_conv_split_f32: bad shape");
  FGVC_REQUIRE(KS == 1 || KS == 3, FGVC_ERR_UNSUPPORTED, "fgvc_conv_split_f32: kernel size %d (1 or 3, stride 1)", KS);
  FGVC_REQUIRE(Cin > 0 && Cin % 32 == 0 && Cout > 0 && Cout % 64 == 0, FGVC_ERR_UNSUPPORTED,
               "fgvc_conv_split_f32: Cin=%d must be a multiple of 32 and Cout=%d of 64", Cin, Cout);
  FGVC_REQUIRE(conv_pad_ok(H, W, Hp, Wp), FGVC_ERR_INVALID_ARG, "fgvc_conv_split_f32: padded size %dx%d too small for %dx%d", Hp, Wp, H, W);
  FGVC_REQUIRE(aligned16(x) && aligned16(w) && aligned16(bias) && aligned16(residual) && aligned16(y_split) && aligned16(y_f32),
               FGVC_ERR_INVALID_ARG, "fgvc_conv_split_f32: 16-byte alignment required");
  FGVC_REQUIRE((const void*)x != (const void*)y_split, FGVC_ERR_INVALID_ARG, "fgvc_conv_split_f32: in-place not supported");
  if (N == 0) return FGVC_OK;
  return conv_split_launch(x, w, bias, residual, y_split, y_f32, N, H, W, Hp, Wp, Cin, Cout, KS, relu, in_fmt, in_scale_log2, out_fmt,
                           out_scale_log2, overflow, (hipStream_t)stream);
}

int fgvc_conv_split_proj_fmt_f32(const uint16_t* x, const uint16_t* w, const uint16_t* x2, const uint16_t* w2, const float* bias,
                                 const float* residual, uint16_t* y_split, float* y_f32, int N, int H, int W, int Hp, int Wp, int Cin, int Cin2,
                                 int relu, int in_fmt, int in_scale_log2, int out_fmt, int out_scale_log2, int* overflow, void* stream) {
  FGVC_REQUIRE(x && w && x2 && w2 && bias && (y_split || y_f32), FGVC_ERR_INVALID_ARG, "fgvc_conv_split_proj_fmt_f32: null pointer");
  FGVC_REQUIRE(in_fmt >= 0 && in_fmt <= 3 && out_fmt >= 0 && out_fmt <= 3, FGVC_ERR_INVALID_ARG, "fgvc_conv_split_proj_fmt_f32: unknown format %d / %d", in_fmt, out_fmt);
  FGVC_REQUIRE(out_fmt == FGVC_ACT_BF16X2 || !y_split || overflow, FGVC_ERR_INVALID_ARG, "fgvc_conv_split_proj_fmt_f32: an f16-format output needs the overflow word");
  FGVC_REQUIRE(in_scale_log2 > -100 && in_scale_log2 < 100 && out_scale_log2 > -100 && out_scale_log2 < 100, FGVC_ERR_INVALID_ARG,
               "fgvc_conv_split_proj_fmt_f32: scale exponent out of range");
  FGVC_REQUIRE(N >= 0 && H > 0 && W > 0, FGVC_ERR_INVALID_ARG, "fgvc_conv_split_proj_fmt_f32: bad shape");
  FGVC_REQUIRE(Cin > 0 && Cin % 32 == 0 && Cin2 > 0 && Cin2 % 32 == 0, FGVC_ERR_UNSUPPORTED,
               "fgvc_conv_split_proj_fmt_f32: Cin=%d and Cin2=%d must be multiples of 32 (Cout is 256)", Cin, Cin2);
  FGVC_REQUIRE(conv_pad_ok(H, W, Hp, Wp), FGVC_ERR_INVALID_ARG, "fgvc_conv_split_proj_fmt_f32: padded size %dx%d too small for %dx%d", Hp, Wp, H, W);
  FGVC_REQUIRE(aligned16(x) && aligned16(w) && aligned16(x2) && aligned16(w2) && aligned16(bias) && aligned16(residual) && aligned16(y_split) &&
               aligned16(y_f32), FGVC_ERR_INVALID_ARG, "fgvc_conv_split_proj_fmt_f32: 16-byte alignment required");
  FGVC_REQUIRE((const void*)x != (const void*)y_split && (const void*)x2 != (const void*)y_split, FGVC_ERR_INVALID_ARG,
               "fgvc_conv_split_proj_fmt_f32: in-place not supported");
  if (N == 0) return FGVC_OK;
  return conv_split_launch(x, w, bias, residual, y_split, y_f32, N, H, W, Hp, Wp, Cin, 256, 3, relu, in_fmt, in_scale_log2, out_fmt,
                           out_scale_log2, overflow, (hipStream_t)stream, nullptr, 1, x2, w2, Cin2);
}

static int conv_split_bank_common(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual, void* bank, int N,
                                  int H, int W, int Hp, int Wp, int Cin, int KS, int relu, int in_fmt, int in_scale_log2,
                                  int normalize, void* stream, int row_bytes) {
  FGVC_REQUIRE(x && w && bias && bank, FGVC_ERR_INVALID_ARG, "fgvc_conv_split_bank_f16f6p_f32: null pointer");
  FGVC_REQUIRE(in_fmt >= 0 && in_fmt <= 3, FGVC_ERR_INVALID_ARG, "fgvc_conv_split_bank_f16f6p_f32: unknown format %d", in_fmt);
  FGVC_REQUIRE(in_scale_log2 > -100 && in_scale_log2 < 100, FGVC_ERR_INVALID_ARG, "fgvc_conv_split_bank_f16f6p_f32: scale exponent out of range");
  FGVC_REQUIRE(N >= 0 && H > 0 && W > 0, FGVC_ERR_INVALID_ARG, "fgvc_conv_split_bank_f16f6p_f32: bad shape");
  FGVC_REQUIRE(KS == 3, FGVC_ERR_UNSUPPORTED, "fgvc_conv_split_bank_f16f6p_f32: kernel size %d (3 x 3, stride 1: the last convolution of a BasicBlock)", KS);
  FGVC_REQUIRE(Cin > 0 && Cin % 32 == 0, FGVC_ERR_UNSUPPORTED, "fgvc_conv_split_bank_f16f6p_f32: Cin=%d must be a multiple of 32 (Cout is 256)", Cin);
  FGVC_REQUIRE(conv_pad_ok(H, W, Hp, Wp), FGVC_ERR_INVALID_ARG, "fgvc_conv_split_bank_f16f6p_f32: padded size %dx%d too small for %dx%d", Hp, Wp, H, W);
  FGVC_REQUIRE(aligned16(x) && aligned16(w) && aligned16(bias) && aligned16(residual) && aligned16(bank), FGVC_ERR_INVALID_ARG,
               "fgvc_conv_split_bank_f16f6p_f32: 16-byte alignment required");
  if (N == 0) return FGVC_OK;
  return conv_split_launch(x, w, bias, residual, nullptr, nullptr, N, H, W, Hp, Wp, Cin, 256, KS, relu, in_fmt, in_scale_log2, 0, 0, nullptr,
                           (hipStream_t)stream, reinterpret_cast<unsigned char*>(bank), normalize, nullptr, nullptr, 0, row_bytes);
}

int fgvc_conv_split_bank_f16f6p_f32(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual, void* bank, int N,
                                    int H, int W, int Hp, int Wp, int Cin, int KS, int relu, int in_fmt, int in_scale_log2,
                                    int normalize, void* stream) {
  return conv_split_bank_common(x, w, bias, residual, bank, N, H, W, Hp, Wp, Cin, KS, relu, in_fmt, in_scale_log2, normalize, stream, 1024);
}

int fgvc_conv_split_bank_f16f6x_f32(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual, void* bank, int N,
                                    int H, int W, int Hp, int Wp, int Cin, int KS, int relu, int in_fmt, int in_scale_log2,
                                    int normalize, void* stream) {
  return conv_split_bank_common(x, w, bias, residual, bank, N, H, W, Hp, Wp, Cin, KS, relu, in_fmt, in_scale_log2, normalize, stream, 2048);
}

/* debug: the 32 s_memtime sums workgroup 77 leaves with fgvc_set_option("conv64_variant", 8) (tools/experiments/time_conv64_variants.py) */
int fgvc_conv64_probe(int64_t* out32) {
  FGVC_REQUIRE(out32 != nullptr, FGVC_ERR_INVALID_ARG, "fgvc_conv64_probe: null pointer");
  return conv64_probe_read(reinterpret_cast<long long*>(out32)) == 0 ? FGVC_OK : FGVC_ERR_LAUNCH;
}

int fgvc_conv64_split_f32(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual, uint16_t* y_split,
                          float* y_f32, int N, int H, int W, int Hp, int Wp, int relu, void* stream) {
  return fgvc_conv64_split_res_f32(x, w, bias, residual, nullptr, y_split, y_f32, N, H, W, Hp, Wp, relu, stream);
}

int fgvc_conv64_split_res_f32(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual,
                              const uint16_t* residual_split, uint16_t* y_split, float* y_f32, int N, int H, int W, int Hp, int Wp,
                              int relu, void* stream) {
  return fgvc_conv64_split_fmt_f32(x, w, bias, residual, residual_split, y_split, y_f32, N, H, W, Hp, Wp, relu, FGVC_ACT_BF16X2, 0,
                                   FGVC_ACT_BF16X2, 0, nullptr, stream);
}

int fgvc_conv64_split_fmt_f32(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual,
                              const uint16_t* residual_split, uint16_t* y_split, float* y_f32, int N, int H, int W, int Hp, int Wp,
                              int relu, int in_fmt, int in_scale_log2, int out_fmt, int out_scale_log2, int* overflow, void* stream) {
  FGVC_REQUIRE(x && w && bias && (y_split || y_f32), FGVC_ERR_INVALID_ARG, "fgvc_conv64_split_f32: null pointer");
  FGVC_REQUIRE((in_fmt == FGVC_ACT_BF16X2 || in_fmt == FGVC_ACT_F16F8) && (out_fmt == FGVC_ACT_BF16X2 || out_fmt == FGVC_ACT_F16F8),
               FGVC_ERR_UNSUPPORTED, "fgvc_conv64_split_fmt_f32: formats %d -> %d (bf16x2 or f16f8)", in_fmt, out_fmt);
  FGVC_REQUIRE(out_fmt == FGVC_ACT_BF16X2 || !y_split || overflow, FGVC_ERR_INVALID_ARG,
               "fgvc_conv64_split_fmt_f32: an f16-format output needs the overflow word");
  FGVC_REQUIRE(in_scale_log2 > -100 && in_scale_log2 < 100 && out_scale_log2 > -100 && out_scale_log2 < 100, FGVC_ERR_INVALID_ARG,
               "fgvc_conv64_split_fmt_f32: scale exponent out of range");
  FGVC_REQUIRE(!(residual && residual_split), FGVC_ERR_INVALID_ARG, "fgvc_conv64_split_res_f32: one residual, f32 or split, not both");
  FGVC_REQUIRE(!residual_split || in_fmt == FGVC_ACT_BF16X2, FGVC_ERR_UNSUPPORTED,
               "fgvc_conv64_split_fmt_f32: a split residual is read as (hi, lo) bf16: bf16x2 tensors only");
  FGVC_REQUIRE(N >= 0 && H > 0 && W > 0, FGVC_ERR_INVALID_ARG, "fgvc_conv64_split_f32: bad shape");
  FGVC_REQUIRE(conv_pad_ok(H, W, Hp, Wp), FGVC_ERR_INVALID_ARG, "fgvc_conv64_split_f32: padded size %dx%d too small for %dx%d", Hp, Wp, H, W);
  FGVC_REQUIRE(aligned16(x) && aligned16(w) && aligned16(bias) && aligned16(residual) && aligned16(residual_split) && aligned16(y_split) &&
                   aligned16(y_f32),
               FGVC_ERR_INVALID_ARG, "fgvc_conv64_split_f32: 16-byte alignment required");
  FGVC_REQUIRE((const void*)x != (const void*)y_split && (!residual_split || (const void*)residual_split != (const void*)y_split),
               FGVC_ERR_INVALID_ARG, "fgvc_conv64_split_f32: in-place not supported");
  if (N == 0) return FGVC_OK;
  return conv64_launch(x, w, bias, residual, residual_split, y_split, y_f32, N, H, W, Hp, Wp, relu, in_fmt, in_scale_log2, out_fmt,
                       out_scale_log2, overflow, (hipStream_t)stream);
}

int fgvc_conv_s2_split_f32(const uint16_t* x, const uint16_t* w, const float* bias, uint16_t* y_split, float* y_f32, int N,
                           int H, int W, int Hp, int Wp, int Cin, int Cout, int KS, int Hop, int Wop, int relu, void* stream) {
  return fgvc_conv_s2_split_fmt_f32(x, w, bias, y_split, y_f32, N, H, W, Hp, Wp, Cin, Cout, KS, Hop, Wop, relu, FGVC_ACT_BF16X2, 0, nullptr, stream);
}

int fgvc_conv_s2_split_fmt_f32(const uint16_t* x, const uint16_t* w, const float* bias, uint16_t* y_split, float* y_f32, int N,
                               int H, int W, int Hp, int Wp, int Cin, int Cout, int KS, int Hop, int Wop, int relu, int out_fmt,
                               int out_scale_log2, int* overflow, void* stream) {
  FGVC_REQUIRE(x && w && bias && (y_split || y_f32), FGVC_ERR_INVALID_ARG, "fgvc_conv_s2_split_f32: null pointer");
  FGVC_REQUIRE(out_fmt >= 0 && out_fmt <= 3 && out_scale_log2 > -100 && out_scale_log2 < 100, FGVC_ERR_INVALID_ARG, "fgvc_conv_s2_split_fmt_f32: bad output format / scale");
  FGVC_REQUIRE(out_fmt == FGVC_ACT_BF16X2 || !y_split || overflow, FGVC_ERR_INVALID_ARG, "fgvc_conv_s2_split_fmt_f32: an f16-format output needs the overflow word");
  FGVC_REQUIRE(N >= 0 && H > 0 && W > 0, FGVC_ERR_INVALID_ARG, "fgvc_conv_s2_split_f32: bad shape");
  FGVC_REQUIRE(KS == 1 || KS == 3, FGVC_ERR_UNSUPPORTED, "fgvc_conv_s2_split_f32: kernel size %d (1 or 3, stride 2)", KS);
  FGVC_REQUIRE(Cin > 0 && Cin % 32 == 0 && Cout > 0 && Cout % 32 == 0, FGVC_ERR_UNSUPPORTED,
               "fgvc_conv_s2_split_f32: Cin=%d and Cout=%d must be multiples of 32", Cin, Cout);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;     // both forms: floor((H + 2 pad - KS) / 2) + 1 with pad = KS / 2
  FGVC_REQUIRE(conv_pad_ok(H, W, Hp, Wp), FGVC_ERR_INVALID_ARG, "fgvc_conv_s2_split_f32: padded size %dx%d too small for %dx%d", Hp, Wp, H, W);
  FGVC_REQUIRE(!y_split || conv_pad_ok(Ho, Wo, Hop, Wop), FGVC_ERR_INVALID_ARG,
               "fgvc_conv_s2_split_f32: padded output size %dx%d too small for %dx%d", Hop, Wop, Ho, Wo);
  FGVC_REQUIRE(aligned16(x) && aligned16(w) && aligned16(bias) && aligned16(y_split) && aligned16(y_f32),
               FGVC_ERR_INVALID_ARG, "fgvc_conv_s2_split_f32: 16-byte alignment required");
  FGVC_REQUIRE((const void*)x != (const void*)y_split, FGVC_ERR_INVALID_ARG, "fgvc_conv_s2_split_f32: in-place not supported");
  if (N == 0) return FGVC_OK;
  return conv_s2_launch(x, w, bias, y_split, y_f32, N, Hp, Wp, Cin, Cout, KS, Ho, Wo, Hop, Wop, relu, out_fmt, out_scale_log2, overflow,
                        (hipStream_t)stream);
}

int fgvc_stem7_split_f32(const float* x, const uint16_t* w, const float* bias, uint16_t* y_split, float* y_f32, int N, int H,
                         int W, int Hop, int Wop, int relu, void* stream) {
  return fgvc_stem7_split_fmt_f32(x, w, bias, y_split, y_f32, N, H, W, Hop, Wop, relu, FGVC_ACT_BF16X2, 0, nullptr, stream);
}

int fgvc_stem7_split_fmt_f32(const float* x, const uint16_t* w, const float* bias, uint16_t* y_split, float* y_f32, int N, int H,
                             int W, int Hop, int Wop, int relu, int out_fmt, int out_scale_log2, int* overflow, void* stream) {
  FGVC_REQUIRE(x && w && bias && (y_split || y_f32), FGVC_ERR_INVALID_ARG, "fgvc_stem7_split_f32: null pointer");
  FGVC_REQUIRE(out_fmt == FGVC_ACT_BF16X2 || out_fmt == FGVC_ACT_F16F8, FGVC_ERR_UNSUPPORTED, "fgvc_stem7_split_fmt_f32: output format %d", out_fmt);
  FGVC_REQUIRE(out_fmt == FGVC_ACT_BF16X2 || !y_split || overflow, FGVC_ERR_INVALID_ARG,
               "fgvc_stem7_split_fmt_f32: an f16-format output needs the overflow word");
  FGVC_REQUIRE(out_scale_log2 > -100 && out_scale_log2 < 100, FGVC_ERR_INVALID_ARG, "fgvc_stem7_split_fmt_f32: scale exponent out of range");
  FGVC_REQUIRE(N >= 0 && H > 0 && W > 0, FGVC_ERR_INVALID_ARG, "fgvc_stem7_split_f32: bad shape");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;     // floor((H + 6 - 7) / 2) + 1
  FGVC_REQUIRE(!y_split || conv_pad_ok(Ho, Wo, Hop, Wop), FGVC_ERR_INVALID_ARG,
               "fgvc_stem7_split_f32: padded output size %dx%d too small for %dx%d", Hop, Wop, Ho, Wo);
  FGVC_REQUIRE(aligned16(w) && aligned16(bias) && aligned16(y_split) && aligned16(y_f32) && ((uintptr_t)x & 3) == 0,
               FGVC_ERR_INVALID_ARG, "fgvc_stem7_split_f32: alignment (16 bytes; 4 for x)");
  if (N == 0) return FGVC_OK;
  return stem7_launch(x, w, bias, y_split, y_f32, N, H, W, Ho, Wo, Hop, Wop, relu, out_fmt, out_scale_log2, overflow, (hipStream_t)stream);
}

int fgvc_nhwc_to_split_f32(float* x, uint16_t* out, int N, int C, int H, int W, int Hp, int Wp, int relu, void* stream) {
  FGVC_REQUIRE(x && out, FGVC_ERR_INVALID_ARG, "fgvc_nhwc_to_split_f32: null pointer");
  FGVC_REQUIRE(N >= 0 && C > 0 && C % 32 == 0 && H > 0 && W > 0, FGVC_ERR_INVALID_ARG,
               "fgvc_nhwc_to_split_f32: bad shape (C must be a multiple of 32)");
  FGVC_REQUIRE(conv_pad_ok(H, W, Hp, Wp), FGVC_ERR_INVALID_ARG, "fgvc_nhwc_to_split_f32: padded size %dx%d too small for %dx%d", Hp, Wp, H, W);
  FGVC_REQUIRE(aligned16(x) && aligned16(out), FGVC_ERR_INVALID_ARG, "fgvc_nhwc_to_split_f32: 16-byte alignment required");
  FGVC_REQUIRE((long long)N * H * W * C / 4 < (1ll << 31) * 256, FGVC_ERR_UNSUPPORTED, "fgvc_nhwc_to_split_f32: tensor too large");
  if (N == 0) return FGVC_OK;
  return nhwc_to_split_launch(x, out, N, C, H, W, Hp, Wp, relu, (hipStream_t)stream);
}

static int normalize_split_common(const char* what, const float* in, float* out_f32, uint16_t* out_split, int N, int C, int H, int W,
                                  int normalize, int fmt, void* stream) {
  FGVC_REQUIRE(in && (out_f32 || out_split), FGVC_ERR_INVALID_ARG, "%s: null pointer", what);
  FGVC_REQUIRE(N >= 0 && C > 0 && C % 4 == 0 && H > 0 && W > 0, FGVC_ERR_INVALID_ARG, "%s: bad shape", what);
  FGVC_REQUIRE(aligned16(in) && aligned16(out_f32) && aligned16(out_split), FGVC_ERR_INVALID_ARG, "%s: 16-byte alignment required", what);
  if (N == 0) return FGVC_OK;
  return normalize_nhwc_launch(in, out_f32, out_split, N, C, H, W, normalize, fmt, (hipStream_t)stream);
}

int fgvc_normalize_split_nhwc_f32(const float* in, float* out_f32, uint16_t* out_split, int N, int C, int H, int W, int normalize,
                                  void* stream) {
  return normalize_split_common("fgvc_normalize_split_nhwc_f32", in, out_f32, out_split, N, C, H, W, normalize, 0, stream);
}

int fgvc_normalize_split_f16x2_nhwc_f32(const float* in, float* out_f32, uint16_t* out_split, int N, int C, int H, int W,
                                        int normalize, void* stream) {
  return normalize_split_common("fgvc_normalize_split_f16x2_nhwc_f32", in, out_f32, out_split, N, C, H, W, normalize, 1, stream);
}

int fgvc_normalize_split_f16f6p_nhwc_f32(const float* in, uint8_t* rows, int N, int C, int H, int W, int normalize, void* stream) {
  FGVC_REQUIRE(C == 256, FGVC_ERR_UNSUPPORTED, "fgvc_normalize_split_f16f6p_nhwc_f32: C=%d unsupported (256 only)", C);
  return normalize_split_common("fgvc_normalize_split_f16f6p_nhwc_f32", in, nullptr, reinterpret_cast<uint16_t*>(rows), N, C, H, W, normalize, 2, stream);
}

int fgvc_normalize_split_f16f6x_nhwc_f32(const float* in, uint8_t* rows, int N, int C, int H, int W, int normalize, void* stream) {
  FGVC_REQUIRE(C == 256, FGVC_ERR_UNSUPPORTED, "fgvc_normalize_split_f16f6x_nhwc_f32: C=%d unsupported (256 only)", C);
  return normalize_split_common("fgvc_normalize_split_f16f6x_nhwc_f32", in, nullptr, reinterpret_cast<uint16_t*>(rows), N, C, H, W, normalize, 3, stream);
}

int fgvc_normalize_nhwc_f32(const float* in, float* out, int N, int C, int H, int W, int normalize, void* stream) {
  FGVC_REQUIRE(in && out, FGVC_ERR_INVALID_ARG, "fgvc_normalize_nhwc_f32: null pointer");
  return fgvc_normalize_split_nhwc_f32(in, out, nullptr, N, C, H, W, normalize, stream);
}

int fgvc_gaussian_labels_f32(const float* points, int P, int Hf, int Wf, int stride, float sigma, float* out,
                             void* stream) {
  FGVC_REQUIRE(points && out, FGVC_ERR_INVALID_ARG, "fgvc_gaussian_labels_f32: null pointer");
  FGVC_REQUIRE(P > 0 && Hf > 0 && Wf > 0 && stride >= 1 && sigma > 0.f, FGVC_ERR_INVALID_ARG, "fgvc_gaussian_labels_f32: bad shape");
  return gaussian_launch(points, P, Hf, Wf, stride, sigma, out, (hipStream_t)stream);
}

size_t fgvc_softargmax_workspace_bytes(int n_frames, int P) {
  if (n_frames < 0 || P < 0) return 0;
  return (size_t)n_frames * P * (softargmax_bands() * (2 * 5 + 1) + 1) * sizeof(float);
}

int fgvc_softargmax_top5_f32(const float* labels, int n_frames, int Hf, int Wf, int P, int h, int w,
                             const float* gauss_points, float sigma, double* coords, void* workspace, void* stream) {
  FGVC_REQUIRE(labels && coords, FGVC_ERR_INVALID_ARG, "fgvc_softargmax_top5_f32: null pointer");
  FGVC_REQUIRE(n_frames >= 0 && Hf > 0 && Wf > 0 && P > 0 && h > 0 && w > 0, FGVC_ERR_INVALID_ARG, "fgvc_softargmax_top5_f32: bad shape");
  FGVC_REQUIRE((long long)h * w < (1ll << 31) && h * w >= 5, FGVC_ERR_UNSUPPORTED, "fgvc_softargmax_top5_f32: h*w out of range");
  FGVC_REQUIRE(n_frames <= 65535 && sigma > 0.f, FGVC_ERR_INVALID_ARG, "fgvc_softargmax_top5_f32: bad n_frames/sigma");
  FGVC_REQUIRE(workspace != nullptr || n_frames == 0, FGVC_ERR_INVALID_ARG,
               "fgvc_softargmax_top5_f32: workspace of fgvc_softargmax_workspace_bytes() bytes required");
  FGVC_REQUIRE(P <= 65535, FGVC_ERR_UNSUPPORTED, "fgvc_softargmax_top5_f32: P > 65535");
  if (n_frames == 0) return FGVC_OK;
  return softargmax_launch(labels, n_frames, Hf, Wf, P, h, w, gauss_points, sigma, coords,
                           static_cast<float*>(workspace), (hipStream_t)stream);
}

}  // extern "C"
