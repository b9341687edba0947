// Shared device/host helpers for libfgvc_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/fgvc_hip.h"
#include "sortnet.hpp"

namespace fgvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WAVE = 64;
constexpr int IDX_EMPTY = 0x7fffffff;

// thread-local error text, exported through fgvc_last_error()
void set_error(const char* fmt, ...);

#define FGVC_REQUIRE(cond, code, ...)        \
  do {                                       \
    if (!(cond)) {                           \
      ::fgvc::set_error(__VA_ARGS__);        \
      return (code);                         \
    }                                        \
  } while (0)

#define FGVC_CHECK_LAUNCH(what)                                                   \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      ::fgvc::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));   \
      return FGVC_ERR_LAUNCH;                                                     \
    }                                                                             \
  } while (0)

// Sorted (score desc, index asc) list of the K best candidates seen so far; lives in VGPRs
// (every loop over K is fully unrolled so the arrays are statically indexed).
template <int K>
struct TopK {
  float v[K];
  int ix[K];

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      v[j] = -INFINITY;
      ix[j] = IDX_EMPTY;
    }
  }
  __device__ __forceinline__ bool accepts(float s, int id) const {
    return s > v[K - 1] || (s == v[K - 1] && id < ix[K - 1]);
  }
  // bubble (s,id) down the list; the displaced tail element falls off the end
  __device__ __forceinline__ void insert(float s, int id) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const bool b = s > v[j] || (s == v[j] && id < ix[j]);
      const float tv = v[j];
      const int ti = ix[j];
      v[j] = b ? s : tv;
      ix[j] = b ? id : ti;
      s = b ? tv : s;
      id = b ? ti : id;
    }
  }
};

// Branch-free list for the hot selection loops: strict float compare only (1 v_cmp + 4 v_cndmask per
// position), every lane runs the network and `pred` masks the update.  Equal scores keep ARRIVAL order, so
// callers that need the canonical (score desc, index asc) order among exact ties re-establish it when they
// merge (TopK::insert does the index tie-break).  With nested divergent `if`s around the list hipcc copies
// the whole list at every join (40 v_mov per level, ~1800 cycles per candidate measured) -- hence no branches.
template <int K>
struct TopKF {
  float v[K];
  int ix[K];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      v[j] = -INFINITY;
      ix[j] = IDX_EMPTY;
    }
  }
  __device__ __forceinline__ bool accepts(float s) const { return s > v[K - 1]; }
  __device__ __forceinline__ void insert_if(bool pred, float s, int id) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const bool b = pred && s > v[j];
      const float tv = v[j];
      const int ti = ix[j];
      v[j] = b ? s : tv;
      ix[j] = b ? id : ti;
      s = b ? tv : s;
      id = b ? ti : id;
    }
  }

  // Data-independent alternative for a whole 32x32 score tile (16 candidates per lane): sort the 16
  // candidates with a Batcher network (63 compare-exchanges), take max(cand[i], list[K-1-i]) -- the top K
  // of the union, a bitonic sequence -- and re-sort K wires.  ~(63 + K-net) x 5 VALU ops per tile, no
  // branches, no thresholds: the per-candidate insertion network above costs 16 x K x 5 and is >90 %
  // wasted on rejected candidates, which made selection (not the MFMAs) the bottleneck.
  __device__ __forceinline__ void merge_tile(float (&cs)[16], int (&ci)[16]);
};

#define FGVC_CSWAP(A, I, J)                      \
  {                                              \
    const bool b_ = A##s[J] > A##s[I];           \
    const float hs_ = b_ ? A##s[J] : A##s[I];    \
    const float ls_ = b_ ? A##s[I] : A##s[J];    \
    const int hi_ = b_ ? A##i[J] : A##i[I];      \
    const int li_ = b_ ? A##i[I] : A##i[J];      \
    A##s[I] = hs_; A##s[J] = ls_;                \
    A##i[I] = hi_; A##i[J] = li_;                \
  }

template <int K>
__device__ __forceinline__ void TopKF<K>::merge_tile(float (&cs)[16], int (&ci)[16]) {
#define X(I, J) FGVC_CSWAP(c, I, J)
  FGVC_SORTNET_16(X)
#undef X
  float (&ls)[K] = v;
  int (&li)[K] = ix;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    if (K - 1 - j < 16) {
      const bool b = cs[K - 1 - j] > ls[j];
      ls[j] = b ? cs[K - 1 - j] : ls[j];
      li[j] = b ? ci[K - 1 - j] : li[j];
    }
  }
#define X(I, J) FGVC_CSWAP(l, I, J)
  if constexpr (K == 16) { FGVC_SORTNET_16(X) }
  else if constexpr (K == 10) { FGVC_SORTNET_10(X) }
  else if constexpr (K == 5) { FGVC_SORTNET_5(X) }
#undef X
}

// Same canonical order on packed 64-bit keys: high word = score mapped to an order-preserving unsigned
// (negative floats bit-flipped, non-negative get the sign bit set), low word = ~index, so a larger key is
// a better candidate (higher score, then LOWER index).  One v_cmp_gt_u64 + 4 v_cndmask per list position
// and, more importantly, a branch-free predicated insertion: with nested divergent `if`s around a float
// list hipcc copies the whole list at every join (40 v_mov per level, measured 1800 cycles/candidate).
// Key 0 = empty slot (decodes to index -1, score -inf).  Scores must not be NaN.
template <int K>
struct TopK64 {
  unsigned long long k[K];

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < K; ++j) k[j] = 0ull;
  }
  static __device__ __forceinline__ unsigned long long make_key(float s, int id) {
    unsigned u = __builtin_bit_cast(unsigned, s + 0.0f);            // -0.0 -> +0.0
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned)(~(unsigned)id);
  }
  __device__ __forceinline__ bool accepts(unsigned long long key) const { return key > k[K - 1]; }
  // every lane runs the network; lanes with pred == false leave their list untouched
  __device__ __forceinline__ void insert_if(bool pred, unsigned long long key) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const bool b = pred && key > k[j];
      const unsigned long long t = k[j];
      k[j] = b ? key : t;
      key = b ? t : key;
    }
  }
  __device__ __forceinline__ int index(int j) const { return (int)(~(unsigned)k[j]); }   // empty -> -1
  __device__ __forceinline__ float score(int j) const {
    if (k[j] == 0ull) return -INFINITY;
    unsigned u = (unsigned)(k[j] >> 32);
    u = (u & 0x80000000u) ? (u ^ 0x80000000u) : ~u;
    return __builtin_bit_cast(float, u);
  }
};

// soft-argmax read-out order: value desc, HIGHER index first among equals
template <int K>
struct TopKHi {
  float v[K];
  int ix[K];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      v[j] = -INFINITY;
      ix[j] = -1;
    }
  }
  __device__ __forceinline__ bool accepts(float s, int id) const {
    return s > v[K - 1] || (s == v[K - 1] && id > ix[K - 1]);
  }
  __device__ __forceinline__ void insert(float s, int id) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const bool b = s > v[j] || (s == v[j] && id > ix[j]);
      const float tv = v[j];
      const int ti = ix[j];
      v[j] = b ? s : tv;
      ix[j] = b ? id : ti;
      s = b ? tv : s;
      id = b ? ti : id;
    }
  }
};

// Blocks are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2): give every XCD a
// CONTIGUOUS chunk of the logical grid so neighbouring tiles (which share key windows) hit one L2.
// Bijective for any n (cdna_hip_programming.md T1).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() also drains vmcnt(0), and on CDNA4 vmcnt
// counts stores: in a loop that streams results to HBM every iteration would wait for its stores to land.
// Global loads feeding LDS are still waited for by the compiler-counted vmcnt at their ds_write.
__device__ __forceinline__ uint16_t f2bf(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);  // RNE; inputs are finite
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

// gfx950 converts in hardware: two f32 -> two bf16 (round to nearest even) in one v_cvt_pk_bf16_f32; low half = a
typedef __bf16 fgvc_bf16x2 __attribute__((ext_vector_type(2)));
typedef float fgvc_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2bf_pk(float a, float b) {
  const fgvc_f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, fgvc_bf16x2));
}
// (hi, lo) split of four f32: hi = bf16(x), lo = bf16(x - hi), as the four-element words the split tensors store
__device__ __forceinline__ void split_bf16_4(const f32x4& x, ushort4& hv, ushort4& lv) {
  const uint32_t h0 = f2bf_pk(x.x, x.y), h1 = f2bf_pk(x.z, x.w);
  const float fx = __builtin_bit_cast(float, h0 << 16), fy = __builtin_bit_cast(float, h0 & 0xFFFF0000u);
  const float fz = __builtin_bit_cast(float, h1 << 16), fw = __builtin_bit_cast(float, h1 & 0xFFFF0000u);
  const uint32_t l0 = f2bf_pk(x.x - fx, x.y - fy), l1 = f2bf_pk(x.z - fz, x.w - fw);
  const uint2 hh = {h0, h1}, ll = {l0, l1};
  hv = __builtin_bit_cast(ushort4, hh);
  lv = __builtin_bit_cast(ushort4, ll);
}

// ---- the f16 forms of the encoder's split activation tensors (conv_split.hip has the arithmetic) ----
typedef _Float16 fgvc_f16x8 __attribute__((ext_vector_type(8)));
typedef int fgvc_i32x8 __attribute__((ext_vector_type(8)));
typedef int fgvc_i32x4 __attribute__((ext_vector_type(4)));

// f16f8 operand scales (powers of two; mirrored by fgvc_amd/ops.py): h8 = e4m3(h 2^-A), l8 = e4m3(l 2^B)
constexpr int F8_AX = 7, F8_BX = 3;     // activations: |h| <= 65504 -> 512 (the top half-binade saturates at 448: flagged as overflow anyway), |l| <= 32 -> 256
constexpr int F8_AW = 2, F8_BW = 9;     // weights (|h| <= 2^10): h -> 2^8, l (<= 2^-1) -> 2^8

// four f32 -> the words of a split tensor.  FMT 1: h (4 f16), l8, h8 (4 e4m3 each);  FMT 2: h, l (4 f16 each).  `ovf` is raised when
// |s x| leaves the f16 range.
__device__ __forceinline__ void split_f16_4(const f32x4& x, float s, uint2& hw, uint32_t& l8, uint32_t& h8, uint2& lw, bool& ovf) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  f16x4 h, l;
  float hf[4], lf[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float xs = x[i] * s;
    ovf |= fabsf(xs) > 57344.f;                       // 448 * 2^F8_AX: beyond it the e4m3 form of h would have to saturate
    const float c = __builtin_amdgcn_fmed3f(xs, -65504.f, 65504.f);
    h[i] = (_Float16)c;
    hf[i] = (float)h[i];
    lf[i] = c - hf[i];
    l[i] = (_Float16)lf[i];
  }
  hw = __builtin_bit_cast(uint2, h);
  lw = __builtin_bit_cast(uint2, l);
  constexpr float SA = 1.0f / (float)(1 << F8_AX), SB = (float)(1 << F8_BX);
  int a = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(hf[0] * SA, -448.f, 448.f), __builtin_amdgcn_fmed3f(hf[1] * SA, -448.f, 448.f), 0, false);
  a = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(hf[2] * SA, -448.f, 448.f), __builtin_amdgcn_fmed3f(hf[3] * SA, -448.f, 448.f), a, true);
  int b = __builtin_amdgcn_cvt_pk_fp8_f32(lf[0] * SB, lf[1] * SB, 0, false);
  b = __builtin_amdgcn_cvt_pk_fp8_f32(lf[2] * SB, lf[3] * SB, b, true);
  h8 = (uint32_t)a;
  l8 = (uint32_t)b;
}

// FMT 3 (f16 + FP6, round 4): the words of one (pixel, 32-channel chunk) of a split tensor [h 64 B | slot 4 | slot 5 | slot 6 | slot 7]:
// h = f16(s x) as in FMT 1; the two 32-element FP6 (e2m3) operands of the chunk with one E8M0 scale each,
//     h6 = e2m3(h / 2^sh),   l6 = e2m3(f16(2^11 (s x - h)) / 2^sl),   2^s the smallest power of two that keeps the block's largest
//     magnitude at or below 7.5 (v_cvt_scalef32_pk32_fp6_f16: round to nearest even), element e = channel e in bits [6 e, 6 e + 6)
//     of a 24-byte string = 16 bytes "main" + 8 bytes "tail";
// slot 4 = l6 main, slot 5 = h6 main, slot 6 = [l6 tail | byte 127 + sl - 11 | 0], slot 7 = [h6 tail | byte 127 + sh | 0]: lane half hh of
// v_mfma_scale_f32_32x32x64_f8f6f4 reads slots 4 + hh and 6 + hh as eight registers -- six of FP6 data (the instruction ignores the
// rest), the seventh carries its block's scale byte.  (Weights: the same with h6 in slots 4 / 6 and l6 in 5 / 7, so that lane half 0
// multiplies h6_w . l6_x and lane half 1 l6_w . h6_x: ops.prepare_conv_split_f16.)
// Called by a whole wave with the accumulator layout of the convolution epilogues: lane (pixel n, half h) holds channels
// 8 g + 4 h + k of the chunk in v[g][k].  The halves trade registers (v_permlane32_swap) so that lane (n, 0) owns all 32 h values
// and lane (n, 1) all 32 residuals: one conversion instruction each, no cross-lane maximum.  Returns the lane's own h words
// (hw[g]: channels 8 g + 4 h ..+4) and the 2 x 16 bytes it stores at byte 80 - 16 h (main6) and 112 - 16 h (tail6) of the row.
typedef int fgvc_i32x6 __attribute__((ext_vector_type(6)));
typedef int fgvc_i32x16 __attribute__((ext_vector_type(16)));
// Round 6, fewer instructions for the same bits (the conversion was 12.7 vector instructions per value in conv256p_kernel's epilogue):
//   * ONE running maximum M = max |c| over the clamped values gives both the overflow flag (|s x| > 57344 <=> |c| > 57344: the clamp sits
//     above the bound) and the block's largest |h| (rounding to f16 is monotonic and odd: max |f16(c)| = f16(max |c|));
//   * the residual's maximum is taken over c - h (exact) and scaled by 2^11 once per block (exact);
//   * `lo`: the clamp's lower bound -- a caller whose ReLU has no other reader passes 0 instead of applying it (v_med3 does both).
__device__ __forceinline__ void split_f16f6_chunk(const f32x4 (&v)[4], float s, int h, uint2 (&hw)[4], fgvc_i32x4& main6, fgvc_i32x4& tail6,
                                                  bool& ovf, float lo = -65504.f) {
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  unsigned A[8], B[8];                 // A: packed h pairs, B: packed f16(2^11 l) pairs; register 2 g + j = channels 8 g + 4 h + 2 j, + 1
  float mc = 0.f, md = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float c0 = __builtin_amdgcn_fmed3f(v[g][2 * j] * s, lo, 65504.f);
      const float c1 = __builtin_amdgcn_fmed3f(v[g][2 * j + 1] * s, lo, 65504.f);
      mc = fmaxf(fmaxf(mc, fabsf(c0)), fabsf(c1));
      f16x2 hp, lp;
      hp[0] = (_Float16)c0;
      hp[1] = (_Float16)c1;
      const float d0 = c0 - (float)hp[0], d1 = c1 - (float)hp[1];      // exact (the low bits of c)
      md = fmaxf(fmaxf(md, fabsf(d0)), fabsf(d1));
      lp[0] = (_Float16)(d0 * 2048.f);
      lp[1] = (_Float16)(d1 * 2048.f);
      A[2 * g + j] = __builtin_bit_cast(unsigned, hp);
      B[2 * g + j] = __builtin_bit_cast(unsigned, lp);
    }
  ovf |= mc > 57344.f;                                   // (the bound of FMT 1, kept: one calibration rule for both formats)
  const float mh = (float)(_Float16)mc;
  const float ml = md * 2048.f;                          // = max |2^11 (c - h)|: the scaling is exact
#pragma unroll
  for (int g = 0; g < 4; ++g) hw[g] = {A[2 * g], A[2 * g + 1]};
  // upper lanes of A <-> lower lanes of B: lane (n, 0) then holds h of channels 8 g + k (A) and 8 g + 4 + k (B), lane (n, 1) the residuals
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const auto t = __builtin_amdgcn_permlane32_swap(A[r], B[r], false, false);
    A[r] = t[0]; B[r] = t[1];
  }
  // (the maximum of the two halves as UNSIGNED integers -- the order of non-negative floats: hipcc 7.2 folds fmaxf() of the two
  // results of the swap builtin into its first result alone, tools/experiments/dbg_act_f16f6.py found the blocks one scale short)
  const auto tm = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mh), __builtin_bit_cast(unsigned, ml), false, false);
  const unsigned mu = tm[0] > tm[1] ? tm[0] : tm[1];
  const float m16 = (float)(_Float16)__builtin_bit_cast(float, mu);     // the block's largest f16 magnitude (rounding is monotonic: = the largest rounded value)
  const unsigned mb = __builtin_bit_cast(unsigned, m16);
  int field = (int)(mb >> 23) - 2 + ((mb & 0x7fffffu) > 0x700000u ? 1 : 0);    // 127 + s: m / 2^s in (3.75, 7.5]
  field = field < 32 ? 32 : field;                      // (an all-zero block: any scale)
  fgvc_i32x16 T;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      T[4 * g + j] = (int)A[2 * g + j];                 // channels 8 g + 2 j, + 1
      T[4 * g + 2 + j] = (int)B[2 * g + j];             // channels 8 g + 4 + 2 j, + 1
    }
  fgvc_i32x6 c6;
  const unsigned sc_bits = (unsigned)field << 23;
  asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(c6) : "v"(T), "v"(sc_bits));   // (early-clobber: not on top of the scale)
  main6 = {c6[0], c6[1], c6[2], c6[3]};
  tail6 = {c6[4], c6[5], field - (h ? 11 : 0), 0};
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__host__ __device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__host__ __device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__host__ __device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace fgvc
