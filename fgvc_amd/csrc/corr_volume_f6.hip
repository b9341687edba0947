// Dense correlation volume, parity-grade, at 1.5 bf16-MFMA times per tile (f16f8: 2, bf16x3: 3).
//
// Arithmetic.  As fgvc_corr_volume_f16f8 (corr_volume_f8.hip), with the cross terms in block-scaled FP6 instead of FP8:
//     h  = f16(256 x)               11 significand bits, exact products h_k * h_q in f32
//     h6 = e2m3(h / 2^sh)           h again: 4 significand bits, E8M0 scale 2^sh per 32 consecutive channels
//     l6 = e2m3(l / 2^sl)           l = 256 (256 x - h), the residual of h, same format
//     2^16 <k, q>  =  sum h_k h_q  +  2^-8 (sum h6_k l6_q  +  sum l6_k h6_q)          (sum l l = 2^-22, dropped)
// The cross sums are 2^-11 of the first sum, so their 4 significand bits carry them to ~2^-15 of the result; e2m3 has only
// three binades (values 0 .. 7.5), the per-block scale puts the block maximum into the top one, and an element 8x below its
// block's maximum still has 3 bits.  Simulated against float64 before it was built (tools/sim_split_formats.py): the error is the
// same as fp8's (6e-5 logit on Gaussian rows, 3.4e-4 on adversarially sparse / one-hot / heavy-tailed rows; bound asserted: 1e-3);
// FP4 (e2m1) would miss the bar (1.3e-3).  v_mfma_scale_f32_16x16x128_f8f6f4 retires a K-128 block of FP6 operands in the 16
// cycles the f16 form needs for K 32, so per 32 x 32 tile and C = 256: 32 f16 MFMAs + 16 scaled FP6 MFMAs = 512 + 256 = 768
// matrix-pipe cycles (f16f8: 1024, bf16x3: 1536).
// Operand layout of the FP6 form (probed with exact integers, tools/micro/probe_fp6_scaled.hip): lane (r = l & 15, g = l >> 4)
// holds row / column r, k = 32 g + i, element i in bits [6 i, 6 i + 6) of its SIX operand registers; the scale byte that opsel picks
// from lane (r, g)'s scale register applies to exactly those 32 elements -- one scale block per lane.
//
// Row format (fgvc_split_f16f6), 1 KiB per pixel, C = 256:
//     [0, 512)    h, 256 f16
//     [512, 704)  h6: K-128 block u at 512 + 96 u = [lane piece 0: 4 x 16 B (g = 0..3)][lane piece 1: 4 x 8 B]; lane g's 24 bytes
//                 (16 + 8) are channels 128 u + 32 g + i, i = 0..31, 6 bits each, little-endian
//     [704, 896)  l6, the same layout
//     [896, 912)  scales: dword g = {sh(u=0,g), sh(u=1,g), sl(u=0,g), sl(u=1,g)} as E8M0 bytes of 2^(s-4): the two scales of a
//                 product carry the 2^-8 between them
//     [912, 1024) zero
// Structure: corr_volume_f16f8_v2_kernel (query fragments resident as B operands, 64-key stages through LDS by LDS-DMA issued by
// the older waves, 16 x 16 MFMA shapes, lane-swapped 128-byte row pieces, row classes for whole-line stores, wave stagger).
#include "common.hpp"

namespace fgvc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr float F6_S = 256.f;
constexpr int F6_ROWB = 1024, F6_H6 = 512, F6_L6 = 704, F6_SC = 896, F6_END = 912;

// E8M0 exponent s with max / 2^s <= 7.5 (the largest e2m3 value); an all-zero block gets a small harmless scale
__device__ __forceinline__ int f6_scale_exp(float m) {
  if (!(m > 0.f)) return -40;
  int e;
  const float f = frexpf(m * (1.0f / 7.5f), &e);       // m / 7.5 = f 2^e, f in [0.5, 1)
  int s = (f > 0.5f) ? e : e - 1;                       // ceil(log2(m / 7.5))
  if (m * exp2f((float)-s) > 7.5f) ++s;                 // guards the rounding of the division above
  return imax(s, -40);
}

// |y| <= 7.5 -> e2m3 code, round to nearest even
__device__ __forceinline__ unsigned f6_code(float y) {
  const float a = fabsf(y);
  const float inv_step = a < 2.f ? 8.f : (a < 4.f ? 4.f : 2.f);
  const float r = fminf(__builtin_rintf(a * inv_step) / inv_step, 7.5f);
  const float c = r < 2.f ? 8.f * r : (r < 4.f ? 8.f + 4.f * r : 16.f + 2.f * r);
  return (unsigned)c | (y < 0.f ? 32u : 0u);
}

// 32 codes -> 192 bits
__device__ __forceinline__ void f6_pack(const unsigned (&c)[32], unsigned (&w)[6]) {
#pragma unroll
  for (int i = 0; i < 6; ++i) w[i] = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int bit = 6 * i, word = bit >> 5, sh = bit & 31;
    w[word] |= c[i] << sh;
    if (sh > 26) w[word + 1] |= c[i] >> (32 - sh);
  }
}

// one thread per (pixel, 32-channel block)
__global__ __launch_bounds__(256) void split_f16f6_kernel(const float* __restrict__ feat, unsigned char* __restrict__ out,
                                                           long long n_blocks) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_blocks) return;
  const long long pix = t >> 3;
  const int blk = (int)(t & 7), u = blk >> 2, g = blk & 3;
  const float* src = feat + pix * 256 + 32 * blk;
  float hf[32], lf[32];
  unsigned char* row = out + pix * F6_ROWB;
  float mh = 0.f, ml = 0.f;
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(src + 4 * v);
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    f16x4 hv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xs = x[j] * F6_S;
      const _Float16 h = (_Float16)xs;
      hv[j] = h;
      hf[4 * v + j] = (float)h;
      lf[4 * v + j] = (xs - (float)h) * F6_S;
      mh = fmaxf(mh, fabsf(hf[4 * v + j]));
      ml = fmaxf(ml, fabsf(lf[4 * v + j]));
    }
    *reinterpret_cast<f16x4*>(row + 64 * blk + 8 * v) = hv;
  }
  const int sh = f6_scale_exp(mh), sl = f6_scale_exp(ml);
  const float ih = exp2f((float)-sh), il = exp2f((float)-sl);
  unsigned ch[32], cl[32], wh[6], wl[6];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    ch[i] = f6_code(hf[i] * ih);
    cl[i] = f6_code(lf[i] * il);
  }
  f6_pack(ch, wh);
  f6_pack(cl, wl);
  const i32x4 h4 = {(int)wh[0], (int)wh[1], (int)wh[2], (int)wh[3]}, l4 = {(int)wl[0], (int)wl[1], (int)wl[2], (int)wl[3]};
  const i32x2 h2 = {(int)wh[4], (int)wh[5]}, l2 = {(int)wl[4], (int)wl[5]};
  *reinterpret_cast<i32x4*>(row + F6_H6 + 96 * u + 16 * g) = h4;
  *reinterpret_cast<i32x2*>(row + F6_H6 + 96 * u + 64 + 8 * g) = h2;
  *reinterpret_cast<i32x4*>(row + F6_L6 + 96 * u + 16 * g) = l4;
  *reinterpret_cast<i32x2*>(row + F6_L6 + 96 * u + 64 + 8 * g) = l2;
  row[F6_SC + 4 * g + u] = (unsigned char)(sh + 127 - 4);
  row[F6_SC + 4 * g + 2 + u] = (unsigned char)(sl + 127 - 4);
  if (blk < 7) *reinterpret_cast<i32x4*>(row + F6_END + 16 * blk) = i32x4{0, 0, 0, 0};
}

int split_f16f6_launch(const float* feat, unsigned char* out, long long n_pixels, hipStream_t s) {
  const long long nb = n_pixels * 8;
  split_f16f6_kernel<<<(unsigned)((nb + 255) / 256), 256, 0, s>>>(feat, out, nb);
  FGVC_CHECK_LAUNCH("fgvc_split_f16f6");
  return FGVC_OK;
}

// six-register FP6 operand from its 16-byte and 8-byte pieces (registers 6 and 7 of the builtin's type stay undefined: the
// instruction reads six)
__device__ __forceinline__ i32x8 f6_operand(const i32x4& a, const i32x2& b) {
  const i32x8 a8 = __builtin_shufflevector(a, a, 0, 1, 2, 3, -1, -1, -1, -1);
  const i32x8 b8 = __builtin_shufflevector(b, b, 0, 1, -1, -1, -1, -1, -1, -1);
  return __builtin_shufflevector(a8, b8, 0, 1, 2, 3, 8, 9, -1, -1);
}

template <int NW, int DEBUG, bool SDMA = true>   // SDMA: staging DMAs with a scalar base (A/B); DEBUG: 1 = no volume stores, 2 = no MFMAs (results wrong); results right: 8 = no wave stagger,
                               // 16 = DMA spread over both tiles of a stage, 64 = F fragments re-read in one piece, 128 = pair-major workgroup ids; 256 / 512 = no f16 / no FP6 MFMAs (results wrong);
                               // 32 = s_memtime probe of one workgroup, written over the first floats of vol
__global__ __launch_bounds__(NW * 64, 2) void corr_volume_f16f6_kernel(const unsigned char* __restrict__ q_sp,
                                                                      const unsigned char* __restrict__ k_sp, int HWq, int HWk,
                                                                      float out_scale, float* __restrict__ vol, int s_tile,
                                                                      int c_half, int n_tiles, int period, int m32, int skew) {
  constexpr int SUB = 2, ROWB = F6_ROWB, LDB = ROWB + 32, ROWS = 32 * SUB, BUFB = ROWS * LDB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  // static wave priorities (round 4's attempt on this kernel; corr6_skew bits 8 / 9 / 10: younger half 1, older half 1, younger half 3)
  if ((skew & 0x100) && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  if ((skew & 0x200) && wave < NW / 2) __builtin_amdgcn_s_setprio(1);
  if ((skew & 0x400) && wave >= NW / 2) __builtin_amdgcn_s_setprio(3);
  skew &= 0xff;
  // Work split.  A tile = (256-query column tile xq, row class cls) with s_tile 64-key stages; tiles are taken in PAIRS (2 P, 2 P + 1)
  // and a pair's 2 s_tile stages are cut into c_half equal pieces, one per workgroup: c_half / 2 key chunks per tile.  An odd c_half
  // lets a workgroup run from the tail of one tile into the head of the next (two segments = two prologues): 2.5 chunks per tile at
  // 480p are 505 workgroups = 1.97 rounds over the 256 CUs, 2.4 prologues per CU instead of the 4 of five whole chunks.  Workgroups
  // with the same piece index walk the same key rows in step (the L2 serves them once).
  // (Tried: the linear stage space of all tiles in 256 / 512 / 768 / 1024 equal pieces -- any grid size, e.g. exactly one workgroup
  // per CU, but no two workgroups in step on the same key rows: 0.86 ms against 0.65 at 480p for every grid size.  The L2 sharing of
  // the key rows between the workgroups of a piece index is worth more than the rounds.)
  // Workgroup -> (piece, pair), piece-major over an XCD-contiguous rank: blocks b and b + 8 share an XCD (and its private L2), so the
  // ~n_pairs workgroups of a piece index -- the ones that read the same key rows in step -- sit on 2-3 XCDs instead of all 8 and
  // a key row crosses the fabric 2-3 times per piece instead of 8 (DEBUG & 128: pair-major ids, every piece on every XCD).
  const int n_pairs_ = (int)gridDim.x / c_half;
  int pair, piece;
  if constexpr (DEBUG & 128) {
    pair = blockIdx.x / c_half;
    piece = blockIdx.x - pair * c_half;
  } else {
    const int rank = xcd_remap(blockIdx.x, gridDim.x);
    piece = rank / n_pairs_;
    pair = rank - piece * n_pairs_;
  }
  // piece boundaries: equal cuts, except that with an odd c_half the middle piece -- the one that runs from one tile into the next
  // and pays a second prologue -- is `(c_half - 1) skew` stages shorter and every other piece `skew` stages longer
  auto cut = [&](int i) {
    const int base = (int)((long long)i * 2 * s_tile / c_half);
    if (!(c_half & 1) || i == 0 || i == c_half) return base;
    const int m = c_half >> 1;
    return base + (i <= m ? skew * i : skew * (i - 1) - (c_half - 1) * skew);
  };
  const int r0 = cut(piece), r1 = cut(piece + 1);
  auto run_segment = [&](int tile_idx, int st0, int st1) {        // stages [st0, st1) of tile tile_idx
  const int xq = tile_idx / period;
  const int cls = tile_idx - xq * period;                       // row class: key rows j = period * v + cls
  const int shift = (cls * m32) & 31;                           // its query tiles start `shift` queries early
  const int qw0 = xq * (NW * 32) + wave * 32 - shift;           // wave-uniform: first query of this wave's tile
  const int n_v = (HWk - cls + period - 1) / period;            // virtual rows of this class

  const bool probe = (DEBUG & 32) != 0;   // s_memtime probe: waves 0 and 4 of every workgroup leave a record in row 0 of vol
  long long p_start = 0, p_pro = 0, p_comp = 0, p_store = 0, p_sync = 0, c0 = 0;
  int p_stages = 0;
  long long p_rt0 = 0;
  if (probe) {
    p_start = __builtin_amdgcn_s_memtime();
    p_rt0 = (long long)__builtin_amdgcn_s_memrealtime();
  }
  // query fragments (B operands) of the two query halves
  f16x8 bq16[2][8];
  i32x8 bq6h[2][2], bq6l[2][2];
  int sq[2];
  if constexpr (SDMA && NW == 8) {
    // The query rows come through the LDS, whole rows by LDS-DMA (one coalesced KiB per instruction), in two passes of 128 rows (the
    // two stage buffers hold 128): read straight from global memory in the MFMA layout, every 16-lane group of a load touches 16
    // different rows -- 64 cache-line requests per instruction, 13 000 cycles of address path per prologue.
    const int xq0 = xq * (NW * 32) - shift;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int j = wave * 16 + i;                                   // row of the staging area, 0..127
        const int qrow = imin(imax(xq0 + pass * 128 + j, 0), HWq - 1);
        const unsigned char* src = q_sp + (size_t)qrow * ROWB;
        const uint32_t dst = (uint32_t)(size_t)(const __attribute__((address_space(3))) unsigned char*)&smem[j * LDB];
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(16u * (uint32_t)lane), "s"(src), "s"(dst) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if ((wave >> 2) == pass) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const unsigned char* qp = &smem[((wave & 3) * 32 + 16 * qt + r) * LDB];
#pragma unroll
          for (int t = 0; t < 8; ++t) bq16[qt][t] = *reinterpret_cast<const f16x8*>(qp + 16 * g + 64 * t);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            bq6h[qt][u] = f6_operand(*reinterpret_cast<const i32x4*>(qp + F6_H6 + 96 * u + 16 * g),
                                     *reinterpret_cast<const i32x2*>(qp + F6_H6 + 96 * u + 64 + 8 * g));
            bq6l[qt][u] = f6_operand(*reinterpret_cast<const i32x4*>(qp + F6_L6 + 96 * u + 16 * g),
                                     *reinterpret_cast<const i32x2*>(qp + F6_L6 + 96 * u + 64 + 8 * g));
          }
          sq[qt] = *reinterpret_cast<const int*>(qp + F6_SC + 4 * g);
        }
      }
      __syncthreads();                                                 // the rows have been read: the area is free again
    }
  } else {
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const unsigned char* qp = q_sp + (size_t)imin(imax(qw0 + 16 * qt + r, 0), HWq - 1) * ROWB;
#pragma unroll
    for (int t = 0; t < 8; ++t) bq16[qt][t] = *reinterpret_cast<const f16x8*>(qp + 16 * g + 64 * t);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bq6h[qt][u] = f6_operand(*reinterpret_cast<const i32x4*>(qp + F6_H6 + 96 * u + 16 * g),
                               *reinterpret_cast<const i32x2*>(qp + F6_H6 + 96 * u + 64 + 8 * g));
      bq6l[qt][u] = f6_operand(*reinterpret_cast<const i32x4*>(qp + F6_L6 + 96 * u + 16 * g),
                               *reinterpret_cast<const i32x2*>(qp + F6_L6 + 96 * u + 64 + 8 * g));
    }
    sq[qt] = *reinterpret_cast<const int*>(qp + F6_SC + 4 * g);
  }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
    for (int t = 0; t < 8; ++t) asm volatile("" ::"v"(bq16[qt][t]));
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      asm volatile("" ::"v"(bq6h[qt][u]));
      asm volatile("" ::"v"(bq6l[qt][u]));
    }
    asm volatile("" ::"v"(sq[qt]));
  }
  const int kb0 = 2 * st0;                            // in units of 32 virtual rows
  const int kb1 = imin(2 * st1, cdiv(n_v, 32));
  auto stage_load = [&](int kb, int buf) {
#pragma unroll
    for (int i = 0; i < ROWS / NW; ++i) {
      const int row = wave * (ROWS / NW) + i;
      const int pix = imin((kb * 32 + row) * period + cls, HWk - 1);
      if constexpr (SDMA) {
        const unsigned char* src = k_sp + (size_t)pix * ROWB;
        const uint32_t dst = (uint32_t)(size_t)(const __attribute__((address_space(3))) unsigned char*)&smem[buf * BUFB + row * LDB];
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(16u * (uint32_t)lane), "s"(src), "s"(dst) : "memory");
      } else {
        const unsigned char* src = k_sp + (size_t)pix * ROWB + 16 * lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)&smem[buf * BUFB + row * LDB], 16, 0, 0);
      }
    }
  };
  // waves 0-3 (the older wave of each SIMD: it wins the issue arbitration and would otherwise wait at the stage barrier) stage
  // all 64 rows of the next stage, two per multiply part; waves 4-7 none (corr_volume_f8.hip, s_memtime probe)
  const bool stager = wave < NW / 2;
  // (the row is wave-uniform: scalar base + one lane-offset register, no 64-bit vector address arithmetic per row)
  const uint32_t dma_lane_off = 16u * (uint32_t)lane;
  auto stage_row = [&](int kb, int buf, int i) {
    const int row = wave * (2 * ROWS / NW) + i;
    const int pix = imin((kb * 32 + row) * period + cls, HWk - 1);
    if constexpr (SDMA) {
      const unsigned char* src = k_sp + (size_t)pix * ROWB;
      const uint32_t dst = (uint32_t)(size_t)(const __attribute__((address_space(3))) unsigned char*)&smem[buf * BUFB + row * LDB];
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(dma_lane_off), "s"(src), "s"(dst) : "memory");
    } else {
      const unsigned char* src = k_sp + (size_t)pix * ROWB + 16 * lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)&smem[buf * BUFB + row * LDB], 16, 0, 0);
    }
  };
  static_assert(2 * ROWS / NW == 16, "two DMA rows per multiply part of a staging wave: 2 tiles x 4 parts");
  if constexpr (!((DEBUG & 1024) != 0)) stage_load(kb0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the DMAs are inline assembly: the compiler's own wait at the barrier does not count them
  __syncthreads();
  if (probe) p_pro = __builtin_amdgcn_s_memtime() - p_start;

  const size_t row_pitch = (size_t)period * HWq;
  const int scol = qw0 + (lane & 31);                                // the column this lane STORES (after the lane swap)
  const int lane_off = 8 * (lane >> 5) * period * HWq + scol;        // + row 8 (lane >> 5) of the row pair a store covers
  const bool wave_full = qw0 >= 0 && qw0 + 31 < HWq;
  const bool defer = (DEBUG & 8) ? false : wave >= NW / 2;
  f32x4 acc[2][2];
  int pend_v = -1, n_counted = 0;

  auto store_tile = [&](int vrow0) {
    if constexpr (DEBUG & 1) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) asm volatile("" ::"v"(acc[kt][qt]));
      n_counted = -1;
      return;
    }
    float* tile = vol + (size_t)(vrow0 * period + cls) * HWq;
    const bool full = wave_full && vrow0 + 32 <= n_v;                 // wave-uniform
    if (!full) n_counted = -1;
    else if (n_counted >= 0) n_counted += 16;
    // x = rows 16 kt + 4 g + i of query half 0, y = the same rows of query half 1  ->  after the swap x = rows 16 kt + i (+ 8)
    // x 32 queries, y = rows 16 kt + 4 + i (+ 8).  Inline assembly with hand-placed wait states (see corr_volume_f8.hip).
    float x[8], y[8];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x[4 * kt + i] = acc[kt][0][i] * out_scale;
        y[4 * kt + i] = acc[kt][1][i] * out_scale;
      }
    asm volatile("s_nop 4" ::: "memory");
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[e]), "+v"(y[e]));
    asm volatile("s_nop 1" ::: "memory");
    if (full) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float* p0 = tile + (size_t)(16 * (e >> 2) + (e & 3)) * row_pitch;
        __builtin_nontemporal_store(x[e], p0 + lane_off);
        __builtin_nontemporal_store(y[e], p0 + 4 * row_pitch + lane_off);
      }
    } else if (scol >= 0 && scol < HWq) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r0 = 16 * (e >> 2) + (e & 3), r1 = r0 + 4;
        float* p0 = tile + (size_t)r0 * row_pitch;
        if (8 * (lane >> 5) < n_v - vrow0 - r0) __builtin_nontemporal_store(x[e], p0 + lane_off);
        if (8 * (lane >> 5) < n_v - vrow0 - r1) __builtin_nontemporal_store(y[e], p0 + 4 * row_pitch + lane_off);
      }
    }
  };

  // A operand register sets: F = the 8 f16 fragments of a K-128 block (2 key halves x 4 K-32 steps), P = its FP6 operands
  // (2 key halves x {h6, l6}); S = the scale dwords of a tile's two key halves.  P of a block is read at the start of its F
  // part, F of the next block at the start of the P part.
  f16x8 F[2][4];
  i32x8 Ph[2], Pl[2];
  int S[2];
  auto load_F = [&](const unsigned char* ka, int u) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) F[kt][tt] = *reinterpret_cast<const f16x8*>(ka + 16 * kt * LDB + 16 * g + 64 * (4 * u + tt));
  };
  auto load_P = [&](const unsigned char* ka, int u) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const unsigned char* p = ka + 16 * kt * LDB + F6_H6 + 96 * u;
      Ph[kt] = f6_operand(*reinterpret_cast<const i32x4*>(p + 16 * g), *reinterpret_cast<const i32x2*>(p + 64 + 8 * g));
      Pl[kt] = f6_operand(*reinterpret_cast<const i32x4*>(p + 192 + 16 * g), *reinterpret_cast<const i32x2*>(p + 192 + 64 + 8 * g));
    }
  };
  auto load_S = [&](const unsigned char* ka) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) S[kt] = *reinterpret_cast<const int*>(ka + 16 * kt * LDB + F6_SC + 4 * g);
  };

  auto load_F_half = [&](const unsigned char* ka, int u, int half) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int tt = 2 * half; tt < 2 * half + 2; ++tt)
        F[kt][tt] = *reinterpret_cast<const f16x8*>(ka + 16 * kt * LDB + 16 * g + 64 * (4 * u + tt));
  };
  // DMA placement.  vmcnt counts loads and stores in issue order, so the wait that retires a stage's DMA also waits for every
  // OLDER store to be acknowledged by memory -- microseconds under a saturated write stream.  All 16 rows of a staging wave are
  // therefore issued inside the stage's FIRST tile, ahead of both store bursts of the stage: the wait before the barrier is then
  // vmcnt(32) and covers nothing younger than the previous stage's stores.  (DEBUG & 16: the earlier placement, 8 rows per tile,
  // vmcnt(16): waits for the first tile's burst.)
  constexpr bool DMA_EARLY = !(DEBUG & 16);
  constexpr bool F_HALVES = !(DEBUG & 64);
  // DEBUG & 1024: the kernel's STORES alone -- same grid, same row classes, same lane swap and address sequence, same barriers;
  // no key-row DMA, no LDS reads, no MFMAs (zeros are stored): what the write stream by itself costs (bench.py: store_replay)
  constexpr bool STORE_ONLY = (DEBUG & 1024) != 0;
  auto stage_rows = [&](int kb, int buf, int sb, int slot) {   // slot 0..3 of tile sb
    if constexpr (DMA_EARLY) {
      if (sb == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_row(kb, buf, 4 * slot + i);
      }
    } else {
      stage_row(kb, buf, 8 * sb + 2 * slot);
      stage_row(kb, buf, 8 * sb + 2 * slot + 1);
    }
  };

  int buf = 0;
  for (int kb = kb0; kb < kb1; kb += SUB) {
    const bool more = kb + SUB < kb1;
    n_counted = 0;
    if constexpr (!STORE_ONLY) load_F(&smem[buf * BUFB + r * LDB], 0);
    if (probe) c0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int sb = 0; sb < SUB; ++sb) {
      if (kb + sb >= kb1) break;                       // wave-uniform (ragged tail of the chunk)
      const unsigned char* ka = &smem[buf * BUFB + (sb * 32 + r) * LDB];
      const bool next_here = sb + 1 < SUB && kb + sb + 1 < kb1;
      __builtin_amdgcn_sched_barrier(0);
      if (defer && pend_v >= 0) store_tile(pend_v);    // waves 4-7: the previous tile leaves under the partner's multiplies
      if (probe && defer) { const long long c1 = __builtin_amdgcn_s_memtime(); p_store += c1 - c0; c0 = c1; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) acc[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (!STORE_ONLY) load_S(ka);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if constexpr (STORE_ONLY) continue;
        load_P(ka, u);                                 // lands during the F part
        if (more && stager) stage_rows(kb + SUB, buf ^ 1, sb, 2 * u);
        // the F fragments of the NEXT block: the first two K-32 steps are re-read as soon as their last reader has issued, the
        // other two at the start of the P part (256 pipe cycles before their first reader: not enough on their own)
        const unsigned char* nka = (u == 0) ? ka : ka + 32 * LDB;
        const bool nxt = (u == 0) || next_here;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if constexpr (!(DEBUG & 2) && !(DEBUG & 256)) {
#pragma unroll
            for (int tt = 2 * half; tt < 2 * half + 2; ++tt)
#pragma unroll
              for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                  acc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(F[kt][tt], bq16[qt][4 * u + tt], acc[kt][qt], 0, 0, 0);
          } else {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
              for (int tt = 2 * half; tt < 2 * half + 2; ++tt) asm volatile("" ::"v"(F[kt][tt]));
          }
          __builtin_amdgcn_sched_barrier(0);
          if (F_HALVES) {
            if (nxt) load_F_half(nka, 1 - u, half);
          } else if (half == 1) {
            if (nxt) load_F(nka, 1 - u);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (more && stager) stage_rows(kb + SUB, buf ^ 1, sb, 2 * u + 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(DEBUG & 2) && !(DEBUG & 512)) {
          // scale bytes (opsel): 0 / 1 = h6 of block u = 0 / 1, 2 / 3 = l6; each is 2^(s - 4), so a product enters at 2^-8
#define FGVC_F6_MFMA(A, B, OA, OB) \
  acc[kt][qt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc[kt][qt], 2, 2, OA, S[kt], OB, sq[qt])
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
              if (u == 0) FGVC_F6_MFMA(Ph[kt], bq6l[qt][0], 0, 2);
              else FGVC_F6_MFMA(Ph[kt], bq6l[qt][1], 1, 3);
            }
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
              if (u == 0) FGVC_F6_MFMA(Pl[kt], bq6h[qt][0], 2, 0);
              else FGVC_F6_MFMA(Pl[kt], bq6h[qt][1], 3, 1);
            }
#undef FGVC_F6_MFMA
        } else {
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            asm volatile("" ::"v"(Ph[kt]));
            asm volatile("" ::"v"(Pl[kt]));
            asm volatile("" ::"v"(S[kt]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const int vrow0 = (kb + sb) * 32;
      if (probe) { const long long c1 = __builtin_amdgcn_s_memtime(); p_comp += c1 - c0; c0 = c1; }
      if (defer) pend_v = vrow0;
      else store_tile(vrow0);
      if (probe && !defer) { const long long c1 = __builtin_amdgcn_s_memtime(); p_store += c1 - c0; c0 = c1; }
    }
    if (more && stager) {
      // staging waves never defer: their stores of this stage are the 16 x (whole tiles stored) youngest operations
      if (DMA_EARLY && n_counted == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else if (!DMA_EARLY && n_counted == 32 && kb + 1 < kb1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();
    if (probe) { p_sync += __builtin_amdgcn_s_memtime() - c0; ++p_stages; }
    buf ^= 1;
  }
  if (defer && pend_v >= 0) store_tile(pend_v);
  if (probe && lane == 0 && (wave & 3) == 0) {
    // 48-byte records over the first floats of vol (tools/experiments/time_corr6.py); row 0 is only written by the first stage of the
    // class-0 workgroups of the first key chunk, long before any workgroup ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int wg = blockIdx.x;
    int* o = reinterpret_cast<int*>(vol) + 12 * (2 * (2 * wg + (tile_idx & 1)) + (wave >> 2));   // one record per segment and wave
    const long long now = __builtin_amdgcn_s_memtime();
    o[0] = (int)p_pro; o[1] = (int)p_comp; o[2] = (int)p_store; o[3] = (int)p_sync; o[4] = (int)(now - p_start); o[5] = p_stages;
    *reinterpret_cast<long long*>(o + 6) = p_rt0;                       // 100 MHz, the same counter on every CU
    *reinterpret_cast<long long*>(o + 8) = (long long)__builtin_amdgcn_s_memrealtime();
    o[10] = (int)__builtin_amdgcn_s_getreg(6 << 11 | 20);   // HW_REG_XCC_ID
    o[11] = wg;
  }
  };   // run_segment
  // at most two segments; the LDS buffers of the first are free once its last stage barrier has passed (the deferred tile lives
  // in registers)
  for (int seg = 0; seg < 2; ++seg) {
    const int t = 2 * pair + seg;
    const int a = imax(r0, seg * s_tile) - seg * s_tile, b = imin(r1, (seg + 1) * s_tile) - seg * s_tile;
    if (t < n_tiles && a < b) run_segment(t, a, b);
  }
}

static int g_corr6_debug = 0;
static int g_corr6_sdma = 1;
void set_corr6_sdma(int v) { g_corr6_sdma = v; }
static int g_corr6_skew = 0;   // stages moved from the two-segment piece to the others: -0.8 % / 0 / +0.7 % on three boxes at 2 (tools/experiments/try_corr6_skew.py): off
void set_corr6_skew(int v) { g_corr6_skew = v; }
static int g_corr6_cost_pro = 20000, g_corr6_cost_stage = 5700;      // cycles, tools/experiments/time_corr6.py
void set_corr6_debug(int v) { g_corr6_debug = v; }

int corr_volume_f16f6_launch(const unsigned char* q, const unsigned char* k, int HWq, int HWk, float temperature, float* vol,
                             hipStream_t s) {
  const int m32 = HWq & 31;
  int g = 32;
  while (g > 1 && (m32 % g) != 0) g >>= 1;                 // gcd(m32, 32) (m32 == 0 -> 32)
  int period = 32 / g;
  if (period > 4 || (g_corr6_debug & 4)) period = 1;       // too many classes (or ablation): unshifted, straddling stores
  const int n_q = cdiv(HWq + (period > 1 ? 31 : 0), 256);  // shifted classes start up to 31 queries early
  const int n_vb = cdiv(cdiv(HWk, period), 32);            // 32-row blocks of virtual rows per class
  // half-chunks per tile pair: the c that minimises  rounds over the 256 CUs x (prologues + stages per workgroup)
  const int s_tile = cdiv(n_vb, 2), n_tiles = n_q * period, n_pairs = cdiv(n_tiles, 2);
  int c_half = 2;
  {
    double best = -1;
    for (int c = 1; c <= 64; ++c) {
      const long long wgs = (long long)n_pairs * c;
      const double stages = 2.0 * s_tile / c;
      if (stages < 4) break;
      const double prologues = (c & 1) ? 1.0 + 1.0 / c : 1.0;      // one workgroup in c runs across the tile boundary
      const double cost = (double)((wgs + 255) / 256) * (g_corr6_cost_pro * prologues + g_corr6_cost_stage * stages);
      if (best < 0 || cost < best) {
        best = cost;
        c_half = c;
      }
    }
  }
  if (g_corr6_debug >> 12) c_half = g_corr6_debug >> 12;      // tuning override: half-chunks per tile pair
  dim3 grid(n_pairs * c_half);
  const float out_scale = 1.0f / (temperature * F6_S * F6_S);
  const int mm = period > 1 ? m32 : 0;
#define FGVC_C6(D) corr_volume_f16f6_kernel<8, D><<<grid, 512, 0, s>>>(q, k, HWq, HWk, out_scale, vol, s_tile, c_half, n_tiles, period, mm, g_corr6_skew)
  if (g_corr6_debug & 1024) {          // the store stream alone (results: zeros)
    FGVC_C6(1026);
    FGVC_CHECK_LAUNCH("fgvc_corr_volume_f16f6");
    return FGVC_OK;
  }
  switch (g_corr6_debug & 1019) {
    case 128: FGVC_C6(128); break;
    case 32 + 256: FGVC_C6(288); break;      // probe, no f16 MFMAs (results wrong)
    case 32 + 512: FGVC_C6(544); break;      // probe, no FP6 MFMAs (results wrong)
    case 32: FGVC_C6(32); break;
    case 33: FGVC_C6(33); break;
    case 34: FGVC_C6(34); break;
    case 0:
      if (g_corr6_sdma) FGVC_C6(0);
      else corr_volume_f16f6_kernel<8, 0, false><<<grid, 512, 0, s>>>(q, k, HWq, HWk, out_scale, vol, s_tile, c_half, n_tiles, period, mm, g_corr6_skew);
      break;
    case 1: FGVC_C6(1); break;
    case 2: FGVC_C6(2); break;
    case 3: FGVC_C6(3); break;
    case 8: FGVC_C6(8); break;
    case 9: FGVC_C6(9); break;
    case 16: FGVC_C6(16); break;
    case 64: FGVC_C6(64); break;
    case 80: FGVC_C6(80); break;
    default:
      set_error("fgvc_corr_volume_f16f6: corr6_debug = %d is not an instantiated variant", g_corr6_debug);
      return FGVC_ERR_INVALID_ARG;
  }
#undef FGVC_C6
  FGVC_CHECK_LAUNCH("fgvc_corr_volume_f16f6");
  return FGVC_OK;
}

// A linear sweep of 16-byte stores over `n` floats (the write stream of a kernel that has nothing else to do): the ceiling the volume
// kernel's store pattern is measured against in bench.py (`roofline.store_ceiling_gbps`).  2048 workgroups x 256 threads, grid-stride.
__global__ __launch_bounds__(256) void store_sweep_kernel(float* __restrict__ v, long long n4, int nt) {
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4* p = reinterpret_cast<f32x4*>(v);
  if (nt) {
    for (long long i = tid; i < n4; i += stride) __builtin_nontemporal_store(z, p + i);
  } else {
    for (long long i = tid; i < n4; i += stride) p[i] = z;
  }
}

int store_sweep_launch(float* v, long long n_floats, int nontemporal, hipStream_t s) {
  store_sweep_kernel<<<2048, 256, 0, s>>>(v, n_floats / 4, nontemporal);
  FGVC_CHECK_LAUNCH("fgvc_debug_store_sweep_f32");
  return FGVC_OK;
}

}  // namespace fgvc
