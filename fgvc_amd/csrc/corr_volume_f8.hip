// Dense correlation volume, parity-grade (every entry within 1e-3 logit of the f32 product; measured ~1e-4) at TWO bf16-MFMA
// times per tile instead of the three of bf16x3 -- the variant the north_star's ">= 50 % of the HBM roofline" is claimed on.
//
// Arithmetic.  Every feature value x (L2-normalised rows, |x| <= 1) is stored as three numbers (fgvc_split_f16f8):
//     h  = f16(256 x)            11 significand bits, exact products h_k * h_q in f32
//     h8 = e4m3(h)               h again, 4 significand bits
//     l8 = e4m3(256 (256 x - h)) the residual of h (|residual| <= 2^-11 |h|), 4 significand bits
// and   2^16 <k, q>  =  sum h_k h_q  +  sum h_k l_q  +  sum l_k h_q  +  sum l_k l_q.
// The first sum runs on v_mfma_f32_32x32x16_f16 (exact products, f32 accumulate).  The two cross sums are 2^-11 of the first,
// so 4 significand bits carry them to ~2^-15 of the result: they run on the block-scaled fp8 instruction
// v_mfma_scale_f32_32x32x64_f8f6f4 (h8 x l8 with a uniform E8M0 scale of 2^-8), which retires K = 64 in the time the f16 form
// needs for K = 32.  The last sum (2^-22) is dropped.  Per 32 x 32 tile and C = 256: 16 f16 MFMAs + 8 scaled fp8 MFMAs = 1024
// matrix-pipe cycles (bf16x3: 48 x 32 = 1536).  Error budget (tools/sim_f16f8.py, float64 reference): max 6e-5 logit on Gaussian
// features, 3.4e-4 on adversarially sparse rows; bf16x3 2e-5 / 1.3e-4; plain bf16 1e-2 / 6e-2.
// Operand layout of the scaled instruction (probed with exact integers, tools/micro/probe_fp8_scaled.hip): lane (r, h) holds
// row / column r, k = 32 h + b in byte b of its 8 operand registers; C/D as every 32x32 MFMA.
//
// Structure: as corr_volume_bf16_kernel (query fragments resident as B operands, 64-key stages through LDS by LDS-DMA, tiles
// stored straight from the accumulator), with a 1-KiB row per pixel = [h 512 B | h8 256 B | l8 256 B].
#include "common.hpp"

namespace fgvc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr float F8_S = 256.f;        // scale of h;  l8 carries another 2^8

// f32 rows -> [h (f16) C | h8 C | l8 C] bytes per pixel
__global__ __launch_bounds__(256) void split_f16f8_kernel(const float* __restrict__ feat, unsigned char* __restrict__ out,
                                                           long long n_vec4, int C) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= n_vec4) return;
  const long long e = g * 4;
  const long long pix = e / C;
  const int c = (int)(e - pix * C);
  const f32x4 x = *reinterpret_cast<const f32x4*>(feat + e);
  const float xs[4] = {x.x * F8_S, x.y * F8_S, x.z * F8_S, x.w * F8_S};
  _Float16 h[4];
  float l[4], hf[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = (_Float16)xs[i];
    hf[i] = (float)h[i];
    l[i] = (xs[i] - hf[i]) * F8_S;
  }
  int h8 = __builtin_amdgcn_cvt_pk_fp8_f32(hf[0], hf[1], 0, false);
  h8 = __builtin_amdgcn_cvt_pk_fp8_f32(hf[2], hf[3], h8, true);
  int l8 = __builtin_amdgcn_cvt_pk_fp8_f32(l[0], l[1], 0, false);
  l8 = __builtin_amdgcn_cvt_pk_fp8_f32(l[2], l[3], l8, true);
  unsigned char* row = out + pix * 4 * C;
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const f16x4 hv = {h[0], h[1], h[2], h[3]};
  *reinterpret_cast<f16x4*>(row + 2 * c) = hv;
  *reinterpret_cast<int*>(row + 2 * C + c) = h8;
  *reinterpret_cast<int*>(row + 3 * C + c) = l8;
}

int split_f16f8_launch(const float* feat, unsigned char* out, long long n_pixels, int C, hipStream_t s) {
  const long long n4 = n_pixels * C / 4;
  split_f16f8_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, s>>>(feat, out, n4, C);
  FGVC_CHECK_LAUNCH("fgvc_split_f16f8");
  return FGVC_OK;
}

// ------------------------------------------------------------------------------------------
// Row alignment.  vol is [HWk][HWq] f32, contiguous: row j starts at byte 4 j HWq, i.e. at float offset phi_j = (j HWq) mod 32
// inside a 128-byte line.  A wave's 128-byte row segment [q0, q0 + 32) is one whole line iff (phi_j + q0) mod 32 == 0; otherwise
// every store straddles two lines and HBM sees half-line writes (measured: 0.70 -> 0.54 ms for the same kernel with an aligned
// pitch).  With m = HWq mod 32 the phase has period p = 32 / gcd(m, 32) in j.  For p <= 4 (480p: m = 16, p = 2) the launch
// runs p classes of workgroups: class c takes the key rows j = p v + c (a "virtual" row v) and shifts its query tiles by
// o_c = (c m) mod 32 to the left, so that every store of every class writes whole lines.  Larger periods run unshifted.
// ------------------------------------------------------------------------------------------
template <int NW, int SUB, int DEBUG>   // DEBUG (profiling ablations, results wrong): 1 = no volume stores, 2 = no MFMAs
__global__ __launch_bounds__(NW * 64, 2) void corr_volume_f16f8_kernel(const unsigned char* __restrict__ q_sp,
                                                                      const unsigned char* __restrict__ k_sp, int HWq, int HWk,
                                                                      float out_scale, float* __restrict__ vol, int kchunk,
                                                                      int period, int m32) {
  constexpr int ROWB = 1024, LDB = ROWB + 16, ROWS = 32 * SUB, BUFB = ROWS * LDB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hi = lane >> 5;
  const int cls = blockIdx.z;                                   // row class: key rows j = period * v + cls
  const int shift = (cls * m32) & 31;                           // query tiles start `shift` queries early
  const int qw0 = blockIdx.x * (NW * 32) + wave * 32 - shift;   // wave-uniform: first query of this wave's tile
  const int q = qw0 + n;                                        // may be < 0 (first tile of a shifted class) or >= HWq
  const int n_v = (HWk - cls + period - 1) / period;            // virtual rows of this class

  // query fragments (B operands): h as 16 f16x8, h8 and l8 as 4 eight-register fp8 operands each
  f16x8 qh[16];
  i32x8 q8h[4], q8l[4];
  {
    const unsigned char* qp = q_sp + (size_t)imin(imax(q, 0), HWq - 1) * ROWB;
#pragma unroll
    for (int j = 0; j < 16; ++j) qh[j] = *reinterpret_cast<const f16x8*>(qp + 32 * j + 16 * hi);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      q8h[s] = *reinterpret_cast<const i32x8*>(qp + 512 + 64 * s + 32 * hi);
      q8l[s] = *reinterpret_cast<const i32x8*>(qp + 768 + 64 * s + 32 * hi);
    }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) asm volatile("" ::"v"(qh[j]));
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    asm volatile("" ::"v"(q8h[s]));
    asm volatile("" ::"v"(q8l[s]));
  }
  const int kb0 = blockIdx.y * kchunk;                // in units of 32 virtual rows
  const int kb1 = imin(kb0 + kchunk, cdiv(n_v, 32));
  auto stage_load = [&](int kb, int buf) {
#pragma unroll
    for (int i = 0; i < ROWS / NW; ++i) {
      const int row = wave * (ROWS / NW) + i;
      const int pix = imin((kb * 32 + row) * period + cls, HWk - 1);
      const unsigned char* src = k_sp + (size_t)pix * ROWB + 16 * lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)&smem[buf * BUFB + row * LDB], 16, 0, 0);
    }
  };
  stage_load(kb0, 0);
  __syncthreads();
  int buf = 0;
  const int lane_off = 4 * hi * period * HWq + q;                  // (row 4 hi of a tile, query q) relative to the tile's first row
  const bool wave_full = qw0 >= 0 && qw0 + 31 < HWq;               // wave-uniform: all 32 queries of this wave exist
  const size_t row_pitch = (size_t)period * HWq;                   // floats between consecutive virtual rows
  f32x16 acc;
  int n_counted = 0;                                               // stores of this stage issued by the known-count path; -1 = unknown

  auto store_tile = [&](int vrow0) {
    if constexpr (DEBUG & 1) {   // profiling ablation: no volume stores
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[r]));
      n_counted = -1;
      return;
    }
    // uniform (SGPR) row base + one loop-invariant per-lane offset: per-row 64-bit VGPR addresses would cost 32 registers
    float* tile = vol + (size_t)(vrow0 * period + cls) * HWq;
    if (wave_full && vrow0 + 32 <= n_v) {                  // wave-uniform
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* rowp = tile + (size_t)((r & 3) + 8 * (r >> 2)) * row_pitch;
        acc[r] *= out_scale;                               // in place: no second set of 16 registers
        asm volatile("" : "+v"(acc[r]));
        __builtin_nontemporal_store(acc[r], rowp + lane_off);
      }
      if (n_counted >= 0) n_counted += 16;
    } else {
      if (q >= 0 && q < HWq) {                             // ragged tile: same uniform row bases, per-lane predicate
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2);
          float* rowp = tile + (size_t)rr * row_pitch;
          acc[r] *= out_scale;
          asm volatile("" : "+v"(acc[r]));
          if (4 * hi < n_v - vrow0 - rr) __builtin_nontemporal_store(acc[r], rowp + lane_off);
        }
      }
      n_counted = -1;
    }
  };

  // A operands.  One step = K 64: 4 f16 fragments (double-buffered, read one step ahead) and the h8 / l8 operands (single-buffered:
  // read right after the scaled MFMAs of the previous step have issued, needed 4 f16 MFMAs = 128 pipe cycles later).  The first
  // step's operands of a tile are read BEFORE the store burst of the tile in front of it.
  f16x8 a16[2][4];
  i32x8 a8h, a8l;
  auto load16 = [&](const unsigned char* ka, int s, int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a16[slot][i] = *reinterpret_cast<const f16x8*>(ka + 32 * (4 * s + i) + 16 * hi);
  };
  auto load8 = [&](const unsigned char* ka, int s) {
    a8h = *reinterpret_cast<const i32x8*>(ka + 512 + 64 * s + 32 * hi);
    a8l = *reinterpret_cast<const i32x8*>(ka + 768 + 64 * s + 32 * hi);
  };

  for (int kb = kb0; kb < kb1; kb += SUB) {
    const bool more = kb + SUB < kb1;
    if (more) stage_load(kb + SUB, buf ^ 1);
    n_counted = 0;
    {
      const unsigned char* ka0 = &smem[buf * BUFB + n * LDB];
      load16(ka0, 0, 0);
      load8(ka0, 0);
    }
#pragma unroll
    for (int sb = 0; sb < SUB; ++sb) {
      if (kb + sb >= kb1) break;                       // wave-uniform (ragged tail of the chunk)
      const unsigned char* ka = &smem[buf * BUFB + (sb * 32 + n) * LDB];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < 4) load16(ka, s + 1, (s + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(DEBUG & 2)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[s & 1][i], qh[4 * s + i], acc, 0, 0, 0);
          // E8M0 scales: A 2^-8 (0x77), B 2^0 (0x7f): the cross sums enter at 2^-8 of the f16 sum's scale
          acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8h, q8l[s], acc, 0, 0, 0, 0x77777777, 0, 0x7f7f7f7f);
          acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8l, q8h[s], acc, 0, 0, 0, 0x77777777, 0, 0x7f7f7f7f);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(a16[s & 1][i]));
          asm volatile("" ::"v"(a8h));
          asm volatile("" ::"v"(a8l));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < 4) load8(ka, s + 1);
      }
      if (sb + 1 < SUB && kb + sb + 1 < kb1) {         // the next tile's first operands, ahead of the stores
        load16(ka + 32 * LDB, 0, 0);
        load8(ka + 32 * LDB, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      store_tile((kb + sb) * 32);
    }
    // the counted wait retires this stage's DMA (issued before every store counted here) and everything older, but leaves the
    // stores of this stage in flight across the barrier
    if (more) {
      if (n_counted == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else if (n_counted == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();   // not __syncthreads(): must not wait for the volume stores above
    buf ^= 1;
  }
}

// ------------------------------------------------------------------------------------------
// v2: the same product on the 16x16 MFMA shapes (v_mfma_f32_16x16x32_f16 + v_mfma_scale_f32_16x16x128_f8f6f4: the chip holds a
// higher clock on them, MI355X_MICROARCH.md "DVFS give-back" item 7; measured here -6 % with the volume stores on), the two waves
// of a SIMD anti-phased, and A operands in two fixed register sets.
//   * a wave's 32 x 32 tile = 2 x 2 tiles of 16 x 16 (key half kt, query half qt), 4 independent accumulators of 4 registers;
//     lane (r = l & 15, g = l >> 4) holds A[key 16 kt + r][k = 8 g + j] (f16, K-32 step) or [k = 32 g + b] (fp8, K-128 block) and
//     D[key 16 kt + 4 g + i][query 16 qt + r] in register i (all probed with exact integers: tools/micro/probe_16x16.hip).
//     1056-byte LDS rows make the f16 fragment reads conflict-free (brute-forced over the ds_read_b128 lane groups).
//   * per K-128 block: F part = 16 f16 MFMAs (4 K-32 steps x 2 x 2 tiles) from register set F (8 fragments), P part = 8 scaled
//     fp8 MFMAs from set P (2 x (h8, l8) operands).  Set P of a block is read at the START of its F part and set F of the next block
//     at the start of the P part: every read has 256 pipe cycles to land, and each set is re-read only after its last reader issued.
//   * stores: v_permlane16_swap of the two query halves turns a register pair into two row pieces of 128 B each (rows 16 kt + i and
//     16 kt + 8 + i | 16 kt + 4 + i and 16 kt + 12 + i): the same 2 x 128-byte store shape as the 32 x 32 accumulator.
//   * stagger (corr8_debug bit 8 switches it off): waves 4-7 (the SIMD partners of waves 0-3) hold a finished tile in its
//     accumulators and store it at the start of their NEXT tile (across the stage barrier too), so that a stage opens with one wave
//     of each SIMD multiplying and the other in the store queue: -1.5 % (round-robin timing, tools/experiments/try_f16f8.py).
//   * the natural fp8 element order (bytes [32 g, 32 g + 32) of a K-128 block) costs a 2-way bank conflict on the P reads; the
//     conflict-free order (two 16-byte pieces 64 bytes apart) needs a shufflevector of two reads, which makes hipcc wait for
//     BOTH reads at once (s_waitcnt lgkmcnt(0) right behind them) instead of counting: measured slower.
// ------------------------------------------------------------------------------------------
template <int NW, int DEBUG>   // DEBUG: 1 = no volume stores, 2 = no MFMAs (results wrong), 8 = no wave stagger (results right),
                               // 32 = s_memtime probe of one workgroup, written over the first floats of vol (tools/experiments/time_corr8.py)
__global__ __launch_bounds__(NW * 64, 2) void corr_volume_f16f8_v2_kernel(const unsigned char* __restrict__ q_sp,
                                                                         const unsigned char* __restrict__ k_sp, int HWq, int HWk,
                                                                         float out_scale, float* __restrict__ vol, int kchunk,
                                                                         int period, int m32) {
  constexpr int SUB = 2, ROWB = 1024, LDB = ROWB + 32, ROWS = 32 * SUB, BUFB = ROWS * LDB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int cls = blockIdx.z;
  const int shift = (cls * m32) & 31;
  const int qw0 = blockIdx.x * (NW * 32) + wave * 32 - shift;   // wave-uniform: first query of this wave's tile
  const int n_v = (HWk - cls + period - 1) / period;            // virtual rows of this class

  const bool probe = (DEBUG & 32) && blockIdx.x == 50 && blockIdx.y == 1 && blockIdx.z == 0;
  long long p_start = 0, p_pro = 0, p_comp = 0, p_store = 0, p_sync = 0, c0 = 0;
  int p_stages = 0;
  if (probe) p_start = __builtin_amdgcn_s_memtime();
  // query fragments (B operands) of the two query halves
  f16x8 bq16[2][8];
  i32x8 bq8h[2][2], bq8l[2][2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const unsigned char* qp = q_sp + (size_t)imin(imax(qw0 + 16 * qt + r, 0), HWq - 1) * ROWB + 16 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) bq16[qt][t] = *reinterpret_cast<const f16x8*>(qp + 64 * t);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bq8h[qt][u] = *reinterpret_cast<const i32x8*>(qp + 512 + 128 * u + 16 * g);     // qp holds + 16 g: bytes [32 g, 32 g + 32)
      bq8l[qt][u] = *reinterpret_cast<const i32x8*>(qp + 768 + 128 * u + 16 * g);
    }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
    for (int t = 0; t < 8; ++t) asm volatile("" ::"v"(bq16[qt][t]));
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      asm volatile("" ::"v"(bq8h[qt][u]));
      asm volatile("" ::"v"(bq8l[qt][u]));
    }
  }
  const int kb0 = blockIdx.y * kchunk;                // in units of 32 virtual rows
  const int kb1 = imin(kb0 + kchunk, cdiv(n_v, 32));
  auto stage_load = [&](int kb, int buf) {
#pragma unroll
    for (int i = 0; i < ROWS / NW; ++i) {
      const int row = wave * (ROWS / NW) + i;
      const int pix = imin((kb * 32 + row) * period + cls, HWk - 1);
      const unsigned char* src = k_sp + (size_t)pix * ROWB + 16 * lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)&smem[buf * BUFB + row * LDB], 16, 0, 0);
    }
  };
  // Staging the NEXT stage costs the issuing wave ~125 cycles per LDS-DMA instruction (64 of them per stage).  The older wave of a
  // SIMD wins the issue arbitration, finishes its multiplies first and then sits ~2000 cycles at the stage barrier (s_memtime
  // probe: tools/experiments/time_corr8.py), while its partner is the critical path.  So waves 0-3 stage ALL 64 rows, two per multiply
  // part, and waves 4-7 none (as a burst behind the barrier, or spread over every wave, the matrix pipe idled ~1000 cycles per stage).
  const bool stager = wave < NW / 2;
  auto stage_row = [&](int kb, int buf, int i) {
    const int row = wave * (2 * ROWS / NW) + i;
    const int pix = imin((kb * 32 + row) * period + cls, HWk - 1);
    const unsigned char* src = k_sp + (size_t)pix * ROWB + 16 * lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)&smem[buf * BUFB + row * LDB], 16, 0, 0);
  };
  static_assert(2 * ROWS / NW == 16, "two DMA rows per multiply part of a staging wave: 2 tiles x 4 parts");
  stage_load(kb0, 0);
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
    // x 32 queries, y = rows 16 kt + 4 + i (+ 8).  Inline assembly: with the builtin hipcc 7.2 stored the FIRST result twice
    // in this kernel, and it pads no hazard wait states inside asm -- so all 16 products first, ONE s_nop, the 8 swaps back to
    // back (each reads registers written at least 8 instructions earlier), one s_nop, then the stores.
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

  // A operand register sets
  f16x8 F[2][4];            // [kt][K-32 step of the block]
  i32x8 Ph[2], Pl[2];       // [kt]: h8 / l8 operands of the block
  auto load_F = [&](const unsigned char* ka, int u) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) F[kt][tt] = *reinterpret_cast<const f16x8*>(ka + 16 * kt * LDB + 64 * (4 * u + tt));
  };
  auto load_P = [&](const unsigned char* ka, int u) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const unsigned char* p = ka + 16 * kt * LDB + 512 + 128 * u + 16 * g;     // ka already holds + 16 g: bytes [32 g, 32 g + 32)
      Ph[kt] = *reinterpret_cast<const i32x8*>(p);
      Pl[kt] = *reinterpret_cast<const i32x8*>(p + 256);
    }
  };

  int buf = 0;
  for (int kb = kb0; kb < kb1; kb += SUB) {
    const bool more = kb + SUB < kb1;
    n_counted = 0;
    load_F(&smem[buf * BUFB + r * LDB + 16 * g], 0);
    if (probe) c0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int sb = 0; sb < SUB; ++sb) {
      if (kb + sb >= kb1) break;                       // wave-uniform (ragged tail of the chunk)
      const unsigned char* ka = &smem[buf * BUFB + (sb * 32 + r) * LDB + 16 * g];
      const bool next_here = sb + 1 < SUB && kb + sb + 1 < kb1;
      __builtin_amdgcn_sched_barrier(0);
      if (defer && pend_v >= 0) store_tile(pend_v);    // waves 4-7: the previous tile leaves under the partner's multiplies
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) acc[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        load_P(ka, u);                                 // lands during the F part
        if (more && stager) {
          stage_row(kb + SUB, buf ^ 1, 8 * sb + 4 * u);
          stage_row(kb + SUB, buf ^ 1, 8 * sb + 4 * u + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(DEBUG & 2)) {
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
              for (int qt = 0; qt < 2; ++qt)
                acc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(F[kt][tt], bq16[qt][4 * u + tt], acc[kt][qt], 0, 0, 0);
        } else {
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) asm volatile("" ::"v"(F[kt][tt]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (u == 0) load_F(ka, 1);                     // lands during the P part
        else if (next_here) load_F(ka + 32 * LDB, 0);
        if (more && stager) {
          stage_row(kb + SUB, buf ^ 1, 8 * sb + 4 * u + 2);
          stage_row(kb + SUB, buf ^ 1, 8 * sb + 4 * u + 3);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(DEBUG & 2)) {
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
              // E8M0 scales: A 2^-8 (0x77), B 2^0 (0x7f): the cross sums enter at 2^-8 of the f16 sum's scale
              acc[kt][qt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(Ph[kt], bq8l[qt][u], acc[kt][qt], 0, 0, 0, 0x77777777, 0, 0x7f7f7f7f);
              acc[kt][qt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(Pl[kt], bq8h[qt][u], acc[kt][qt], 0, 0, 0, 0x77777777, 0, 0x7f7f7f7f);
            }
        } else {
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            asm volatile("" ::"v"(Ph[kt]));
            asm volatile("" ::"v"(Pl[kt]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const int vrow0 = (kb + sb) * 32;
      if (probe) { const long long c1 = __builtin_amdgcn_s_memtime(); p_comp += c1 - c0; c0 = c1; }
      if (defer) pend_v = vrow0;
      else store_tile(vrow0);
      if (probe) { const long long c1 = __builtin_amdgcn_s_memtime(); p_store += c1 - c0; c0 = c1; }
    }
    if (more && stager) {
      // the last DMA of the stage was issued inside the second tile: only that tile's own store burst (16, if it was a whole
      // tile stored right away) is younger and may stay in flight across the barrier.  Waves 4-7 issued no load: no wait.
      if (!defer && n_counted == 32 && kb + 1 < kb1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();
    if (probe) { p_sync += __builtin_amdgcn_s_memtime() - c0; ++p_stages; }
    buf ^= 1;
  }
  if (defer && pend_v >= 0) store_tile(pend_v);
  if (probe && lane == 0) {
    long long* o = reinterpret_cast<long long*>(vol) + 8 * wave;
    o[0] = p_pro; o[1] = p_comp; o[2] = p_store; o[3] = p_sync; o[4] = __builtin_amdgcn_s_memtime() - p_start; o[5] = p_stages;
  }
}

static int g_corr8_debug = 0;
void set_corr8_debug(int v) { g_corr8_debug = v; }

int corr_volume_f16f8_launch(const unsigned char* q, const unsigned char* k, int HWq, int HWk, float temperature, float* vol,
                             hipStream_t s) {
  const int m32 = HWq & 31;
  int g = 32;
  while (g > 1 && (m32 % g) != 0) g >>= 1;                 // gcd(m32, 32) (m32 == 0 -> 32)
  int period = 32 / g;
  if (period > 4 || (g_corr8_debug & 4)) period = 1;       // too many classes (or ablation): unshifted, straddling stores
  const int n_q = cdiv(HWq + (period > 1 ? 31 : 0), 256);  // shifted classes start up to 31 queries early
  const int n_vb = cdiv(cdiv(HWk, period), 32);            // 32-row blocks of virtual rows per class
  // key blocks per workgroup.  One workgroup per CU (135 KB of LDS), ~20 000 cycles of prologue each (query fragments + first
  // stage) and ~5 800 per 64-key stage (s_memtime probe, tools/experiments/time_corr8.py): take the number of key chunks that minimises
  //   rounds over the 256 CUs x (prologue + stages per chunk)
  // (480p: 5 chunks = 1020 workgroups = 3.98 rounds; 720p: 9 chunks = 2025 workgroups = 7.9 rounds -- 4 chunks would leave half of
  // the fourth round empty).
  int kchunk = n_vb + (n_vb & 1);
  {
    long long best = -1;
    for (int c = 1; c <= 32; ++c) {
      int kc = cdiv(n_vb, c);
      kc += kc & 1;                                         // whole 64-key stages
      const long long wgs = (long long)n_q * period * cdiv(n_vb, kc);
      const long long cost = ((wgs + 255) / 256) * (20000ll + 5800ll * (kc / 2));
      if (best < 0 || cost < best) {
        best = cost;
        kchunk = kc;
      }
    }
  }
  if (g_corr8_debug >> 8) kchunk = g_corr8_debug >> 8;
  dim3 grid(n_q, cdiv(n_vb, kchunk), period);
  const float out_scale = 1.0f / (temperature * F8_S * F8_S);
  const int mm = period > 1 ? m32 : 0;
#define FGVC_C8(D) corr_volume_f16f8_kernel<8, 2, D><<<grid, 512, 0, s>>>(q, k, HWq, HWk, out_scale, vol, kchunk, period, mm)
#define FGVC_C8V2(D) corr_volume_f16f8_v2_kernel<8, D><<<grid, 512, 0, s>>>(q, k, HWq, HWk, out_scale, vol, kchunk, period, mm)
  if (g_corr8_debug & 16) {          // the 32x32-shape kernel (v1)
    switch (g_corr8_debug & 3) {     // bit 2 (value 4) is the launch-side "no row classes" switch
      case 0: FGVC_C8(0); break;
      case 1: FGVC_C8(1); break;
      case 2: FGVC_C8(2); break;
      default: FGVC_C8(3); break;
    }
  } else {
    switch (g_corr8_debug & 43) {
      case 32: FGVC_C8V2(32); break;
      case 33: FGVC_C8V2(33); break;
      case 0: FGVC_C8V2(0); break;
      case 1: FGVC_C8V2(1); break;
      case 2: FGVC_C8V2(2); break;
      case 3: FGVC_C8V2(3); break;
      case 8: FGVC_C8V2(8); break;
      case 9: FGVC_C8V2(9); break;
      case 10: FGVC_C8V2(10); break;
      default: FGVC_C8V2(11); break;
    }
  }
#undef FGVC_C8
#undef FGVC_C8V2
  FGVC_CHECK_LAUNCH("fgvc_corr_volume_f16f8");
  return FGVC_OK;
}

}  // namespace fgvc
