// fgvc_pair_topk_f16f6: the windowed correlation + top-k of fgvc_pair_topk_f16x3 at HALF the matrix work and 0.72 of the selection
// work -- the pair kernel in the arithmetic the encoder (f16 + fp8) and the dense volume (f16 + FP6) already use.  Included by
// pair_topk_v5.hip (same translation unit: the bounded spins, the timeout flag and the LDS-DMA helper are shared).
//
// What round 3's profile said about the f16x3 kernel (27 pairs of a 480p clip, 1.31-1.46 ms): the consumers' chain of 48 f16 MFMAs
// runs at 42 cycles per MFMA (2 011 per tile), the selector needs ~1 450, the ring of key blocks alone 0.86 ms.  Here:
//   * Arithmetic (1.5 pipe units instead of 3).  x -> h = f16(256 x), l = 256 (256 x - h) as in fgvc_split_f16f6; the cross sums
//     h_k l_q + l_k h_q are 2^-11 of the main sum, so their operands go to FP6 (e2m3, one E8M0 scale per 32 channels):
//         2^16 <k, q> = sum h_k h_q  +  2^-8 (sum h6_k l6_q + sum l6_k h6_q)
//     16 v_mfma_f32_32x32x16_f16 + 8 v_mfma_scale_f32_32x32x64_f8f6f4 per 32 x 32 tile = 768 pipe cycles (f16x3: 1 536).  Error of
//     a score against float64: that of fgvc_corr_volume_f16f6 (~6e-5 logit at tau 0.07 on Gaussian rows; bar 1e-3).
//     Probed first (tools/micro/probe_fp6_32x32.hip, profiles/r04_probe_fp6_32x32.log): with FP6 operands the 32 x 32 x 64 form takes
//     a lane's scale for that lane's OWN 32 elements (with fp8 operands it does not: conv_split.hip), element e sits in bits
//     [6 e, 6 e + 6), and v_cvt_scalef32_pk32_fp6_f16 computes e2m3(x / scale), round to nearest even, saturating.
//   * Row format (fgvc_split_f16f6p, 1 KiB per pixel like every other bank format) laid out for THIS kernel's lanes: a lane (n, hi)
//     of the 32 x 32 shapes holds channels 16 j + 8 hi + i of f16 fragment j; four fragments 4 v .. 4 v + 3 are the lane's 32
//     elements of 64-channel group v (e = 8 m + i <-> channel 64 v + 16 m + 8 hi + i), and THAT set is the scale block.  So the
//     query's h6 operand needs no registers of its own: one v_cvt_scalef32_pk32_fp6_f16 per group turns the 16 registers of
//     resident f16 fragments into the 6-register FP6 operand right before its MFMA (the query's l6 stays resident: 24 registers).
//     Every piece of a row is addressed as row + 16 hi + constant: one address register per tile.
//   * Selection without a payload.  A score leaves the consumer as (bits(acc + 2^19 + 2^17) << 10) | tag: the biased accumulator
//     lies in one binade, so its 22 low mantissa bits are the score in 2^-20 fixed point (1.4e-5 logit at tau 0.07: a third of
//     the arithmetic's own error), and tag = (63 - position in the block list) << 4 | (15 - register) names the key pixel.  The
//     running list is 10 keys; compare-exchange = v_max_u32 + v_min_u32.  212 vector operations per tile (293).
//   * The block list in plain row-major order: the ring alone runs 21 % faster that way (neighbouring workgroups walk the same key
//     rows at the same time; profiles/r04_pair_ring.log), which the f16x3 kernel could not use (its consumers were the pole).
//     Later in round 4: rows rotated so that every tile visits key row r at step r mod 10 ("aligned walks"), and the workgroups of an
//     XCD own a contiguous range of (run, tile) work in column strips -- see the work-order comment in the kernel.
// Limits: C = 256, k <= 10, normalised rows, at most 64 key blocks per 8 x 16 query tile (a radius-15 disc has 56) -- everything else
// stays on fgvc_pair_topk_f16x3 / fgvc_pair_topk_f32.  Same fail-closed protocol as the f16x3 kernel (bounded spins, poison lists).
#pragma once

namespace fgvc {

typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x6 __attribute__((ext_vector_type(6)));

constexpr int P6_ROWB = 1024, P6_H6M = 512, P6_H6T = 640, P6_L6M = 704, P6_L6T = 832, P6_SC = 896, P6_END = 928;
constexpr float P6_S = 256.f;
constexpr float V7_BIAS = 655360.f;              // 2^19 + 2^17: acc + bias in [2^19 + 2^16, 2^19 + 3 2^16] for |cos| <= 1 (room for 2x)
constexpr int V7_MAX_BLOCKS = 64;                // 6 bits of a key name the block

// E8M0 exponent s with max / 2^s <= 7.5 (the largest e2m3 value); an all-zero block gets a small harmless scale
__device__ __forceinline__ int p6_scale_exp(float m) {
  if (!(m > 0.f)) return -40;
  int e;
  const float f = frexpf(m * (1.0f / 7.5f), &e);
  int s = (f > 0.5f) ? e : e - 1;
  if (m * exp2f((float)-s) > 7.5f) ++s;
  return imax(s, -40);
}
__device__ __forceinline__ unsigned p6_code(float y) {          // |y| <= 7.5 -> e2m3 code, round to nearest even
  const float a = fabsf(y);
  const float inv_step = a < 2.f ? 8.f : (a < 4.f ? 4.f : 2.f);
  const float r = fminf(__builtin_rintf(a * inv_step) / inv_step, 7.5f);
  const float c = r < 2.f ? 8.f * r : (r < 4.f ? 8.f + 4.f * r : 16.f + 2.f * r);
  return (unsigned)c | (y < 0.f ? 32u : 0u);
}

// f32 rows [pixel][256] (L2-normalised) -> fgvc_split_f16f6p rows; one thread per (pixel, group v, lane half hi)
// (rowb = 2048, fgvc_split_f16f6x: the second KiB of a row is the pixel's 256 f32 channels themselves -- the exact rows the refining merge reads)
__global__ __launch_bounds__(256) void split_f16f6p_kernel(const float* __restrict__ feat, unsigned char* __restrict__ out, long long n_blocks, int rowb) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_blocks) return;
  const long long pix = t >> 3;
  const int blk = (int)(t & 7), v = blk >> 1, hi = blk & 1;
  const float* src = feat + pix * 256 + 64 * v + 8 * hi;
  unsigned char* row = out + pix * rowb;
  float hf[32], lf[32];
  float mh = 0.f, ml = 0.f;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
    f16x8v hv;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(src + 16 * m + 4 * q);
      if (rowb > P6_ROWB) *reinterpret_cast<f32x4*>(row + P6_ROWB + 4 * (64 * v + 8 * hi + 16 * m + 4 * q)) = x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = 8 * m + 4 * q + j;
        const float xs = x[j] * P6_S;
        const _Float16 h = (_Float16)xs;
        hv[4 * q + j] = h;
        hf[e] = (float)h;
        lf[e] = (xs - (float)h) * P6_S;
        mh = fmaxf(mh, fabsf(hf[e]));
        ml = fmaxf(ml, fabsf(lf[e]));
      }
    }
    *reinterpret_cast<f16x8v*>(row + 2 * (64 * v + 16 * m + 8 * hi)) = hv;
  }
  const int sh = p6_scale_exp(mh), sl = p6_scale_exp(ml);
  const float ih = exp2f((float)-sh), il = exp2f((float)-sl);
  unsigned wh[6] = {0, 0, 0, 0, 0, 0}, wl[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 32; ++e) {
    const unsigned ch = p6_code(hf[e] * ih), cl = p6_code(lf[e] * il);
    const int bit = 6 * e, word = bit >> 5, s = bit & 31;
    wh[word] |= ch << s;
    wl[word] |= cl << s;
    if (s > 26) {
      wh[word + 1] |= ch >> (32 - s);
      wl[word + 1] |= cl >> (32 - s);
    }
  }
  *reinterpret_cast<i32x4v*>(row + P6_H6M + 32 * v + 16 * hi) = i32x4v{(int)wh[0], (int)wh[1], (int)wh[2], (int)wh[3]};
  *reinterpret_cast<i32x2v*>(row + P6_H6T + 32 * (v >> 1) + 16 * hi + 8 * (v & 1)) = i32x2v{(int)wh[4], (int)wh[5]};
  *reinterpret_cast<i32x4v*>(row + P6_L6M + 32 * v + 16 * hi) = i32x4v{(int)wl[0], (int)wl[1], (int)wl[2], (int)wl[3]};
  *reinterpret_cast<i32x2v*>(row + P6_L6T + 32 * (v >> 1) + 16 * hi + 8 * (v & 1)) = i32x2v{(int)wl[4], (int)wl[5]};
  row[P6_SC + 16 * hi + v] = (unsigned char)(sh + 127 - 4);             // 2^(s - 4) each: a product of two FP6 operands enters at 2^-8
  row[P6_SC + 16 * hi + 4 + v] = (unsigned char)(sl + 127 - 4);
  if (v == 0) *reinterpret_cast<i32x2v*>(row + P6_SC + 16 * hi + 8) = i32x2v{0, 0};
  if (blk < 6) *reinterpret_cast<i32x4v*>(row + P6_END + 16 * blk) = i32x4v{0, 0, 0, 0};
}

int split_f16f6p_launch(const float* feat, unsigned char* out, long long n_pixels, int rowb, hipStream_t s) {
  const long long nb = n_pixels * 8;
  split_f16f6p_kernel<<<(unsigned)((nb + 255) / 256), 256, 0, s>>>(feat, out, nb, rowb);
  FGVC_CHECK_LAUNCH("fgvc_split_f16f6p");
  return FGVC_OK;
}

__device__ __forceinline__ i32x6 v7_cat6(const i32x4v& a, const i32x2v& b) { return i32x6{a[0], a[1], a[2], a[3], b[0], b[1]}; }
#define V7_CAT6(A, B) v7_cat6(A, B)
#define V7_SUB8(Q, M) __builtin_shufflevector(Q, Q, 8 * (M), 8 * (M) + 1, 8 * (M) + 2, 8 * (M) + 3, 8 * (M) + 4, 8 * (M) + 5, 8 * (M) + 6, 8 * (M) + 7)

// waves 0-3 consumers, 4-7 selectors, 8-11 producers (one of each per SIMD), as in pair_topk_kernel_v6
template <int K, bool PROBE>
__global__ __launch_bounds__(768, 1) void pair_topk_kernel_v7(PairParamsB p) {
  constexpr int LDB = P6_END + 16;               // LDS row of one pixel: the 928 bytes that carry something + 16 -> 236 dwords = 44 mod 64:
                                                 // conflict-free b128 reads by lanes (n, hi); the zero tail of a row is neither copied nor stored
  constexpr int BUFB = 32 * LDB;
  constexpr int NSLOT = 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * BUFB];
  __shared__ __attribute__((aligned(16))) unsigned char q6h_lds[4][6144];   // per consumer: its queries' h6 pieces, [v][lane] 16 B mains, then [v][lane] 8 B tails
  __shared__ uint32_t blist[V7_MAX_BLOCKS];      // by | bx << 12 | (query blocks that reach it) << 24
  __shared__ __attribute__((aligned(16))) unsigned int hand[4][16 * 64];   // consumer -> selector: a tile's keys, register-major per 16-byte piece
  __shared__ int hand_full[4], hand_free[4];
  __shared__ int blist_n;
  __shared__ int filled[NSLOT], done[NSLOT];
  __shared__ int wg_dead;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = wave & 3, role = wave >> 2;     // role 0 = consumer of query block qb, 1 = its selector, 2 = producer of pixel row qb
  const int n = lane & 31, hi = lane >> 5;

  int g_start = blockIdx.y, g_count = 1;
  if (p.groups) {
    const int2 gr = p.groups[blockIdx.y];
    g_start = gr.x;
    g_count = gr.y;
  }
  int tile = xcd_remap(blockIdx.x, p.n_ty * p.n_tx);
  if (!(p.debug & 512)) {                       // (512: the order of the first build -- an eighth of the tiles of every run per XCD, raster)
    // ---- work order.  The hardware deals workgroups to the eight XCDs by their linear index; every XCD has its own L2.
    // The runs of one length (a "class": equal work per workgroup; the runs come longest first) own a stretch of linear indices;
    // the workgroups an XCD gets from that stretch take a CONTIGUOUS range of the class's (run, tile) sequence, tiles in column
    // strips two tiles wide -- instead of an eighth of the tiles of every run (whose key windows overlap those of seven other L2s).
    // With the aligned walks below: L2 fills 4.62 -> 3.54 GB per 27-pair launch at 480p (one pair per workgroup: 2.50 GB, but 3 %
    // slower: the query prologue per pair); launch time 1.159 -> 1.150 ms there, 10.05 -> 9.77 ms for the 363 pairs of a 64-frame
    // 256 x 256 video, 13.59 -> 12.65 ms for the 123 pairs of 24 frames of 720p (tools/experiments/order_pair_v7.py).
    const int ntile = p.n_ty * p.n_tx;
    const int L = blockIdx.x + gridDim.x * blockIdx.y;
    int g_lo = 0, g_hi = (int)gridDim.y - 1;
    if (p.groups) {
      const int gi = blockIdx.y;
      g_lo = gi; g_hi = gi;
      for (int base = 0; base < (int)gridDim.y; base += 64) {          // maximal stretch of runs of this length around gi
        const int i = base + lane;
        const unsigned long long eq = __ballot(i < (int)gridDim.y && p.groups[imin(i, (int)gridDim.y - 1)].y == g_count);
        const unsigned long long ne = ~eq;
        if (gi >= base && gi < base + 64) {
          const int o = gi - base;
          const unsigned long long below = ne & ((1ull << o) - 1), above = o == 63 ? 0ull : (ne >> (o + 1));
          g_lo = below ? base + 64 - __builtin_clzll(below) : (base == 0 ? 0 : -1);
          g_hi = above ? gi + __builtin_ctzll(above) : -1;
        }
      }
      // (stretches that cross a 64-run chunk are cut at the chunk: still a bijection, only a shorter class)
      if (g_lo < 0) g_lo = (gi / 64) * 64;
      if (g_hi < 0) g_hi = imin((gi / 64) * 64 + 63, (int)gridDim.y - 1);
    }
    const int a_ = g_lo * ntile, b_ = (g_hi + 1) * ntile, x_ = L & 7;
    auto below_x = [&](int m) { return (m >> 3) * x_ + imin(m & 7, x_); };            // integers in [0, m) with residue < x_
    const int L0 = a_ + ((x_ - (a_ & 7) + 8) & 7);                                   // first index of this XCD in the stretch
    const int item = below_x(b_) - below_x(a_) + ((L - L0) >> 3);
    const int gsel = g_lo + item / ntile;
    const int t2 = item - (gsel - g_lo) * ntile;
    if (p.groups) {
      const int2 gr = p.groups[gsel];
      g_start = gr.x;
      g_count = gr.y;
    } else {
      g_start = gsel;
    }
    const int full = (p.n_tx >> 1) * 2 * p.n_ty;
    if (t2 < full) {
      const int strip = t2 / (2 * p.n_ty), r = t2 - strip * 2 * p.n_ty;
      tile = (r >> 1) * p.n_tx + 2 * strip + (r & 1);
    } else {
      tile = (t2 - full) * p.n_tx + p.n_tx - 1;
    }
  }
  const int4 pr = p.pairs[g_start];
  const int qf = pr.x;
  const bool masked = (pr.z & FGVC_PAIR_MASKED) != 0;
  const int reach_y = masked ? p.reach_y : FGVC_NO_LIMIT;
  const int reach_x = masked ? p.reach_x : FGVC_NO_LIMIT;

  const int ty = tile / p.n_tx, tx = tile - ty * p.n_tx;
  ReachTest reach;
  reach.r2max = masked ? p.r2max : FGVC_NO_LIMIT;
  reach.ry = masked ? p.ry : FGVC_NO_LIMIT;
  reach.rx = masked ? p.rx : FGVC_NO_LIMIT;
  const int TY0 = ty * (2 * QBH), TX0 = tx * (2 * QBW);
  const int QY0 = TY0 + (qb & 1) * QBH, QX0 = TX0 + (qb >> 1) * QBW;
  const int qy = QY0 + (n >> 3), qx = QX0 + (n & 7);

  // ---- prologue 1: the query rows of the four blocks through the ring (coalesced 1 KiB rows by LDS-DMA)
  if (role < 2 && lane < P6_END / 16) {
    const unsigned char* qbase = reinterpret_cast<const unsigned char*>(p.q_hl) + (size_t)qf * p.Hq * p.Wq * p.rowb + 16 * lane;
    i32x4v qr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = role * 16 + i;
      const int y = imin(QY0 + (r >> 3), p.Hq - 1), x = imin(QX0 + (r & 7), p.Wq - 1);
      qr[i] = *reinterpret_cast<const i32x4v*>(qbase + ((size_t)y * p.Wq + x) * p.rowb);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<i32x4v*>(&smem[qb * BUFB + (role * 16 + i) * LDB + 16 * lane]) = qr[i];
  }
  // ---- prologue 2 (overlaps the DMA): the key blocks this super-tile visits, row by row (debug & 8192: alternating from both ends)
  if (wave == 0) {
    const int by_lo = imax(0, TY0 - imin(reach_y, TY0)) / QBH;
    const int by_hi = imin(p.Hk - 1, TY0 + 2 * QBH - 1 + imin(reach_y, p.Hk)) / QBH;
    const int bxl = imax(0, TX0 - imin(reach_x, TX0)) / QBW;
    const int bxh = imin(p.Wk - 1, TX0 + 2 * QBW - 1 + imin(reach_x, p.Wk)) / QBW;
    const int nbx = bxh - bxl + 1;
    const int nall = (by_hi - by_lo + 1) * nbx;
    // Block rows in the order of (row mod M), M = the rows of a full window: workgroups that start together then read the same key
    // rows at the same time whatever their own tile row ("aligned walks"; debug & 1024: every tile from its own top row)
    const int M_ = (2 * QBH - 1 + imin(reach_y, p.Hk)) / QBH + (imin(reach_y, p.Hk) + QBH - 1) / QBH + 1;
    int rot_s = ((by_lo + M_ - 1) / M_) * M_;
    if (rot_s > by_hi || (p.debug & 1024)) rot_s = by_lo;
    auto reach_bits = [&](int c) -> uint32_t {
      const int j_ = c / nbx, bx = bxl + c % nbx;
      const int by = rot_s + j_ <= by_hi ? rot_s + j_ : by_lo + (j_ - (by_hi - rot_s + 1));
      uint32_t m = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b)
        m |= (uint32_t)reach(TY0 + (b & 1) * QBH, TX0 + (b >> 1) * QBW, by * QBH, bx * QBW) << b;
      return m ? ((uint32_t)by | ((uint32_t)bx << 12) | (m << 24)) : 0u;
    };
    int total = 0;
    for (int base = 0; base < nall; base += 64) {
      const int c = base + lane;
      total += __popcll(__ballot(c < nall && reach_bits(c) != 0u));
    }
    // the host has checked that the mask reaches at most V7_MAX_BLOCKS blocks; should a launch get here with more anyway, the pair
    // gets EMPTY lists (-1 / -inf), never truncated ones
    const int ncand = total > V7_MAX_BLOCKS ? 0 : nall;
    int count = 0;
    const int head = (total + 1) >> 1;
    for (int base = 0; base < ncand; base += 64) {
      const int c = base + lane;
      const uint32_t ent = c < ncand ? reach_bits(c) : 0u;
      const unsigned long long bal = __ballot(ent != 0u);
      if (ent) {
        const int r = count + __popcll(bal & ((1ull << lane) - 1));
        blist[(p.debug & 8192) ? (r < head ? 2 * r : 2 * (total - 1 - r) + 1) : r] = ent;
      }
      count += __popcll(bal);
    }
    if (lane == 0) blist_n = (p.debug & 524288) ? 0 : count;      // 524288: ablation (results wrong): no key blocks -- what a workgroup costs by itself
  }
  if (tid < NSLOT) {
    filled[tid] = 0;
    done[tid] = 0;
    hand_full[tid] = 0;
    hand_free[tid] = 0;
  }
  if (tid == 0) wg_dead = (p.debug & 4096) ? 1 : 0;      // 4096: fault injection for the fail-closed test
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (p.debug & 1048576) return;                        // ablation (nothing written): launch + query rows + block list only
  const int n_steps = blist_n;
  bool dead = false;
  // Wave priority: the SIMD arbitrates vector issue by priority, then age, and the selector -- the wave with the most vector work -- is
  // younger than its consumer.  One s_setprio 1 for the selectors: 1.155 -> 1.131 ms per 480p launch (priority 3 the same; a prioritised
  // consumer nothing, a prioritised producer +9 %: tools/experiments/prio_pair_v7.py).  debug & 16384: everybody at priority 0.
  if (role == 1 && !(p.debug & 16384)) __builtin_amdgcn_s_setprio(1);
  if (role == 0 && (p.debug & 65536)) __builtin_amdgcn_s_setprio(1);       // (experiment switches; results unchanged)
  if (role == 2 && (p.debug & 131072)) __builtin_amdgcn_s_setprio(1);
  if (role == 2) {
    // =========================================== producer: pixel row qb of every key block ===========================================
    __syncthreads();                                                 // the consumers have read their query operands: the ring is free
    const uint32_t lane16 = 16u * lane;
    const int n_total = g_count * n_steps;
    int cur_pair = -1;
    const unsigned char* kbase = nullptr;
    // Key rows through the producer's REGISTERS (global_load_dwordx4 -> ds_write_b128), PD blocks in flight (LDS-DMA, the f16x3 kernel's way, was measured beside it: the same launch time).  Measured on the
    // f16x3 kernel's ring (profiles/r04_pair_ring.log): a CU takes in one 1-KiB LDS-DMA instruction per ~33 cycles whatever the four
    // producer waves do -- 0.63 ms for this launch's 317 520 key blocks even with every row an L2 hit and half its bytes masked off, 0.88
    // as it is -- while the vector-memory path delivers 64 B per cycle and CU, twice that, and a producer wave owns 168 registers it has
    // no other use for: 8 rows x 4 registers per block, four blocks = 128 registers of prefetch that do not wait for a free LDS slot.
    // A block's rows are stored once its slot is free; the LDS executes a wave's operations in order, so the `filled` count that
    // follows the eight stores is seen after their data.
    constexpr int PD = 4;
    i32x4v R[PD][8];
    int iss_pi = 0, iss_e = 0;                                       // (pair, list entry) of the next block to issue: blocks are issued in order
    auto issue = [&](int G, i32x4v (&r)[8]) {                        // (no division per block: an integer division is a dozen vector operations)
      const int pi = iss_pi, e = iss_e;
      if (++iss_e == n_steps) { iss_e = 0; ++iss_pi; }
      if (pi != cur_pair) {
        cur_pair = pi;
        const int kf = p.pairs[g_start + pi].y;
        kbase = reinterpret_cast<const unsigned char*>(p.k_hl) + (size_t)kf * p.Hk * p.Wk * p.rowb;
      }
      const uint32_t ent = __builtin_amdgcn_readfirstlane(blist[e]);
      const int sby = ent & 0xfff, sbx = (ent >> 12) & 0xfff;
      const int ky = imin(sby * QBH + qb, p.Hk - 1), kx0 = sbx * QBW;
      const unsigned char* src = kbase + ((size_t)ky * p.Wk + kx0) * p.rowb;
      const int xmax = p.Wk - 1 - kx0;
      if ((p.debug & 1) || lane >= P6_END / 16) return;                  // (1: ablation, results wrong: the ring's counters only, no bytes moved)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned char* rowp = src + (size_t)imin(i, xmax) * p.rowb;      // wave-uniform: scalar base + one lane-offset register
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[i]) : "v"(lane16), "s"(rowp) : "memory");
      }
    };
    const uint32_t dst_lane = lds_addr_of(smem) + (uint32_t)((qb * 8) * LDB) + lane16;
    auto commit = [&](int G, i32x4v (&r)[8]) {
      const int slot = G & (NSLOT - 1), gen = G / NSLOT;
      if (gen > 0) spin_ge<1, false>(&done[slot], 4 * gen, dead, &wg_dead);     // block G - NSLOT released by all four consumers
      asm volatile("" ::: "memory");
      const uint32_t dst = dst_lane + (uint32_t)(slot * BUFB);
      if ((p.debug & 1) == 0 && lane < P6_END / 16) {                    // 58 lanes x 16 B = the 928 bytes of a row that carry something
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(dst), "v"(r[i]), "n"(i * LDB) : "memory");
      }
      asm volatile("" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(&filled[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
#pragma unroll
    for (int b = 0; b < PD; ++b)
      if (b < n_total) issue(b, R[b]);
    for (int G0 = 0; G0 < n_total; G0 += PD) {
#pragma unroll
      for (int b = 0; b < PD; ++b) {
        const int G = G0 + b;
        if (G < n_total) {                                               // wave-uniform
          // the loads return in order: block G is complete once at most the 8 (PD - 1) loads of the younger blocks are outstanding
          const int younger = imin(PD - 1, n_total - 1 - G);
          if (younger >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
          else if (younger == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          commit(G, R[b]);
          if (G + PD < n_total) issue(G + PD, R[b]);
        }
      }
    }
    if (dead) g_pair_v5_timeout = 1;
    return;
  }

  const int n_loop = n_steps;
  if (role == 0) {
    // =============================================== consumer: query block qb ===============================================
    // resident B operands of query n (lane half hi): the 16 f16 fragments as four 16-register groups (a group = the 32 elements of
    // 64-channel group v: MFMA operands by quarters, the FP6 conversion's source as a whole), the l6 pieces, the scale bytes
    f16x32 qh4[4];
    i32x6 q6l[4];
    int sqH, sqL;
    {
      const unsigned char* qp = &smem[qb * BUFB + n * LDB + 16 * hi];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const f16x8 f = *reinterpret_cast<const f16x8*>(qp + 32 * (4 * v + m));
#pragma unroll
          for (int i = 0; i < 8; ++i) qh4[v][8 * m + i] = f[i];
        }
        q6l[v] = v7_cat6(*reinterpret_cast<const i32x4v*>(qp + P6_L6M + 32 * v),
                         *reinterpret_cast<const i32x2v*>(qp + P6_L6T + 32 * (v >> 1) + 8 * (v & 1)));
      }
      const i32x2v sc = *reinterpret_cast<const i32x2v*>(qp + P6_SC);
      sqH = sc[0];
      sqL = sc[1];
    }
    // the queries' own h6 pieces: the same for every tile.  Made once (v_cvt_scalef32_pk32_fp6_f16: 2^sh = (scale byte + 4) << 23 as a
    // float) and parked in this consumer's 6 KiB of LDS -- mains [v][lane] 16 B, tails [v][lane] 8 B: contiguous, conflict-free
    const uint32_t a_q6h = lds_addr_of(&q6h_lds[qb][16 * lane]), a_q6t = lds_addr_of(&q6h_lds[qb][4096 + 8 * lane]);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      u32x6 h6;
      const unsigned int sc_bits = (unsigned int)(((sqH >> (8 * v)) & 255) + 4) << 23;
      asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(h6) : "v"(qh4[v]), "v"(sc_bits));      // (early-clobber: the result must not land on the scale)
      *reinterpret_cast<i32x4v*>(&q6h_lds[qb][1024 * v + 16 * lane]) = i32x4v{(int)h6[0], (int)h6[1], (int)h6[2], (int)h6[3]};
      *reinterpret_cast<i32x2v*>(&q6h_lds[qb][4096 + 512 * v + 8 * lane]) = i32x2v{(int)h6[4], (int)h6[5]};
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      asm volatile("" : "+v"(qh4[v]));
      asm volatile("" : "+v"(q6l[v]));
    }
    asm volatile("" : "+v"(sqH), "+v"(sqL));
    __syncthreads();                               // the ring is free for key blocks
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;     // (never read before a chain has written it; defined for the compiler's sake)
    int t_con = 0;                                 // tiles handed over so far (over the whole run)
    unsigned int* const hw = &hand[qb][4 * lane];
    const uint32_t lane_off = (uint32_t)(n * LDB + 16 * hi);
    const uint32_t smem_l = lds_addr_of(smem);
    const uint32_t a_hand_free = lds_addr_of(&hand_free[qb]);
    // The consumer's stream is CONTINUOUS over the tiles of a run (round 4; measured before: a chain of 16 MFMAs took 1 150 cycles
    // whatever the depth of its fragment ring -- 576 for the MFMAs, the rest an LDS round trip at its head and a drain at its end).
    // Which list entries this consumer computes is a 64-bit mask in scalar registers; inside a tile's chain (after the last read of
    // its key block) the NEXT tile is found, its key block waited for -- blocks land in list order, so it covers the entries skipped in
    // between, which are released on the spot -- and the chain's last quarter issues the next tile's first reads into the registers its
    // own last MFMAs have just consumed.  Between two chains only the hand-over remains (scores -> keys in place, four LDS stores).
    unsigned long long comp_mask = __ballot(lane < n_steps && ((blist[imin(lane, V7_MAX_BLOCKS - 1)] >> (24 + qb)) & 1u) != 0u);
    const int n_total = g_count * n_steps;
    // first list position >= the one after (cur_pi, cur_e) (over the run) that this consumer computes; n_total: none.  The position is
    // kept as (pair, entry) so that no step divides by n_steps (an integer division is a dozen vector operations on the port this
    // kernel is bound by)
    const int first_e = comp_mask ? __builtin_ctzll(comp_mask) : 0;
    int cur_pi = 0, cur_e = -1;
    auto next_comp = [&]() -> int {
      if (comp_mask == 0ull) return n_total;
      const unsigned long long m = cur_e >= 63 ? 0ull : (comp_mask >> (cur_e + 1));
      if (m) cur_e += 1 + __builtin_ctzll(m);
      else { ++cur_pi; cur_e = first_e; }
      return cur_pi < g_count ? cur_pi * n_steps + cur_e : n_total;
    };
    auto wait_block = [&](int G) { spin_ge<1, false>(&filled[G & (NSLOT - 1)], 4 * (G / NSLOT + 1), dead, &wg_dead); };
    // Entries [G0, G1) are not this consumer's: release each once it has landed (a release must never be sent for a block that is not
    // staged yet: the counters are per slot, pair_topk_kernel_v5).  Blocks land in list order -- every producer wave stores its rows of
    // block G before those of G + 1 -- so waiting for the last of up to NSLOT entries covers the group; never further ahead than that:
    // the producers cannot stage block G + NSLOT before THIS consumer has released block G.
    auto skip = [&](int G0, int G1) {
      for (int g = G0; g < G1;) {
        const int h = imin(g + NSLOT - 1, G1 - 1);
        wait_block(h);
        for (int G = g; G <= h; ++G)
          if (lane == 0) __hip_atomic_fetch_add(&done[G & (NSLOT - 1)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        g = h + 1;
      }
    };
    const bool probe = PROBE && blockIdx.x == 100 && blockIdx.y == 5;       // debug & 256: s_memtime stamps (tools/experiments/time_pair_v7.py)
    long long pr_wait = 0, pr_hand = 0, pr_hwait = 0, pr_chain = 0, pr_t0 = probe ? __builtin_amdgcn_s_memtime() : 0;
    int pr_tiles = 0, pr_slow = 0;
    int G_cur = next_comp();
    skip(0, G_cur);
    if (G_cur < n_total) {
      wait_block(G_cur);
      uint32_t ka_l = smem_l + (uint32_t)((G_cur & (NSLOT - 1)) * BUFB) + lane_off, ka_n = ka_l;
      f16x8 ah[4];
      i32x4v xm, ym, qm;
      i32x2v xt, yt, qt, ksc;
      int peek_free, peek_fill;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the hand-counted waits below count the stream's own reads only)
#define FGVC_V7_PART 1
#include "pair_v7.inc"
#undef FGVC_V7_PART
      for (;;) {
        const int slot = G_cur & (NSLOT - 1);
        // the next tile: pure scalar arithmetic on the mask; the counter of its key block is asked for inside the chain
        const int G_next = next_comp();
        const int G_peek = imin(G_next, n_total - 1);
        const uint32_t a_next_filled = lds_addr_of(&filled[G_peek & (NSLOT - 1)]);
        const long long c0 = probe ? __builtin_amdgcn_s_memtime() : 0;
#define V7_RELEASE()                                                                                                         \
  do {                                                                                                                       \
    /* every LDS read of this block has been issued (the LDS executes a wave's operations in order): release the slot */     \
    if (lane == 0) __hip_atomic_fetch_add(&done[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                  \
    asm volatile("" ::: "memory");                                                                                           \
  } while (0)
#define V7_LOOKAHEAD()                                                                                                       \
  do {                                                                                                                       \
    /* fast path: the block asked about has landed (so have the entries skipped before it: blocks land in list order) and   \
       is within NSLOT of this one (the producers did not need any of the skipped entries released to stage it) */           \
    if (G_peek - G_cur <= NSLOT && __builtin_amdgcn_readfirstlane(peek_fill) >= 4 * (G_peek / NSLOT + 1)) {                  \
      for (int G = G_cur + 1; G < G_next; ++G)                                                                               \
        if (lane == 0) __hip_atomic_fetch_add(&done[G & (NSLOT - 1)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);    \
    } else {                                                                                                                 \
      const long long w0 = probe ? __builtin_amdgcn_s_memtime() : 0;                                                         \
      skip(G_cur + 1, G_next);                                                                                               \
      if (G_next < n_total) wait_block(G_next);                                                                              \
      if (probe) { pr_wait += __builtin_amdgcn_s_memtime() - w0; ++pr_slow; }                                                \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* (the compiler's own LDS reads above, not counted below) */     \
    }                                                                                                                        \
    ka_n = G_next < n_total ? smem_l + (uint32_t)((G_next & (NSLOT - 1)) * BUFB) + lane_off : ka_l;                          \
  } while (0)
#define FGVC_V7_PART 3
#include "pair_v7.inc"
#undef FGVC_V7_PART
#undef V7_RELEASE
#undef V7_LOOKAHEAD
#define FGVC_V7_PART 5
#include "pair_v7.inc"
#undef FGVC_V7_PART
        // MFMA result -> vector read: the last MFMA's passes must have written back
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc), "+v"(peek_free), "+v"(peek_fill));
        if (probe) { pr_chain += __builtin_amdgcn_s_memtime() - c0; ++pr_tiles; }
        // ---- hand the tile over (the next tile's first reads are in flight): scores -> keys, four stores, the count
        {
          const long long h0 = probe ? __builtin_amdgcn_s_memtime() : 0;
          if ((p.debug & 2) == 0) {                // (2: ablation, results wrong: nothing handed over, the selectors have gone home)
            if (__builtin_amdgcn_readfirstlane(peek_free) < t_con) {
              spin_ge<1, false>(&hand_free[qb], t_con, dead, &wg_dead);      // the selector has read the tile before
              if (probe) { pr_hwait += __builtin_amdgcn_s_memtime() - h0; ++pr_slow; }
            }
            asm volatile("" ::: "memory");
            // the raw accumulators: the selector (whose vector unit has the slack) turns them into keys
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
              *reinterpret_cast<f32x4*>(hw + g4 * 256) = f32x4{acc[4 * g4 + 0], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]};
            asm volatile("" ::: "memory");           // the LDS executes a wave's operations in order: the count follows the data
            if (lane == 0) __hip_atomic_fetch_add(&hand_full[qb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ++t_con;
          }
          if (probe) pr_hand += __builtin_amdgcn_s_memtime() - h0;
        }
#define FGVC_V7_PART 4
#include "pair_v7.inc"
#undef FGVC_V7_PART
        if (G_next >= n_total) break;
        G_cur = G_next;
        ka_l = ka_n;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the last tile's look-ahead reads: nobody's operands)
    }
    if (probe && lane == 0) {
      long long* o = &g_pair_v5_probe[8 * qb];
      o[0] = __builtin_amdgcn_s_memtime() - pr_t0; o[1] = pr_wait; o[2] = pr_hand; o[3] = pr_hwait; o[4] = pr_chain; o[5] = pr_tiles; o[6] = pr_slow; o[7] = n_steps;
    }
    if (dead) g_pair_v5_timeout = 1;
    return;
  }

  // =============================================== selector of query block qb ===============================================
  __syncthreads();                                 // (the consumers read their query operands)
  if (p.debug & 2) return;
  unsigned int lk[K];                              // running list, ASCENDING: lk[0] = K-th best ... lk[K-1] = best; 0 = empty
  unsigned int ck[16];
  int v_dy0 = 0, v_dx0 = 0;
  int s_r2lim = -1;
  auto select_tile = [&]() {
#define FGVC_V7_PART 2
    if constexpr (K == 10) {
#define FGVC_V7_K 10
#include "pair_v7.inc"
#undef FGVC_V7_K
    } else {
#define FGVC_V7_K 5
#include "pair_v7.inc"
#undef FGVC_V7_K
    }
#undef FGVC_V7_PART
  };
  int t_sel = 0;
  const unsigned int* const hr = &hand[qb][4 * lane];
  for (int pi = 0; pi < g_count; ++pi) {           // ---- the pairs of the run
#pragma unroll
    for (int j = 0; j < K; ++j) lk[j] = 0u;
    for (int e = 0; e < n_loop; ++e) {
      const uint32_t ent = __builtin_amdgcn_readfirstlane(blist[e]);
      if (((ent >> (24 + qb)) & 1) == 0) continue;   // not a tile of this query block
      spin_ge<1, false>(&hand_full[qb], t_sel + 1, dead, &wg_dead);
      asm volatile("" ::: "memory");
      float sc_[16];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(hr + g4 * 256);
        sc_[4 * g4 + 0] = kv.x; sc_[4 * g4 + 1] = kv.y; sc_[4 * g4 + 2] = kv.z; sc_[4 * g4 + 3] = kv.w;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      {   // scores -> keys: (bits(acc + 2^19 + 2^17) << 10) | (63 - list position) << 4 | (15 - register)
        const unsigned int tag0 = ((unsigned int)(63 - e) << 4) | 15u;
#pragma unroll
        for (int r = 0; r < 16; ++r) ck[r] = (__builtin_bit_cast(unsigned int, sc_[r] + V7_BIAS) << 10) | (tag0 - r);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(ck[r]));
      if (lane == 0) __hip_atomic_fetch_add(&hand_free[qb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      ++t_sel;
      {
        const int ky0 = (int)(ent & 0xfff) * QBH, kx0 = (int)((ent >> 12) & 0xfff) * QBW;
        const bool interior = ky0 + QBH <= p.Hk && kx0 + QBW <= p.Wk;
        const bool circle_only = reach.ry >= FGVC_NO_LIMIT && reach.rx >= FGVC_NO_LIMIT;
        if (interior && circle_only) {              // wave-uniform
          v_dy0 = ky0 - qy;
          v_dx0 = kx0 + 4 * hi - qx;
          s_r2lim = reach.r2max;
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dy = ky0 - qy + (r >> 2), dx = kx0 + 4 * hi - qx + (r & 3);
            const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
            const int cy = imin(ady, 32767), cx = imin(adx, 32767);
            const bool ok = (ky0 + (r >> 2) < p.Hk) & (kx0 + 4 * hi + (r & 3) < p.Wk) &
                            (cy * cy + cx * cx <= reach.r2max) & (ady <= reach.ry) & (adx <= reach.rx);
            ck[r] = ok ? ck[r] : 0u;
          }
          v_dy0 = 0;
          v_dx0 = 0;
          s_r2lim = FGVC_NO_LIMIT;                  // (r & 3)^2 <= FGVC_NO_LIMIT - (r >> 2)^2: always kept
        }
      }
      if ((p.debug & 2048) == 0) select_tile();     // 2048: ablation (results wrong): hand-over only
    }

    // ---- epilogue: two partial lists per query (the two lane halves) -> canonical top-K.  Entries become 64-bit words
    //      (score : ~pixel), larger = better (higher score, then LOWER pixel index)
    long long L[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const unsigned int key = lk[i];
      const int r = 15 - (int)(key & 15u);
      const uint32_t bent = blist[63 - (int)((key >> 4) & 63u)];
      const int pix = ((int)(bent & 0xfff) * QBH + (r >> 2)) * p.Wk + (int)((bent >> 12) & 0xfff) * QBW + 4 * hi + (r & 3);
      const bool em = key == 0u;
      L[i] = em ? 0ll : (long long)(((unsigned long long)(key >> 10) << 32) | (unsigned long long)(~(uint32_t)pix));
    }
    // a lane's list is ascending in (score, tag); entries with the same score may be out of pixel order
#define X(I, J)                                   \
  {                                               \
    const bool b_ = L[I] < L[J];                  \
    const long long lo_ = b_ ? L[I] : L[J];       \
    const long long hi_ = b_ ? L[J] : L[I];       \
    L[J] = lo_; L[I] = hi_;                       \
  }
    if constexpr (K == 10) { FGVC_SORTNET_10(X) }    // descending in I < J ...
    else { FGVC_SORTNET_5(X) }
#undef X
#pragma unroll
    for (int i = 0; i < K / 2; ++i) {                // ... so reverse: ascending like the running list
      const long long tmp = L[i];
      L[i] = L[K - 1 - i];
      L[K - 1 - i] = tmp;
    }
    {
      long long B[K];
#pragma unroll
      for (int i = 0; i < K; ++i) B[i] = __shfl_xor(L[i], 32);
#pragma unroll
      for (int i = 0; i < K; ++i) L[i] = L[i] > B[K - 1 - i] ? L[i] : B[K - 1 - i];
#define X(I, J)                                   \
    {                                             \
      const bool b_ = L[I] > L[J];                \
      const long long lo_ = b_ ? L[J] : L[I];     \
      const long long hi_ = b_ ? L[I] : L[J];     \
      L[I] = lo_; L[J] = hi_;                     \
    }
      if constexpr (K == 10) { FGVC_VMERGE_ASC_10(X) }
      else { FGVC_VMERGE_ASC_5(X) }
#undef X
    }
    const bool poison = dead || __builtin_amdgcn_readfirstlane(*(volatile int*)&wg_dead) != 0;
    if (poison) g_pair_v5_timeout = 1;
    if (hi == 0 && qy < p.Hq && qx < p.Wq) {
      const size_t oo = ((size_t)(g_start + pi) * p.Hq * p.Wq + (size_t)qy * p.Wq + qx) * p.kout;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (j < p.kout) {
          const long long v = L[K - 1 - j];
          const int sk = (int)(v >> 32);              // 22-bit score: (cos + 2) 2^20
          const bool em = sk == 0;
          p.idx_out[oo + j] = poison ? 0 : em ? -1 : (int)~(uint32_t)v;
          p.score_out[oo + j] = poison ? PAIR_POISON_SCORE : em ? -INFINITY : (float)sk * 0x1p-20f - 2.0f;
        }
      }
    }
  }   // pairs of the run
  if (dead) g_pair_v5_timeout = 1;
}

// key blocks (QBH x QBW pixels) an interior 2 QBH x 2 QBW query tile reaches under the mask predicate (the kernel lists exactly these)
static int pair_v7_blocks_reached(int r2max, int ry, int rx, int reach_y, int reach_x) {
  const int ny = (2 * QBH - 1 + 2 * reach_y) / QBH + 2, nx = (2 * QBW - 1 + 2 * reach_x) / QBW + 2;
  const int TY0 = (reach_y / QBH + 1) * QBH, TX0 = (reach_x / QBW + 1) * QBW;      // a tile far enough from the frame's top-left corner
  int count = 0;
  for (int by = 0; by < ny + reach_y / QBH + 2; ++by)
    for (int bx = 0; bx < nx + reach_x / QBW + 2; ++bx) {
      bool any = false;
      for (int b = 0; b < 4; ++b) {
        const int wy0 = TY0 + (b & 1) * QBH, wx0 = TX0 + (b >> 1) * QBW, ky0 = by * QBH, kx0 = bx * QBW;
        const long long dy = imax(0, imax(ky0 - (wy0 + QBH - 1), wy0 - (ky0 + QBH - 1)));
        const long long dx = imax(0, imax(kx0 - (wx0 + QBW - 1), wx0 - (kx0 + QBW - 1)));
        any = any || (dy * dy + dx * dx <= r2max && dy <= ry && dx <= rx);
      }
      count += any;
    }
  return count;
}

int pair_topk_v8_launch(const PairParamsB& p0, int n_pairs, int n_groups, int topk, hipStream_t s);      // pair_topk_v8.hpp

int pair_topk_v7_launch(const uint16_t* q_sp, const uint16_t* k_sp, const int32_t* pairs, int n_pairs, int Hq, int Wq, int Hk, int Wk,
                        int r2max, int ry, int rx, int topk, const int32_t* groups, int n_groups, int32_t* idx_out, float* score_out,
                        int row_bytes, hipStream_t s) {
  PairParamsB p;
  p.rowb = row_bytes;
  p.q_hl = q_sp; p.k_hl = k_sp; p.pairs = reinterpret_cast<const int4*>(pairs);
  p.Hq = Hq; p.Wq = Wq; p.Hk = Hk; p.Wk = Wk;
  p.r2max = r2max; p.ry = ry; p.rx = rx;
  int rr = 0;
  while (rr < 46340 && (long long)(rr + 1) * (rr + 1) <= (long long)r2max) ++rr;
  p.reach_y = imin(ry, rr); p.reach_x = imin(rx, rr);
  p.kout = topk;
  p.n_ty = cdiv(Hq, 2 * QBH); p.n_tx = cdiv(Wq, 2 * QBW);
  p.idx_out = idx_out; p.score_out = score_out;
  p.groups = reinterpret_cast<const int2*>(groups);
  p.debug = g_pair_v5_debug;
  if (p.reach_y > 4096 || p.reach_x > 4096 || pair_v7_blocks_reached(r2max, ry, rx, p.reach_y, p.reach_x) > V7_MAX_BLOCKS) {
    set_error("fgvc_pair_topk_f16f6: the mask reaches more than %d key blocks per query tile; use fgvc_pair_topk_f16x3", V7_MAX_BLOCKS);
    return FGVC_ERR_UNSUPPORTED;
  }
  // round 6: the one-role kernel (pair_topk_v8.hpp: a key block staged once for EIGHT query blocks, every wave multiplies and selects
  // for itself) is bit-identical and moves 0.65 of the bytes, but 4-7 % SLOWER at every BASELINE shape (profiles/r06_pair_v8.log): it
  // is kept behind pair_f16_debug & 4194304 (tests hold the two kernels against each other), this kernel stays the default
  if ((p.debug & 4194304) && p.reach_y <= 16 && p.reach_x <= 16 && (long long)Hk * Wk * row_bytes < (1ll << 31))
    return pair_topk_v8_launch(p, n_pairs, n_groups, topk, s);
  dim3 grid(p.n_ty * p.n_tx, groups ? n_groups : n_pairs);
  if (topk <= 5) pair_topk_kernel_v7<5, false><<<grid, 768, 0, s>>>(p);
  else if (p.debug & 256) pair_topk_kernel_v7<10, true><<<grid, 768, 0, s>>>(p);
  else pair_topk_kernel_v7<10, false><<<grid, 768, 0, s>>>(p);
  FGVC_CHECK_LAUNCH("fgvc_pair_topk_f16f6");
  return FGVC_OK;
}

}  // namespace fgvc
