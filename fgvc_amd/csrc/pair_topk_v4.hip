// fgvc_pair_topk_bf16x4: the windowed correlation + top-k of fgvc_pair_topk_f32 on the bf16 matrix pipe, f32-grade.
//
// Why: v3 (pair_topk_v3.hip) is bound by v_mfma_f32_32x32x2_f32 -- 128 MFMAs x 64 cycles = 8192 cycles per 32x32 tile.
// The bf16 pipe is 16x faster per FLOP.  With every feature value split into hi = bf16(x), lo = bf16(x - hi)
// (fgvc_split_bf16; x - hi - lo <= 2^-18 |x|) the four products hi*hi + hi*lo + lo*hi + lo*lo reproduce the f32 dot
// product to ~1e-7 (products of bf16 are exact in f32, accumulation is f32): 64 v_mfma_f32_32x32x16_bf16 = 2048
// cycles per tile.  At that rate the selection, not the MFMA, sets the time, so it is rebuilt around integer keys:
//   * queries are pre-scaled by 2^28 (exponent add on the bf16 fragments, exact), the accumulator is converted to a
//     fixed-point integer whose low 4 bits are replaced by the candidate's register number:
//     key = (score * 2^28 & ~15) | (15 - r), i.e. 24 fractional score bits (6e-8, the f32 resolution at |score| ~ 1).  One 32-bit word orders a tile's candidates by
//     (score desc, key pixel asc), so a compare-exchange is v_max_i32 + v_min_i32 instead of v_cmp + 4 v_cndmask;
//   * 16 candidates -> 60-comparator selection network (top 10, sorted);  running list kept ASCENDING so that
//     max(list[i], cand[i]) is V-shaped and a 15-comparator bitonic merger (Lang's arbitrary-n form) re-sorts it;
//     the list's payload is the base pixel of the block a key came from (the low 4 key bits give the rest).
// Work split: 8 waves = 4 query blocks (2x2 blocks of 4x8 pixels) x 2 parities.  Key blocks of the super-tile's
// reach are streamed through a 4-slot LDS ring by LDS-DMA (1 KiB per key pixel: [hi 256 | lo 256] bf16); the block of
// step t is multiplied by the waves of parity t&1 while the other four stage block t+2 and convert/sort/merge the
// tile they produced at step t-1 (its accumulators stay in registers across the barrier) -- the two waves of a SIMD
// alternate between the matrix pipe and the VALU.  One LDS barrier per step.  The sequence of key blocks (and which of
// the four query blocks reach each) is computed ONCE per workgroup into an LDS list: evaluating the reach predicate in
// every wave at every step made the CU's scalar unit the bottleneck (measured: 0.9 us per step with nothing else on).
// Each query ends with 4 partial lists (2 parities x 2 lane halves), merged canonically at the end.
//
// Precondition: feature rows are L2-normalised (|q.k| <= 1 up to rounding), as fgvc_normalize_chw_to_hwc_f32 makes
// them; the fixed-point key holds |score| < 8.
#include "pair_common.hpp"

namespace fgvc {

template <int K, int NPROD>
__global__ __launch_bounds__(512, 1) void pair_topk_kernel_v4(PairParamsB p) {
  constexpr int C = 256;
  constexpr int LDB = 2 * C * 2 + 16;          // padded LDS row of one pixel: [hi | lo] + 16 B -> conflict-free b128
  constexpr int BUFB = 32 * LDB;
  constexpr int NSLOT = 4;
  constexpr int KS = C / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * BUFB];
  __shared__ uint32_t blist[V4_LIST_CAP];      // by | bx << 12 | (query blocks that reach it) << 24, row-major
  __shared__ int blist_n;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = wave & 3, par = wave >> 2;      // waves w and w+4 share a SIMD: they alternate roles
  const int n = lane & 31, hi = lane >> 5;

  const int4 pr = p.pairs[blockIdx.y];
  const int qf = pr.x, kf = pr.y;
  const bool masked = (pr.z & FGVC_PAIR_MASKED) != 0;
  const int reach_y = masked ? p.reach_y : FGVC_NO_LIMIT;
  const int reach_x = masked ? p.reach_x : FGVC_NO_LIMIT;

  const int tile = xcd_remap(blockIdx.x, p.n_ty * p.n_tx);
  const int ty = tile / p.n_tx, tx = tile - ty * p.n_tx;
  ReachTest reach;
  reach.r2max = masked ? p.r2max : FGVC_NO_LIMIT;
  reach.ry = masked ? p.ry : FGVC_NO_LIMIT;
  reach.rx = masked ? p.rx : FGVC_NO_LIMIT;
  const int TY0 = ty * (2 * QBH), TX0 = tx * (2 * QBW);
  const int QY0 = TY0 + (qb & 1) * QBH, QX0 = TX0 + (qb >> 1) * QBW;
  const int qy = QY0 + (n >> 3), qx = QX0 + (n & 7);

  // ---- prologue 1: the query rows of the four blocks through the ring (coalesced 1 KiB rows by LDS-DMA; per-lane
  //      16-byte gathers of 32 rows per instruction kept the CU's vector-memory path busy for ~9 us per workgroup)
  {
    const uint16_t* qbase = p.q_hl + (size_t)qf * p.Hq * p.Wq * (2 * C) + 8 * lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = par * 16 + i;                                  // row of query block qb
      const int y = imin(QY0 + (r >> 3), p.Hq - 1), x = imin(QX0 + (r & 7), p.Wq - 1);
      lds_dma_16(qbase + ((size_t)y * p.Wq + x) * (2 * C), &smem[qb * BUFB + r * LDB]);
    }
  }
  // ---- prologue 2 (overlaps the DMA): the list of key blocks this super-tile visits
  if (wave == 0) {
    const int by_lo = imax(0, TY0 - imin(reach_y, TY0)) / QBH;
    const int by_hi = imin(p.Hk - 1, TY0 + 2 * QBH - 1 + imin(reach_y, p.Hk)) / QBH;
    const int bxl = imax(0, TX0 - imin(reach_x, TX0)) / QBW;
    const int bxh = imin(p.Wk - 1, TX0 + 2 * QBW - 1 + imin(reach_x, p.Wk)) / QBW;
    const int nbx = bxh - bxl + 1;
    // the host has checked that a MASKED pair's reach fits the list; an unmasked pair on a larger key grid than the list holds
    // (the caller promised there was none: `all_masked`) gets EMPTY lists (-1 / -inf), never truncated ones
    const int nall = (by_hi - by_lo + 1) * nbx;
    const int ncand = nall > V4_LIST_CAP ? 0 : nall;
    int count = 0;
    for (int base = 0; base < ncand; base += 64) {
      const int c = base + lane;
      const int by = by_lo + c / nbx, bx = bxl + c % nbx;
      uint32_t m = 0;
      if (c < ncand) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
          m |= (uint32_t)reach(TY0 + (b & 1) * QBH, TX0 + (b >> 1) * QBW, by * QBH, bx * QBW) << b;
      }
      const unsigned long long bal = __ballot(m != 0);
      if (m) blist[count + __popcll(bal & ((1ull << lane) - 1))] = (uint32_t)by | ((uint32_t)bx << 12) | (m << 24);
      count += __popcll(bal);
    }
    if (lane == 0) blist_n = count;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // query fragments (MFMA B operand): lane (n, hi) holds channels 16j + 8hi .. +7 of query n, hi and lo parts, scaled by
  // 2^28 (add 28 to both bf16 exponents of every dword; a zero becomes 2^-99, harmless): the accumulator is then the
  // score in 2^-28 fixed point, of which the selection keeps 24 fractional bits -- every key value is an exact f32, so
  // "equal output score" and "equal key" are the same thing and ties can be ordered canonically
  bf16x8 qh[KS], ql[KS];
  {
    const unsigned char* qp = &smem[qb * BUFB + n * LDB + 16 * hi];
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      uint4 a = *reinterpret_cast<const uint4*>(qp + 32 * j);
      uint4 b = *reinterpret_cast<const uint4*>(qp + 2 * C + 32 * j);
      a.x += 0x0E000E00u; a.y += 0x0E000E00u; a.z += 0x0E000E00u; a.w += 0x0E000E00u;
      b.x += 0x0E000E00u; b.y += 0x0E000E00u; b.z += 0x0E000E00u; b.w += 0x0E000E00u;
      qh[j] = __builtin_bit_cast(bf16x8, a);
      ql[j] = __builtin_bit_cast(bf16x8, b);
    }
  }
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    asm volatile("" ::"v"(qh[j]));
    asm volatile("" ::"v"(ql[j]));
  }
  __syncthreads();                               // the ring is free for key blocks

  // uniform 64-bit base + 32-bit lane offset: the DMA then addresses as SGPR-pair + VGPR offset (no 64-bit VALU math)
  const unsigned char* kbase = reinterpret_cast<const unsigned char*>(p.k_hl) + (size_t)kf * p.Hk * p.Wk * (4 * C);
  const uint32_t lane16 = 16u * lane;
  // a wave stages pixel row `qb` of a key block (8 pixels = 8 KiB), one 1-KiB pixel per call
  // stage_begin computes the per-lane source of pixel 0 once; stage_pixel adds a clamped uniform offset (few scalar
  // ops, so that it fits between two MFMAs of the chain)
  const unsigned char* st_src = nullptr;
  int st_imax = 0;
  unsigned char* st_dst = nullptr;
  auto stage_begin = [&](uint32_t e, int slot) {
    const int sby = e & 0xfff, sbx = (e >> 12) & 0xfff;
    const int ky = imin(sby * QBH + qb, p.Hk - 1), kx0 = sbx * QBW;
    st_src = kbase + ((size_t)ky * p.Wk + kx0) * (4 * C) + lane16;
    st_imax = p.Wk - 1 - kx0;                                                       // >= 0: the block starts inside the frame
    st_dst = &smem[slot * BUFB + (qb * 8) * LDB];
  };
  auto stage_pixel = [&](int i) {
    lds_dma_16(st_src + (size_t)imin(i, st_imax) * (4 * C), st_dst + i * LDB);
  };
  const int n_steps = blist_n;
  const bool do_stage = (p.debug & 4) == 0;
  // every parity class stages the blocks it multiplies itself: block t during step t-2 (blocks 0 / 1 here)
  uint32_t e_cur = par < n_steps ? blist[par] : 0u;
  e_cur = __builtin_amdgcn_readfirstlane(e_cur);
  if (par < n_steps && do_stage) {
    stage_begin(e_cur, par);
#pragma unroll
    for (int i = 0; i < 8; ++i) stage_pixel(i);
  }

  int lk[K], lb[K];                              // running list, ASCENDING: lk[0] = K-th best ... lk[K-1] = best
#pragma unroll
  for (int j = 0; j < K; ++j) {
    lk[j] = KEY_EMPTY;
    lb[j] = -1;
  }
  f32x16 acc0, acc1;                             // tile of this wave's last MFMA step; converted one step later
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc0[r] = 0.f;
    acc1[r] = 0.f;
  }
  int cand_base = 0, cand_ky0 = 0, cand_kx0 = 0;

  // Runs unconditionally once per window (a branch around it makes hipcc copy the 20 list registers at the join, ~25 %
  // more VALU work): when the wave computed no tile, `tile_ok` = false turns the predicate off and every key is EMPTY.
  auto merge_tile = [&](bool tile_ok) {
    // fixed-point key (low 4 bits replaced by 15 - register number) + mask predicate
    int ck[16];
    const int r2lim = tile_ok ? reach.r2max : -1;
    {
      const int ky0 = cand_ky0, kx0 = cand_kx0;
      const int dy0 = ky0 - qy, dx0 = kx0 + 4 * hi - qx;
      const bool interior = ky0 + QBH <= p.Hk && kx0 + QBW <= p.Wk;
      const bool circle_only = reach.ry >= FGVC_NO_LIMIT && reach.rx >= FGVC_NO_LIMIT;
      if (interior && circle_only) {
        int ylim[4], xsq[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          ylim[a] = r2lim - __mul24(dy0 + a, dy0 + a);
          xsq[a] = __mul24(dx0 + a, dx0 + a);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int fx = (int)(acc0[r] + acc1[r]);
          ck[r] = (xsq[r & 3] <= ylim[r >> 2]) ? ((fx & ~15) | (15 - r)) : KEY_EMPTY;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dy = dy0 + (r >> 2), dx = dx0 + (r & 3);
          const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
          const bool ok = (ky0 + (r >> 2) < p.Hk) & (kx0 + 4 * hi + (r & 3) < p.Wk) &
                          (__mul24(dy, dy) + __mul24(dx, dx) <= r2lim) & (ady <= reach.ry) & (adx <= reach.rx);
          const int fx = (int)(acc0[r] + acc1[r]);
          ck[r] = ok ? ((fx & ~15) | (15 - r)) : KEY_EMPTY;
        }
      }
    }
#define X(I, J) FGVC_V4_DESC(ck, I, J)
    if constexpr (K == 10) { FGVC_SELNET_16_TOP10(X) }
    else { FGVC_SELNET_16_TOP5(X) }
#undef X
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const bool b = ck[i] > lk[i];
      lk[i] = max(ck[i], lk[i]);
      lb[i] = b ? cand_base : lb[i];
    }
#define X(I, J)                                     \
    {                                               \
      const bool b_ = lk[I] > lk[J];                \
      const int lo_ = min(lk[I], lk[J]);            \
      const int hi_ = max(lk[I], lk[J]);            \
      const int bi_ = b_ ? lb[J] : lb[I];           \
      const int bj_ = b_ ? lb[I] : lb[J];           \
      lk[I] = lo_; lk[J] = hi_;                     \
      lb[I] = bi_; lb[J] = bj_;                     \
    }
    if constexpr (K == 10) { FGVC_VMERGE_ASC_10(X) }
    else { FGVC_VMERGE_ASC_5(X) }
#undef X
  };

  if (p.debug & 16) return;                      // ablation: prologue only
  uint32_t e_own = e_cur;                                           // list entry of this wave's block in the window
  uint32_t e_next_v = (par + 2 < n_steps) ? blist[par + 2] : 0u;    // ... and in the next window (staged meanwhile)
  // the 64-MFMA product of key block t (must be this wave's: t = par mod 2) with the wave's 32 queries; stages block
  // t+2 meanwhile.  Returns false (nothing computed) when the block is outside this query block's reach.
  auto compute = [&](int t) -> bool {
    const uint32_t e = e_own;
    const bool more = t + 2 < n_steps;
    const uint32_t e2 = __builtin_amdgcn_readfirstlane(e_next_v);
    const bool stage = more && do_stage;
    if (stage) stage_begin(e2, (t + 2) & (NSLOT - 1));
    e_own = e2;
    e_next_v = (t + 4 < n_steps) ? blist[t + 4] : 0u;
    const bool comp = ((e >> (24 + qb)) & 1) != 0;
    if (comp && (p.debug & 2) == 0) {
      const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      const unsigned char* ka = &smem[(t & (NSLOT - 1)) * BUFB + n * LDB + 16 * hi];
      constexpr int G = 2, NG = KS / G;            // two k16 steps (8 MFMAs) of A fragments in flight
      bf16x8 ah[2][G], al[2][G];
#pragma unroll
      for (int i = 0; i < G; ++i) {
        ah[0][i] = *reinterpret_cast<const bf16x8*>(ka + 32 * i);
        al[0][i] = *reinterpret_cast<const bf16x8*>(ka + 2 * C + 32 * i);
      }
      __builtin_amdgcn_sched_barrier(0);
      // the matrix segment outranks the partner wave's VALU segment in the SIMD's issue arbitration (which otherwise goes
      // to the older wave): an MFMA needs 8 of every 32 issue cycles, the partner keeps the rest (measured -6 %)
      if ((p.debug & 128) == 0) __builtin_amdgcn_s_setprio(3);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
          for (int i = 0; i < G; ++i) {
            ah[(g + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(ka + 32 * ((g + 1) * G + i));
            al[(g + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(ka + 2 * C + 32 * ((g + 1) * G + i));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const int j = g * G + i;
          if ((j & 1) && stage) {                  // block t+2, one pixel per 8 MFMAs, issued in the shadow of a running MFMA
            __builtin_amdgcn_sched_barrier(0);
            stage_pixel(j >> 1);
            __builtin_amdgcn_sched_barrier(0);
          }
          // two chains from the zero constant: the leading terms (k_hi) in acc0, the corrections (k_lo) in acc1
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g & 1][i], qh[j], j == 0 ? zero : acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[g & 1][i], qh[j], j == 0 ? zero : acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g & 1][i], ql[j], acc0, 0, 0, 0);
          if constexpr (NPROD == 4) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[g & 1][i], ql[j], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if ((p.debug & 128) == 0) __builtin_amdgcn_s_setprio(0);
    } else if (stage) {
#pragma unroll
      for (int i = 0; i < 8; ++i) stage_pixel(i);
    }
    cand_ky0 = (int)(e & 0xfff) * QBH;
    cand_kx0 = (int)((e >> 12) & 0xfff) * QBW;
    cand_base = cand_ky0 * p.Wk + cand_kx0;
    return comp;
  };
  // One barrier per window of two key blocks (2w for parity 0, 2w+1 for parity 1).  Parity 0 multiplies first and
  // converts/sorts/merges second; parity 1 merges its previous tile first and multiplies second -- the two waves of a
  // SIMD hand the matrix pipe to each other without a barrier in between.
  const int n_win = (p.debug & 64) ? 0 : (n_steps + 1) / 2;
  const bool do_merge = (p.debug & 1) == 0;
  if (par == 0) {
    for (int w = 0; w < n_win; ++w) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // my DMA pixels (issued in the previous window) landed
      lds_barrier();
      const bool ok = compute(2 * w);
      if (do_merge) merge_tile(ok);
    }
  } else {
    bool ok = false;
    for (int w = 0; w < n_win; ++w) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      if (do_merge) merge_tile(ok);
      ok = (2 * w + 1 < n_steps) ? compute(2 * w + 1) : false;
    }
    if (do_merge) merge_tile(ok);
  }
  if (p.debug & 32) return;                      // ablation: no epilogue

  // ---- epilogue: four partial lists per query (2 lane halves x 2 parities) -> canonical top-K.  Entries become
  //      64-bit words (score_fx : ~pixel), larger = better (higher score, then LOWER pixel index); two sorted lists
  //      merge as max(A[i], B[K-1-i]) + the V-merger.
  long long L[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int key = lk[i];
    const int r = 15 - (key & 15);
    const int pix = lb[i] + 4 * hi + (r >> 2) * p.Wk + (r & 3);
    const bool e = key == KEY_EMPTY;
    const uint32_t lo = e ? 0u : ~(uint32_t)pix;
    L[i] = (long long)(((unsigned long long)(uint32_t)(key & ~15) << 32) | lo);
  }
  // a lane's list is ascending in (score, tag); entries of different tiles with the same score may be out of pixel order
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
  auto merge_with = [&](const long long (&B)[K]) {
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
  };
  {
    long long B[K];
#pragma unroll
    for (int i = 0; i < K; ++i) B[i] = __shfl_xor(L[i], 32);
    merge_with(B);
  }
  __syncthreads();                               // ring no longer needed: exchange buffer [i][query] of 64-bit words
  long long* xl = reinterpret_cast<long long*>(smem);
  if (par == 1 && hi == 0) {
#pragma unroll
    for (int i = 0; i < K; ++i) xl[i * 128 + qb * 32 + n] = L[i];
  }
  __syncthreads();
  if (par == 0) {
    long long B[K];
#pragma unroll
    for (int i = 0; i < K; ++i) B[i] = xl[i * 128 + qb * 32 + n];
    merge_with(B);
    if (hi == 0 && qy < p.Hq && qx < p.Wq) {
      const size_t oo = ((size_t)blockIdx.y * p.Hq * p.Wq + (size_t)qy * p.Wq + qx) * p.kout;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (j < p.kout) {
          const long long v = L[K - 1 - j];
          const int sk = (int)(v >> 32);
          const bool e = sk == KEY_EMPTY;
          p.idx_out[oo + j] = e ? -1 : (int)~(uint32_t)v;
          p.score_out[oo + j] = e ? -INFINITY : (float)sk * 0x1p-28f;
        }
      }
    }
  }
}

static int g_pair_v4_debug = 0;
static int g_pair_v4_products = 4;
void set_pair_v4_debug(int v) { g_pair_v4_debug = v; }
void set_pair_v4_products(int v) { g_pair_v4_products = v; }

int pair_topk_v4_launch(const uint16_t* q_hl, const uint16_t* k_hl, const int32_t* pairs, int n_pairs, int Hq, int Wq,
                        int Hk, int Wk, int r2max, int ry, int rx, int topk, int all_masked, int32_t* idx_out,
                        float* score_out, hipStream_t s) {
  PairParamsB p;
  p.q_hl = q_hl; p.k_hl = k_hl; p.pairs = reinterpret_cast<const int4*>(pairs);
  p.Hq = Hq; p.Wq = Wq; p.Hk = Hk; p.Wk = Wk;
  p.r2max = r2max; p.ry = ry; p.rx = rx;
  int rr = 0;  // floor(sqrt(r2max)) in integers
  while (rr < 46340 && (long long)(rr + 1) * (rr + 1) <= (long long)r2max) ++rr;
  p.reach_y = imin(ry, rr); p.reach_x = imin(rx, rr);
  p.kout = topk;
  p.n_ty = cdiv(Hq, 2 * QBH); p.n_tx = cdiv(Wq, 2 * QBW);
  p.idx_out = idx_out; p.score_out = score_out;
  p.groups = nullptr;
  p.debug = g_pair_v4_debug;
  {  // the per-workgroup block list must hold every key block a super-tile can reach: the mask's reach for a masked pair,
     // the whole key grid for a pair without FGVC_PAIR_MASKED (the caller says whether there is one: pairs live on the device)
    const long long nby = imin(cdiv(Hk, QBH), (2 * QBH - 1 + 2 * (long long)imin(p.reach_y, Hk)) / QBH + 2);
    const long long nbx = imin(cdiv(Wk, QBW), (2 * QBW - 1 + 2 * (long long)imin(p.reach_x, Wk)) / QBW + 2);
    const long long whole = (long long)cdiv(Hk, QBH) * cdiv(Wk, QBW);
    const long long need = all_masked ? nby * nbx : whole;
    if (need > V4_LIST_CAP || Hk >= 4096 * QBH || Wk >= 4096 * QBW) {
      set_error("fgvc_pair_topk_bf16x4: key grid %dx%d needs %lld > %d key blocks per query tile (%s); use fgvc_pair_topk_f32",
                Hk, Wk, need, V4_LIST_CAP, all_masked ? "mask reach" : "a pair without FGVC_PAIR_MASKED scans the frame");
      return FGVC_ERR_UNSUPPORTED;
    }
  }
  dim3 grid(p.n_ty * p.n_tx, n_pairs);
  if (g_pair_v4_products == 4) {
    if (topk <= 5) pair_topk_kernel_v4<5, 4><<<grid, 512, 0, s>>>(p);
    else pair_topk_kernel_v4<10, 4><<<grid, 512, 0, s>>>(p);
  } else {
    if (topk <= 5) pair_topk_kernel_v4<5, 3><<<grid, 512, 0, s>>>(p);
    else pair_topk_kernel_v4<10, 3><<<grid, 512, 0, s>>>(p);
  }
  FGVC_CHECK_LAUNCH("fgvc_pair_topk_bf16x4");
  return FGVC_OK;
}

}  // namespace fgvc
