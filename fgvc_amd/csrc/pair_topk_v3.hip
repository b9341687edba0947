// fgvc_pair_topk_f32, variant 3 (C == 256, topk <= 10, analytic mask): wave-specialised like v2, but the
// 16-candidate sorting network moves INTO the MFMA wave.
//
// Measurements that led here (tools/ablate_pair.py, MI355X): with selection in its own wave (v2) the MFMA side
// needs 5.1 ms for the 27 pairs of a 480p clip while the selection wave needs ~7 ms -- about 1300 VALU
// instructions per 32x32 tile, two thirds of them the 63-comparator network that sorts a lane's 16 candidates.
// The MFMA wave, in contrast, issues one v_mfma_f32_32x32x2_f32 per 64 cycles and leaves its VALU idle.  So:
//   M wave, step s:   128 MFMAs of tile s, cut into 8 fenced groups of 16; in the shadow of group g it runs piece g
//                     of the sorting network on the (masked) scores of tile s-1 which it kept in registers;
//                     then it publishes the top-K of tile s-1 (K scores + K indices per lane) to LDS.
//   S wave, step s:   stages key block s+1 by LDS-DMA, merges the published top-K of tile s-2 into its running
//                     list: max(cand[K-1-j], list[j]) (bitonic) + a K-wire re-sort  (~190 VALU ops).
// Everything else (two 8x8 halves with private key rings, block reach tests, barriers) is as in v2.
#include "pair_common.hpp"

namespace fgvc {

template <int K>
__global__ __launch_bounds__(512, 2) void pair_topk_kernel_v3(PairParams p) {
  constexpr int C = 256;
  constexpr int LDK = C + 4;
  constexpr int BUF = 32 * LDK;
  constexpr int RING = 2 * 2 * BUF;            // [half][buf]
  constexpr int PB = 2 * K * 64;               // published top-K of one M wave: [K scores | K indices][lane]
  __shared__ __attribute__((aligned(16))) float smem[RING + 4 * PB + 8];
  float* pub = smem + RING;
  int* tile_blk = reinterpret_cast<int*>(smem + RING + 4 * PB);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = wave & 3;
  const int half = w >> 1;
  const int n = lane & 31, hi = lane >> 5;

  const int4 pr = p.pairs[blockIdx.y];
  const int qf = pr.x, kf = pr.y;
  const bool masked = (pr.z & FGVC_PAIR_MASKED) != 0;
  const int reach_y = masked ? p.reach_y : FGVC_NO_LIMIT;
  const int reach_x = masked ? p.reach_x : FGVC_NO_LIMIT;

  const int tile = xcd_remap(blockIdx.x, p.n_ty * p.n_tx);
  const int ty = tile / p.n_tx, tx = tile - ty * p.n_tx;
  HalfCursor it;
  it.r2max = masked ? p.r2max : FGVC_NO_LIMIT;
  it.ry = masked ? p.ry : FGVC_NO_LIMIT;
  it.rx = masked ? p.rx : FGVC_NO_LIMIT;
  it.TY0 = ty * (2 * QBH);
  it.TX0 = tx * (2 * QBW);
  const int QY0 = it.TY0 + (w & 1) * QBH, QX0 = it.TX0 + half * QBW;
  const int qy = QY0 + (n >> 3), qx = QX0 + (n & 7);
  const bool q_valid = qy < p.Hq && qx < p.Wq;
  it.by_lo = imax(0, it.TY0 - imin(reach_y, it.TY0)) / QBH;
  it.by_hi = imin(p.Hk - 1, it.TY0 + 2 * QBH - 1 + imin(reach_y, p.Hk)) / QBH;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int hx0 = it.TX0 + h * QBW;
    it.bxl[h] = imax(0, hx0 - imin(reach_x, hx0)) / QBW;
    it.bxh[h] = imin(p.Wk - 1, hx0 + QBW - 1 + imin(reach_x, p.Wk)) / QBW;
  }
  float* ring = smem + half * (2 * BUF);
  // cursors of BOTH halves in every wave (identical step count = identical barrier count)
  int by0 = it.by_lo, bx0 = it.bxl[0], by1 = it.by_lo, bx1 = it.bxl[1];
  it.seek(0, by0, bx0);
  it.seek(1, by1, bx1);
  int tail = 2;                                 // pipeline drain: sort (+1 step) and merge (+1 step)

  if (wave < 4) {
    // =============================== M role ===============================
    float qreg[C / 2];
    {
      const int qyc = imin(qy, p.Hq - 1), qxc = imin(qx, p.Wq - 1);
      const float* qp = p.qfeat + ((size_t)qf * p.Hq * p.Wq + (size_t)qyc * p.Wq + qxc) * C + 4 * hi;
#pragma unroll
      for (int j = 0; j < C / 8; ++j) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(qp + 8 * j);
        qreg[4 * j + 0] = t.x; qreg[4 * j + 1] = t.y; qreg[4 * j + 2] = t.z; qreg[4 * j + 3] = t.w;
      }
    }
    float cs[16];                               // candidates of the PREVIOUS tile (masked scores) ...
    int ci[16];                                 // ... and their key pixel indices
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      cs[r] = -INFINITY;
      ci[r] = IDX_EMPTY;
    }
    bool have_prev = false;                     // wave-uniform
    if (lane == 0) tile_blk[w] = -1;
    __syncthreads();
    int step = 0;
    while (true) {
      const bool active = it.valid(by0) || it.valid(by1);
      if (!active) {
        if (tail == 0) break;
        --tail;
      }
      const int buf = step & 1;
      __syncthreads();                             // barrier 1: S has copied the previously published list
      const int cby = half ? by1 : by0, cbx = half ? bx1 : bx0;
      const int ky0 = cby * QBH, kx0 = cbx * QBW;
      const bool comp = it.valid(cby) && it.reach(QY0, QX0, ky0, kx0);
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (comp) {
        const float* ka = &ring[buf * BUF + n * LDK + 4 * hi];
        constexpr int G = 4, NG = 8;               // 8 groups of 4 ds_read_b128 = 16 MFMAs
        f32x4 af[2][G];
#pragma unroll
        for (int i = 0; i < G; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(ka + 8 * i);
        __builtin_amdgcn_sched_barrier(0);
#define FGVC_V3_GROUP(g, PIECE)                                                                        \
        {                                                                                              \
          if (g + 1 < NG) {                                                                            \
            _Pragma("unroll") for (int i = 0; i < G; ++i)                                              \
                af[(g + 1) & 1][i] = *reinterpret_cast<const f32x4*>(ka + 8 * ((g + 1) * G + i));     \
          }                                                                                            \
          __builtin_amdgcn_sched_barrier(0);                                                           \
          _Pragma("unroll") for (int i = 0; i < G; ++i) {                                              \
            const f32x4 a = af[g & 1][i];                                                              \
            const int j = g * G + i;                                                                   \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[4 * j + 0], acc, 0, 0, 0);            \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[4 * j + 1], acc, 0, 0, 0);            \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[4 * j + 2], acc, 0, 0, 0);            \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[4 * j + 3], acc, 0, 0, 0);            \
          }                                                                                            \
          PIECE(FGVC_V3_X) /* piece g of the previous tile's sorting network, in the MFMA shadow */   \
          __builtin_amdgcn_sched_barrier(0);                                                           \
        }
#define FGVC_V3_X(I, J) FGVC_CSWAP(c, I, J)
        FGVC_V3_GROUP(0, FGVC_SORTNET_16_P0)
        FGVC_V3_GROUP(1, FGVC_SORTNET_16_P1)
        FGVC_V3_GROUP(2, FGVC_SORTNET_16_P2)
        FGVC_V3_GROUP(3, FGVC_SORTNET_16_P3)
        FGVC_V3_GROUP(4, FGVC_SORTNET_16_P4)
        FGVC_V3_GROUP(5, FGVC_SORTNET_16_P5)
        FGVC_V3_GROUP(6, FGVC_SORTNET_16_P6)
        FGVC_V3_GROUP(7, FGVC_SORTNET_16_P7)
#undef FGVC_V3_GROUP
      } else if (have_prev) {
        FGVC_SORTNET_16(FGVC_V3_X)                  // nothing to compute this step: just finish the sort
      }
#undef FGVC_V3_X
      // publish the top-K of the previous tile (now sorted)
      if (have_prev) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
          pub[w * PB + j * 64 + lane] = cs[j];
          reinterpret_cast<int*>(pub)[w * PB + (K + j) * 64 + lane] = ci[j];
        }
      }
      if (lane == 0) tile_blk[w] = have_prev ? 1 : -1;
      // the tile just computed becomes the candidate set of the next step (mask predicate applied here)
      if (comp) {
        const int dy0 = ky0 - qy, dx0 = kx0 + 4 * hi - qx;
        const int id0 = ky0 * p.Wk + kx0 + 4 * hi;
        const bool interior = ky0 + QBH <= p.Hk && kx0 + QBW <= p.Wk;
        const bool circle_only = it.ry >= FGVC_NO_LIMIT && it.rx >= FGVC_NO_LIMIT;
        if (interior && circle_only) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dy = dy0 + (r >> 2), dx = dx0 + (r & 3);
            cs[r] = (dy * dy + dx * dx <= it.r2max) ? acc[r] : -INFINITY;
            ci[r] = id0 + (r >> 2) * p.Wk + (r & 3);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dy = dy0 + (r >> 2), dx = dx0 + (r & 3);
            const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
            const bool ok = ky0 + (r >> 2) < p.Hk && kx0 + 4 * hi + (r & 3) < p.Wk &&
                            dy * dy + dx * dx <= it.r2max && ady <= it.ry && adx <= it.rx;
            cs[r] = ok ? acc[r] : -INFINITY;
            ci[r] = id0 + (r >> 2) * p.Wk + (r & 3);
          }
        }
      }
      have_prev = comp;
      if (it.valid(by0)) it.advance(0, by0, bx0);  // cursor update off the MFMA critical path
      if (it.valid(by1)) it.advance(1, by1, bx1);
      __syncthreads();                             // barrier 2
      ++step;
    }
  } else {
    // =============================== S role ===============================
    const float* kbase = p.kfeat + (size_t)kf * p.Hk * p.Wk * C;
    const int sw = wave - 4 - 2 * half;
    auto stage_load = [&](int sby, int sbx, int buf) {
      const int ky0 = sby * QBH, kx0 = sbx * QBW;
      if (ky0 + QBH <= p.Hk && kx0 + QBW <= p.Wk) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = sw * 16 + i;
          const float* src = kbase + ((size_t)(ky0 + (row >> 3)) * p.Wk + kx0) * C + (row & 7) * C + 4 * lane;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)&ring[buf * BUF + row * LDK],
                                           16, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = sw * 16 + i;
          const int ky = imin(ky0 + (row >> 3), p.Hk - 1), kx = imin(kx0 + (row & 7), p.Wk - 1);
          const float* src = kbase + ((size_t)ky * p.Wk + kx) * C + 4 * lane;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)&ring[buf * BUF + row * LDK],
                                           16, 0, 0);
        }
      }
    };
    TopKF<K> top;
    top.init();
    {
      const int sby = half ? by1 : by0, sbx = half ? bx1 : bx0;
      if (it.valid(sby)) stage_load(sby, sbx, 0);
    }
    __syncthreads();
    int step = 0;
    while (true) {
      const bool active = it.valid(by0) || it.valid(by1);
      if (!active) {
        if (tail == 0) break;
        --tail;
      }
      if (it.valid(by0)) it.advance(0, by0, bx0);  // cursors now point at the block of step+1
      if (it.valid(by1)) it.advance(1, by1, bx1);
      const int buf = step & 1;
      float ns[K];
      int ni[K];
      const bool have = tile_blk[w] >= 0;          // written by M before barrier 2 of the previous step
      if (have) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
          ns[j] = pub[w * PB + j * 64 + lane];
          ni[j] = reinterpret_cast<const int*>(pub)[w * PB + (K + j) * 64 + lane];
        }
      }
      __syncthreads();                             // barrier 1
      const int nby = half ? by1 : by0, nbx_ = half ? bx1 : bx0;
      if (it.valid(nby)) stage_load(nby, nbx_, buf ^ 1);
      if (have) {
        // top-K of (sorted candidates) U (sorted list): max(cand[K-1-j], list[j]) is that set, bitonic; re-sort
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const bool bb = ns[K - 1 - j] > top.v[j];
          top.v[j] = bb ? ns[K - 1 - j] : top.v[j];
          top.ix[j] = bb ? ni[K - 1 - j] : top.ix[j];
        }
        float (&ls)[K] = top.v;
        int (&li)[K] = top.ix;
#define X(I, J) FGVC_CSWAP(l, I, J)
        if constexpr (K == 10) { FGVC_SORTNET_10(X) }
        else if constexpr (K == 5) { FGVC_SORTNET_5(X) }
#undef X
      }
      __syncthreads();                             // barrier 2 (drains the DMA: vmcnt(0))
      ++step;
    }
    // the two lanes (n,0) and (n,1) hold disjoint candidates of the same query: canonical merge
    TopK<K> fin;
    fin.init();
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float pv = __shfl_xor(top.v[j], 32);
      const int pi = __shfl_xor(top.ix[j], 32);
      if (top.ix[j] != IDX_EMPTY && fin.accepts(top.v[j], top.ix[j])) fin.insert(top.v[j], top.ix[j]);
      if (pi != IDX_EMPTY && fin.accepts(pv, pi)) fin.insert(pv, pi);
    }
    if (hi == 0 && q_valid) {
      const size_t o = ((size_t)blockIdx.y * p.Hq * p.Wq + (size_t)qy * p.Wq + qx) * p.kout;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (j < p.kout) {
          const bool e = fin.ix[j] == IDX_EMPTY;
          p.idx_out[o + j] = e ? -1 : fin.ix[j];
          p.score_out[o + j] = e ? -INFINITY : fin.v[j];
        }
      }
    }
  }
}

int pair_topk_v3_launch(const PairParams& p, int n_pairs, int topk, hipStream_t s) {
  dim3 grid(p.n_ty * p.n_tx, n_pairs);
  if (topk <= 5)
    pair_topk_kernel_v3<5><<<grid, 512, 0, s>>>(p);
  else
    pair_topk_kernel_v3<10><<<grid, 512, 0, s>>>(p);
  FGVC_CHECK_LAUNCH("fgvc_pair_topk_f32(v3)");
  return FGVC_OK;
}

}  // namespace fgvc
