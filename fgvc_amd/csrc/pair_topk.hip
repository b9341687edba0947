// fgvc_pair_topk_f32: windowed query x key correlation with a running top-k, never writing the
// (HWk x HWq) volume.  Replaces local_attention.py:321-356 of the reference for one key frame.
// The EXACT-f32 form of the operator: any channel count in {32, 64, 128, 256}, top-k up to 16, analytic or dense masks, features
// normalised or not.  The shipped configurations (C = 256, normalised, top-k <= 10) run on fgvc_pair_topk_f16x3 (pair_topk_v5.hip);
// this kernel is the fallback for everything else and the reference the 16-bit form is held to in the tests.
//
// Work decomposition (gfx950, wave64, f32 MFMA):
//   grid.y = (query frame, key frame) pair, grid.x = query tile of 8 rows x 16 cols (XCD-remapped).
//   A workgroup is 8 waves in two roles.  M wave w (0-3) owns one 4x8 block of 32 query pixels and keeps its 32
//   query vectors resident in VGPRs as the B operand of v_mfma_f32_32x32x2_f32 (C/2 registers per
//   lane: lane (n, hi) holds channels {8j+4hi+s}).  Key pixels are visited in aligned 4x8 blocks;
//   a block's 32 key vectors (32 x C f32, channels-last = 32 coalesced rows of C*4 bytes) are
//   staged HBM -> registers -> LDS by the S waves (4-7; double buffered per 8x8 half, padded rows so the
//   ds_read_b128 A-operand reads are bank-conflict free) and consumed by every M wave whose query
//   block can reach it under the mask predicate; blocks no wave can reach are never loaded.
//   One 32x32 score tile = C/2 MFMAs in four interleaved chains (f32 throughout).  The M wave hands the tile to its S wave
//   through the LDS; the S wave applies the mask and keeps the running top-k as a private per-lane sorted list (score desc,
//   index asc) -- no cross-lane traffic until the two lanes that share a query merge their lists with one __shfl_xor(32)
//   sweep at the end.
#include "pair_common.hpp"

namespace fgvc {

template <int C, int K>
__global__ __launch_bounds__(512, 2) void pair_topk_kernel_v2(PairParams p) {
  constexpr int LDK = C + 4;
  constexpr int BUF = 32 * LDK;                // one key block (floats)
  constexpr int NLD = C / 16;                  // float4 loads per S thread per key block (128 loader threads per half)
  constexpr int RING = 2 * 2 * BUF;            // [half][buf]
  constexpr int SB = 16 * 64;                  // one score tile (floats), layout [r][lane]
  __shared__ __attribute__((aligned(16))) float smem[RING + 4 * SB + 8];
  float* sbuf = smem + RING;
  int* tile_blk = reinterpret_cast<int*>(smem + RING + 4 * SB);   // [4] block id of the tile in sbuf[w], -1 = none

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = wave & 3;                      // M wave w <-> S wave 4+w
  const int half = w >> 1;                     // 0: left 8x8, 1: right 8x8
  const int n = lane & 31, hi = lane >> 5;

  const int4 pr = p.pairs[blockIdx.y];
  const int qf = pr.x, kf = pr.y;
  const bool masked = (pr.z & FGVC_PAIR_MASKED) != 0;
  const int reach_y = masked ? p.reach_y : FGVC_NO_LIMIT;
  const int reach_x = masked ? p.reach_x : FGVC_NO_LIMIT;

  const int tile = xcd_remap(blockIdx.x, p.n_ty * p.n_tx);
  const int ty = tile / p.n_tx, tx = tile - ty * p.n_tx;
  HalfIter it;
  it.r2max = masked ? p.r2max : FGVC_NO_LIMIT;
  it.ry = masked ? p.ry : FGVC_NO_LIMIT;
  it.rx = masked ? p.rx : FGVC_NO_LIMIT;
  it.TY0 = ty * (2 * QBH);
  it.TX0 = tx * (2 * QBW);
  const int QY0 = it.TY0 + (w & 1) * QBH, QX0 = it.TX0 + half * QBW;     // vertical stacking inside a half
  const int qy = QY0 + (n >> 3), qx = QX0 + (n & 7);
  const bool q_valid = qy < p.Hq && qx < p.Wq;
  it.by_lo = imax(0, it.TY0 - imin(reach_y, it.TY0)) / QBH;
  const int by_hi = imin(p.Hk - 1, it.TY0 + 2 * QBH - 1 + imin(reach_y, p.Hk)) / QBH;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int hx0 = it.TX0 + h * QBW;
    it.bxl[h] = imax(0, hx0 - imin(reach_x, hx0)) / QBW;
    const int bxh = imin(p.Wk - 1, hx0 + QBW - 1 + imin(reach_x, p.Wk)) / QBW;
    it.nbx[h] = bxh - it.bxl[h] + 1;
    it.nb[h] = (by_hi - it.by_lo + 1) * it.nbx[h];
  }
  float* ring = smem + half * (2 * BUF);

  // Every wave tracks BOTH halves' iterators so the step count (= barrier count) is identical across
  // the workgroup.  The two roles run separate loops so that their register sets do not add up.
  int cur0 = it.next_block(0, -1), cur1 = it.next_block(1, -1);

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
    if (lane == 0) tile_blk[w] = -1;
    __syncthreads();                               // prologue barrier (S staged block 0)
    int step = 0;
    bool pending = false;
    while (cur0 < it.nb[0] || cur1 < it.nb[1] || pending) {
      const int nxt0 = cur0 < it.nb[0] ? it.next_block(0, cur0) : it.nb[0];
      const int nxt1 = cur1 < it.nb[1] ? it.next_block(1, cur1) : it.nb[1];
      const int buf = step & 1;
      __syncthreads();                             // barrier 1: S has copied the previous tile out of sbuf
      const int b = half ? cur1 : cur0;
      bool did = false;
      if (b < it.nb[half]) {
        const int ky0 = it.blk_y(half, b), kx0 = it.blk_x(half, b);
        if (it.reach(QY0, QX0, ky0, kx0) && !(p.debug & 2)) {
          // FOUR independent accumulator chains (channel c = 8j+4hi+s goes to chain s): a dependent
          // v_mfma_f32_32x32x2_f32 chain retires only one MFMA per ~200 cycles per wave (measured), the
          // pipe issues one per 64, so a single chain leaves the matrix core ~70 % idle.
          // A fragments are double buffered in registers, G ds_read_b128 (= 4G MFMAs) per stage; the
          // sched_barrier(0) fences keep hipcc from sinking the reads down to their use (which exposed
          // the LDS latency every 8 MFMAs in v1) -- the s_waitcnt lgkmcnt are still compiler-counted.
          f32x16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          f32x16 acc1 = acc0, acc2 = acc0, acc3 = acc0;
          const float* ka = &ring[buf * BUF + n * LDK + 4 * hi];
          constexpr int G = (C / 8 >= 4) ? 4 : C / 8;
          constexpr int NG = (C / 8) / G;
          f32x4 af[2][G];
#pragma unroll
          for (int i = 0; i < G; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(ka + 8 * i);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
#pragma unroll
              for (int i = 0; i < G; ++i)
                af[(g + 1) & 1][i] = *reinterpret_cast<const f32x4*>(ka + 8 * ((g + 1) * G + i));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < G; ++i) {
              const f32x4 a = af[g & 1][i];
              const int j = g * G + i;
              acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[4 * j + 0], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[4 * j + 1], acc1, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[4 * j + 2], acc2, 0, 0, 0);
              acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[4 * j + 3], acc3, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          acc0 += acc1;
          acc2 += acc3;
          acc0 += acc2;
          const f32x16& acc = acc0;
          // The mask predicate is applied HERE (the MFMA wave has idle VALU issue slots; the selection
          // wave is the busier one): rejected candidates are published as -inf.
          {
            const int dy0 = ky0 - qy, dx0 = kx0 + 4 * hi - qx;
            const bool interior = ky0 + QBH <= p.Hk && kx0 + QBW <= p.Wk;       // wave-uniform
            const bool circle_only = it.ry >= FGVC_NO_LIMIT && it.rx >= FGVC_NO_LIMIT;
            if (interior && circle_only) {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int dy = dy0 + (r >> 2), dx = dx0 + (r & 3);
                sbuf[w * SB + r * 64 + lane] = (dy * dy + dx * dx <= it.r2max) ? acc[r] : -INFINITY;
              }
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int dy = dy0 + (r >> 2), dx = dx0 + (r & 3);
                const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
                const bool ok = ky0 + (r >> 2) < p.Hk && kx0 + 4 * hi + (r & 3) < p.Wk &&
                                dy * dy + dx * dx <= it.r2max && ady <= it.ry && adx <= it.rx;
                sbuf[w * SB + r * 64 + lane] = ok ? acc[r] : -INFINITY;
              }
            }
          }
          did = true;
        }
      }
      if (lane == 0) tile_blk[w] = did ? b : -1;
      __syncthreads();                             // barrier 2
      pending = (cur0 < it.nb[0]) || (cur1 < it.nb[1]);
      cur0 = nxt0;
      cur1 = nxt1;
      ++step;
    }
  } else {
    // =============================== S role ===============================
    const float* kbase = p.kfeat + (size_t)kf * p.Hk * p.Wk * C;
    const int lt = (wave - 4 - 2 * half) * 64 + lane;     // loader thread id within the half: 0..127
    // C == 256: one key row is exactly one 1-KiB LDS-DMA wave instruction (global_load_lds_dwordx4:
    // per-lane global source, wave-uniform LDS row base + lane*16) -> no staging VGPRs at all; the
    // padded row stride is legal because no wave instruction crosses a row.  Narrower rows are staged
    // through registers (few of them are needed there).
    constexpr bool DMA = (C == 256);
    constexpr int NST = DMA ? 1 : NLD;
    f32x4 stage[NST];
    const int sw = wave - 4 - 2 * half;                   // 0/1: which S wave of the half
    auto stage_load = [&](int b, int buf) {
      const int ky0 = it.blk_y(half, b), kx0 = it.blk_x(half, b);
      if constexpr (DMA) {
        if (ky0 + QBH <= p.Hk && kx0 + QBW <= p.Wk) {
          // interior block: a block row is 8 consecutive pixels = 8 KiB contiguous; two row bases per S wave
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
      } else {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
          const int f = lt + 128 * i;
          const int row = f / (C / 4), c4 = f % (C / 4);
          const int ky = imin(ky0 + (row >> 3), p.Hk - 1), kx = imin(kx0 + (row & 7), p.Wk - 1);
          stage[i] = *reinterpret_cast<const f32x4*>(kbase + ((size_t)ky * p.Wk + kx) * C + 4 * c4);
        }
      }
    };
    auto stage_store = [&](int buf) {
      if constexpr (!DMA) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
          const int f = lt + 128 * i;
          const int row = f / (C / 4), c4 = f % (C / 4);
          *reinterpret_cast<f32x4*>(&ring[buf * BUF + row * LDK + 4 * c4]) = stage[i];
        }
      }
    };
    TopKF<K> top;
    top.init();
    {
      const int b0 = half ? cur1 : cur0;
      if (b0 < it.nb[half]) {
        stage_load(b0, 0);
        stage_store(0);
      }
    }
    __syncthreads();                               // prologue barrier
    int step = 0;
    bool pending = false;
    while (cur0 < it.nb[0] || cur1 < it.nb[1] || pending) {
      const int nxt0 = cur0 < it.nb[0] ? it.next_block(0, cur0) : it.nb[0];
      const int nxt1 = cur1 < it.nb[1] ? it.next_block(1, cur1) : it.nb[1];
      const int buf = step & 1;
      float tilev[16];
      const int sel_blk = tile_blk[w];             // written by M before barrier 2 of the previous step
      if (sel_blk >= 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tilev[r] = sbuf[w * SB + r * 64 + lane];
      }
      __syncthreads();                             // barrier 1
      const int bn = half ? nxt1 : nxt0;
      if (bn < it.nb[half] && !(p.debug & 4)) stage_load(bn, buf ^ 1);   // in flight while the selection below runs
      if (sel_blk >= 0 && !(p.debug & 1)) {
        const int ky0 = it.blk_y(half, sel_blk), kx0 = it.blk_x(half, sel_blk);
        const bool use_dense = masked && p.dense_mask != nullptr;          // wave-uniform, rare
        float cs[16];
        int ci[16];
        const int id0 = ky0 * p.Wk + kx0 + 4 * hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          cs[r] = tilev[r];                                    // already -inf where the predicate rejects
          ci[r] = id0 + (r >> 2) * p.Wk + (r & 3);
        }
        if (use_dense) {   // arbitrary user mask: one byte gather per candidate that could still enter
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (q_valid && top.accepts(cs[r])) {
              if (p.dense_mask[(size_t)ci[r] * ((size_t)p.Hq * p.Wq) + (size_t)qy * p.Wq + qx] == 0) cs[r] = -INFINITY;
            }
          }
        }
        if constexpr (K == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) top.insert_if(cs[r] > top.v[0], cs[r], ci[r]);
        } else {
          top.merge_tile(cs, ci);
        }
      }
      if (bn < it.nb[half] && !(p.debug & 4)) stage_store(buf ^ 1);
      __syncthreads();                             // barrier 2
      pending = (cur0 < it.nb[0]) || (cur1 < it.nb[1]);
      cur0 = nxt0;
      cur1 = nxt1;
      ++step;
    }
    // The two lanes (n,0) and (n,1) hold disjoint candidates of the same query: merge them through the
    // canonical comparator (score desc, index asc), which also orders exact ties inside each lane's list.
    {
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
}

static int g_pair_debug = 0;
void set_pair_debug(int v) { g_pair_debug = v; }

template <int C, int K>
static int launch_pair(const PairParams& p, int n_pairs, hipStream_t s) {
  dim3 grid(p.n_ty * p.n_tx, n_pairs);
  pair_topk_kernel_v2<C, K><<<grid, 512, 0, s>>>(p);
  FGVC_CHECK_LAUNCH("fgvc_pair_topk_f32");
  return FGVC_OK;
}

template <int C>
static int dispatch_k(const PairParams& p, int n_pairs, int k, hipStream_t s) {
  // the list length is a compile-time register array; four instantiations cover topk 1..16
  if (k <= 1) return launch_pair<C, 1>(p, n_pairs, s);
  if (k <= 5) return launch_pair<C, 5>(p, n_pairs, s);
  if (k <= 10) return launch_pair<C, 10>(p, n_pairs, s);
  return launch_pair<C, 16>(p, n_pairs, s);
}

int pair_topk_launch(const float* qfeat, const float* kfeat, const int32_t* pairs, int n_pairs, int C,
                     int Hq, int Wq, int Hk, int Wk, int r2max, int ry, int rx, int topk,
                     const uint8_t* dense_mask, int32_t* idx_out, float* score_out, hipStream_t s) {
  PairParams p;
  p.qfeat = qfeat; p.kfeat = kfeat; p.pairs = reinterpret_cast<const int4*>(pairs);
  p.Hq = Hq; p.Wq = Wq; p.Hk = Hk; p.Wk = Wk;
  p.r2max = r2max; p.ry = ry; p.rx = rx;
  int rr = 0;  // floor(sqrt(r2max)) in integers
  while (rr < 46340 && (long long)(rr + 1) * (rr + 1) <= (long long)r2max) ++rr;
  p.reach_y = imin(ry, rr); p.reach_x = imin(rx, rr);
  p.kout = topk;
  p.n_ty = cdiv(Hq, 2 * QBH); p.n_tx = cdiv(Wq, 2 * QBW);
  p.idx_out = idx_out; p.score_out = score_out;
  p.dense_mask = dense_mask;
  p.debug = g_pair_debug;
  switch (C) {
    case 32: return dispatch_k<32>(p, n_pairs, topk, s);
    case 64: return dispatch_k<64>(p, n_pairs, topk, s);
    case 128: return dispatch_k<128>(p, n_pairs, topk, s);
    case 256: return dispatch_k<256>(p, n_pairs, topk, s);
    default:
      set_error("fgvc_pair_topk_f32: C=%d unsupported (32, 64, 128 or 256)", C);
      return FGVC_ERR_UNSUPPORTED;
  }
}

}  // namespace fgvc
