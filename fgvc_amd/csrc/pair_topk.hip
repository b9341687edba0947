// fgvc_pair_topk_f32: windowed query x key correlation with a running top-k, never writing the
// (HWk x HWq) volume.  Replaces local_attention.py:321-356 of the reference for one key frame.
//
// Work decomposition (gfx950, wave64, f32 MFMA):
//   grid.y = (query frame, key frame) pair, grid.x = query tile of 8 rows x 16 cols (XCD-remapped).
//   A workgroup is 4 waves; wave w owns one 4x8 block of 32 query pixels.  The wave keeps its 32
//   query vectors resident in VGPRs as the B operand of v_mfma_f32_32x32x2_f32 (C/2 registers per
//   lane: lane (n, hi) holds channels {8j+4hi+s}).  Key pixels are visited in aligned 4x8 blocks;
//   a block's 32 key vectors (32 x C f32, channels-last = 32 coalesced rows of C*4 bytes) are
//   staged HBM -> registers -> LDS once per workgroup (double buffered, padded rows so the
//   ds_read_b128 A-operand reads are bank-conflict free) and consumed by every wave whose query
//   block can reach it under the mask predicate; blocks no wave can reach are never loaded.
//   One 32x32 score tile = C/2 chained MFMAs (exact f32: a k-ordered fma chain).  Scores land with
//   the query on the lane and 16 keys in the accumulator registers, so the running top-k is a
//   private per-lane sorted list (score desc, index asc) -- no cross-lane traffic until the two
//   lanes that share a query merge their lists with one __shfl_xor(32) sweep at the end.
#include "common.hpp"

namespace fgvc {

constexpr int QBH = 4, QBW = 8;  // pixel block = 4 rows x 8 cols = 32 = one MFMA tile edge

struct PairParams {
  const float* qfeat;
  const float* kfeat;
  const int4* pairs;
  int Hq, Wq, Hk, Wk;
  int r2max, ry, rx;  // mask predicate (FGVC_NO_LIMIT = off)
  int reach_y, reach_x;  // largest |dy|, |dx| the predicate admits (host-computed)
  int n_ty, n_tx;
  int kout;           // entries written per query (<= K); rows of idx_out/score_out have this stride
  const uint8_t* dense_mask;  // optional [HWk][HWq] bool: arbitrary user mask (full-frame traversal)
  int32_t* idx_out;
  float* score_out;
};

template <int C, int K>
__global__ __launch_bounds__(256, 2) void pair_topk_kernel(PairParams p) {
  constexpr int LDK = C + 4;               // padded LDS row (floats): stride 4*odd -> conflict-free b128
  constexpr int BUF = 32 * LDK;            // one key block
  constexpr int NLD = C / 32;              // float4 global loads per thread per key block
  __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hi = lane >> 5;

  const int4 pr = p.pairs[blockIdx.y];
  const int qf = pr.x, kf = pr.y;
  const bool masked = (pr.z & FGVC_PAIR_MASKED) != 0;
  const int r2max = masked ? p.r2max : FGVC_NO_LIMIT;
  const int ry = masked ? p.ry : FGVC_NO_LIMIT;
  const int rx = masked ? p.rx : FGVC_NO_LIMIT;

  const int tile = xcd_remap(blockIdx.x, p.n_ty * p.n_tx);
  const int ty = tile / p.n_tx, tx = tile - ty * p.n_tx;
  const int TY0 = ty * (2 * QBH), TX0 = tx * (2 * QBW);     // workgroup tile origin
  const int QY0 = TY0 + (wave >> 1) * QBH, QX0 = TX0 + (wave & 1) * QBW;  // this wave's block
  const int qy = QY0 + (n >> 3), qx = QX0 + (n & 7);
  const bool q_valid = qy < p.Hq && qx < p.Wq;

  // ---- query vectors -> registers (B operand), clamped address for out-of-image lanes
  float qreg[C / 2];
  {
    const int qyc = imin(qy, p.Hq - 1), qxc = imin(qx, p.Wq - 1);
    const float* qp = p.qfeat + ((size_t)qf * p.Hq * p.Wq + (size_t)qyc * p.Wq + qxc) * C + 4 * hi;
#pragma unroll
    for (int j = 0; j < C / 8; ++j) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(qp + 8 * j);
      qreg[4 * j + 0] = t.x;
      qreg[4 * j + 1] = t.y;
      qreg[4 * j + 2] = t.z;
      qreg[4 * j + 3] = t.w;
    }
  }

  // ---- key block range that the workgroup tile can reach
  const int reach_y = masked ? p.reach_y : FGVC_NO_LIMIT;
  const int reach_x = masked ? p.reach_x : FGVC_NO_LIMIT;
  const int by_lo = imax(0, TY0 - imin(reach_y, TY0)) / QBH;
  const int by_hi = imin(p.Hk - 1, TY0 + 2 * QBH - 1 + imin(reach_y, p.Hk)) / QBH;
  const int bx_lo = imax(0, TX0 - imin(reach_x, TX0)) / QBW;
  const int bx_hi = imin(p.Wk - 1, TX0 + 2 * QBW - 1 + imin(reach_x, p.Wk)) / QBW;
  const int nbx = bx_hi - bx_lo + 1;
  const int nb = (by_hi - by_lo + 1) * nbx;

  // can a 4x8 query block at (wy0,wx0) reach key block (ky0,kx0)?  (all operands wave-uniform)
  auto reach = [&](int wy0, int wx0, int ky0, int kx0) -> bool {
    const int dy = imax(0, imax(ky0 - (wy0 + QBH - 1), wy0 - (ky0 + QBH - 1)));
    const int dx = imax(0, imax(kx0 - (wx0 + QBW - 1), wx0 - (kx0 + QBW - 1)));
    return dy * dy + dx * dx <= r2max && dy <= ry && dx <= rx;
  };
  auto wg_need = [&](int b) -> bool {
    const int ky0 = (by_lo + b / nbx) * QBH, kx0 = (bx_lo + b % nbx) * QBW;
    return reach(TY0, TX0, ky0, kx0) || reach(TY0, TX0 + QBW, ky0, kx0) ||
           reach(TY0 + QBH, TX0, ky0, kx0) || reach(TY0 + QBH, TX0 + QBW, ky0, kx0);
  };
  auto next_block = [&](int b) -> int {
    ++b;
    while (b < nb && !wg_need(b)) ++b;
    return b;
  };

  const float* kbase = p.kfeat + (size_t)kf * p.Hk * p.Wk * C;
  f32x4 stage[NLD];
  auto stage_load = [&](int b) {
    const int ky0 = (by_lo + b / nbx) * QBH, kx0 = (bx_lo + b % nbx) * QBW;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + 256 * i;
      const int row = f / (C / 4), c4 = f % (C / 4);
      const int ky = imin(ky0 + (row >> 3), p.Hk - 1), kx = imin(kx0 + (row & 7), p.Wk - 1);
      stage[i] = *reinterpret_cast<const f32x4*>(kbase + ((size_t)ky * p.Wk + kx) * C + 4 * c4);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + 256 * i;
      const int row = f / (C / 4), c4 = f % (C / 4);
      *reinterpret_cast<f32x4*>(&smem[buf * BUF + row * LDK + 4 * c4]) = stage[i];
    }
  };

  TopK<K> top;
  top.init();

  int b = next_block(-1);
  if (b < nb) {
    stage_load(b);
    stage_store(0);
  }
  __syncthreads();
  int buf = 0;
  while (b < nb) {
    const int bn = next_block(b);
    if (bn < nb) stage_load(bn);  // in flight behind the MFMA chain below

    const int ky0 = (by_lo + b / nbx) * QBH, kx0 = (bx_lo + b % nbx) * QBW;
    if (reach(QY0, QX0, ky0, kx0)) {
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      const float* ka = &smem[buf * BUF + n * LDK + 4 * hi];  // A operand: key row (lane&31)
#pragma unroll
      for (int j = 0; j < C / 8; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ka + 8 * j);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[4 * j + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[4 * j + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[4 * j + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[4 * j + 3], acc, 0, 0, 0);
      }
      // acc[r]: key row m = (r&3) + 8*(r>>2) + 4*hi  ->  (ky_l, kx_l) = (r>>2, (r&3)+4*hi); column = query n
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ky = ky0 + (r >> 2), kx = kx0 + (r & 3) + 4 * hi;
        const int dy = ky - qy, dx = kx - qx;
        const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
        bool ok = ky < p.Hk && kx < p.Wk && dy * dy + dx * dx <= r2max && ady <= ry && adx <= rx;
        const float s = acc[r];
        const int id = ky * p.Wk + kx;
        if (ok && top.accepts(s, id)) {
          // arbitrary dense mask (local_attention.py:329-353 with a user tensor): consulted only for
          // candidates that would enter the list, so the byte gather stays off the common path
          if (masked && p.dense_mask != nullptr && q_valid)
            ok = p.dense_mask[(size_t)id * ((size_t)p.Hq * p.Wq) + (size_t)qy * p.Wq + qx] != 0;
          if (ok) top.insert(s, id);
        }
      }
    }
    if (bn < nb) stage_store(buf ^ 1);
    __syncthreads();
    buf ^= 1;
    b = bn;
  }

  // ---- the two lanes (n, 0) and (n, 1) hold disjoint candidates of the same query: merge
  {
    float pv[K];
    int pi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      pv[j] = __shfl_xor(top.v[j], 32);
      pi[j] = __shfl_xor(top.ix[j], 32);
    }
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (pi[j] != IDX_EMPTY && top.accepts(pv[j], pi[j])) top.insert(pv[j], pi[j]);
  }
  if (hi == 0 && q_valid) {
    const size_t o = ((size_t)blockIdx.y * p.Hq * p.Wq + (size_t)qy * p.Wq + qx) * p.kout;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (j < p.kout) {  // the first kout entries of a sorted top-K list ARE the top-kout
        const bool e = top.ix[j] == IDX_EMPTY;
        p.idx_out[o + j] = e ? -1 : top.ix[j];
        p.score_out[o + j] = e ? -INFINITY : top.v[j];
      }
    }
  }
}

template <int C, int K>
static int launch_pair(const PairParams& p, int n_pairs, hipStream_t s) {
  dim3 grid(p.n_ty * p.n_tx, n_pairs);
  pair_topk_kernel<C, K><<<grid, 256, 0, s>>>(p);
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
