// Declarations shared by the fgvc_pair_topk_f32 kernel variants.
#pragma once
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
  int debug;                  // ablation switches for profiling (results are WRONG when non-zero):
                              // 1 = no selection, 2 = no MFMA, 4 = no key staging, 8 = no geometry (accept all in-bounds)
  int32_t* idx_out;
  float* score_out;
};

// Geometry shared by both roles of the v2 kernel (all members wave-uniform).
struct HalfIter {
  int by_lo, bxl[2], nbx[2], nb[2];
  int TY0, TX0, r2max, ry, rx;

  __device__ __forceinline__ bool reach(int wy0, int wx0, int ky0, int kx0) const {
    const int dy = imax(0, imax(ky0 - (wy0 + QBH - 1), wy0 - (ky0 + QBH - 1)));
    const int dx = imax(0, imax(kx0 - (wx0 + QBW - 1), wx0 - (kx0 + QBW - 1)));
    return dy * dy + dx * dx <= r2max && dy <= ry && dx <= rx;
  }
  __device__ __forceinline__ int blk_y(int h, int b) const { return (by_lo + b / nbx[h]) * QBH; }
  __device__ __forceinline__ int blk_x(int h, int b) const { return (bxl[h] + b % nbx[h]) * QBW; }
  __device__ __forceinline__ bool half_need(int h, int b) const {
    const int ky0 = blk_y(h, b), kx0 = blk_x(h, b), hx0 = TX0 + h * QBW;
    return reach(TY0, hx0, ky0, kx0) || reach(TY0 + QBH, hx0, ky0, kx0);
  }
  __device__ __forceinline__ int next_block(int h, int b) const {
    ++b;
    while (b < nb[h] && !half_need(h, b)) ++b;
    return b;
  }
};

// Division-free variant used by v3: a cursor (by, bx) in block units per half; `by > by_hi` = exhausted.
// (The linear-index iterator above costs four scalar integer divisions per wave per step, on the critical path
// between the two barriers of a step: ~1 us per step measured.)
struct HalfCursor {
  int by_lo, by_hi, bxl[2], bxh[2];
  int TY0, TX0, r2max, ry, rx;

  __device__ __forceinline__ bool reach(int wy0, int wx0, int ky0, int kx0) const {
    const int dy = imax(0, imax(ky0 - (wy0 + QBH - 1), wy0 - (ky0 + QBH - 1)));
    const int dx = imax(0, imax(kx0 - (wx0 + QBW - 1), wx0 - (kx0 + QBW - 1)));
    return dy * dy + dx * dx <= r2max && dy <= ry && dx <= rx;
  }
  __device__ __forceinline__ bool half_need(int h, int by, int bx) const {
    const int ky0 = by * QBH, kx0 = bx * QBW, hx0 = TX0 + h * QBW;
    return reach(TY0, hx0, ky0, kx0) || reach(TY0 + QBH, hx0, ky0, kx0);
  }
  __device__ __forceinline__ bool valid(int by) const { return by <= by_hi; }
  // first needed block at or after (by, bx)
  __device__ __forceinline__ void seek(int h, int& by, int& bx) const {
    while (by <= by_hi && !half_need(h, by, bx)) {
      if (++bx > bxh[h]) {
        bx = bxl[h];
        ++by;
      }
    }
  }
  __device__ __forceinline__ void advance(int h, int& by, int& bx) const {
    if (++bx > bxh[h]) {
      bx = bxl[h];
      ++by;
    }
    seek(h, by, bx);
  }
};

}  // namespace fgvc
