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

// ---- the split-operand kernels (pair_topk_v5.hip: f16 x 3 products)
struct PairParamsB {
  const uint16_t* q_hl;   // [frame][pixel][2][256] f16 bit patterns (h part, l part: fgvc_split_f16x2)
  const uint16_t* k_hl;
  const int4* pairs;
  int Hq, Wq, Hk, Wk;
  int r2max, ry, rx;
  int reach_y, reach_x;
  int n_ty, n_tx;
  int kout;
  int debug;              // profiling ablations (results WRONG): 1 = no selection, 2 = no MFMA, 4 = no staging,
                          // 16 = prologue only, 32 = no epilogue, 64 = no main loop, 128 = no s_setprio around the MFMA chain
  int32_t* idx_out;
  float* score_out;
  const int2* groups;     // optional [n_groups] (first pair, count) -- runs of pairs with one query frame and one
                          // mask flag that a workgroup takes in one go (query prologue once, the ring never drains); null: each pair alone
  int rowb;               // fgvc_pair_topk_f16f6 only: bytes from one pixel's row to the next (1024: fgvc_split_f16f6p rows; 2048: the rows of
                          // fgvc_split_f16f6x, whose second KiB carries the pixel's exact f32 channels for fgvc_merge_refine_topk_f32)
};

// block-to-block reach test of the mask predicate (all operands wave-uniform or per-lane, no state)
struct ReachTest {
  int r2max, ry, rx;
  __device__ __forceinline__ bool operator()(int wy0, int wx0, int ky0, int kx0) const {
    const int dy = imax(0, imax(ky0 - (wy0 + QBH - 1), wy0 - (ky0 + QBH - 1)));
    const int dx = imax(0, imax(kx0 - (wx0 + QBW - 1), wx0 - (kx0 + QBW - 1)));
    return dy * dy + dx * dx <= r2max && dy <= ry && dx <= rx;
  }
};

constexpr int PAIR_LIST_CAP = 4096;   // key blocks a super-tile may have to visit (host-checked)

constexpr int KEY_EMPTY = (int)0x80000000;

// One LDS-DMA wave instruction (64 lanes x 16 B -> 1 KiB at lds_dst), as inline assembly on purpose: for the builtin the
// compiler cannot tell the ring slot being filled from the slot being read and puts `s_waitcnt vmcnt(0)` -- the whole
// global-memory latency -- in front of every following ds_read.  The synchronisation (vmcnt(0) + barrier before the
// slot is read) is explicit in the kernel.
__device__ __forceinline__ void lds_dma_16(const void* src_lane, const void* lds_dst_uniform) {
  const uint32_t lds = (uint32_t)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds_dst_uniform;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src_lane), "s"(lds) : "memory");
}

}  // namespace fgvc
