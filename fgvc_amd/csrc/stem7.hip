// The ResNet stem on the bf16 matrix pipe: 7x7 / stride 2 / pad 3 convolution of the 3-channel frames, BatchNorm folded,
// + bias + ReLU (resnet.py:457-466), straight from the f32 NCHW frames to the two tensors layer 1 reads: dense NHWC f32
// (its first residual) and padded split NHWC.  Same arithmetic as conv_split.hip: activations and weights as (hi, lo) bf16
// pairs, hi*hi + lo*hi + hi*lo accumulated in f32.  Replaces MIOpen's f32 convolution, its bias kernel and the ReLU/split
// pass (0.9 ms of kernel time per 8-frame 480p clip); the kernel is bound by its own 420 MB of output.
//
// GEMM view: one K block per kernel ROW ky -- with the patch kept as NHWC4 bf16 rows (channel 3 = 0) the 7 taps x 4
// channels an output pixel needs from an input row are 28 CONSECUTIVE elements starting at column 2x - 3, i.e. at byte
// 16 (x - x0) of the staged row once the patch starts 3 columns left of the tile: every B operand is an aligned,
// conflict-free ds_read_b128, no im2col buffer.  K = 7 rows x 32 (28 used) = 14 MFMA steps instead of 147 / 16 = 9.2.
// The folded weights (7 x 2 steps x (hi, lo) operands of one 32-channel output tile = 112 VGPRs) stay in REGISTERS of
// persistent workgroups: wave (ct, rp) owns output channels 32 ct.. and output rows 2 rp, 2 rp + 1 of a 4 x 32 pixel tile.
#include "common.hpp"

namespace fgvc {

struct Stem7Params {
  const float* x;          // [N][3][H][W] f32
  const uint16_t* w;       // [7 ky][2 steps][2 cout tiles][hi | lo][64 lanes][8] bf16: MFMA-operand order (ops.prepare_stem7)
  const float* bias;       // [64]
  uint16_t* y_split;       // optional, padded split NHWC [N][Hop][Wop][2][64]
  float* y_f32;            // optional, dense NHWC f32 [N][Ho][Wo][64]
  int N, H, W, Ho, Wo, Hop, Wop, relu;
  int n_ty, n_tx, n_tiles;
  float out_scale;         // out_fmt 1: s_y of the split output
  int out_fmt;             // split output: 0 = (hi, lo) bf16, 1 = f16 + fp8 rows [h 64 B | l8 32 B | h8 32 B] (FGVC_ACT_F16F8)
  int* overflow;           // out_fmt 1: OR-ed with 1 when |s_y y| leaves the f16 range
};

constexpr int ST_TR = 4;                         // output rows per tile
constexpr int ST_ROWS = 2 * ST_TR + 5;           // input rows per tile
constexpr int ST_PW = 70;                        // staged pixels per row: 2 * 31 + 7 = 69, + 1 zero column (k = 28..31)
constexpr int ST_ROWB = ST_PW * 8;               // bytes per staged row and plane (4 channels x bf16)
constexpr int ST_PLANEB = ST_ROWS * ST_ROWB;
constexpr int ST_RS = 144;                       // epilogue tile row stride (bytes)
constexpr int ST_NPIX = ST_ROWS * ST_PW;         // staged pixel slots per tile
constexpr int ST_PPT = (ST_NPIX + 255) / 256;    // ... per thread

__global__ __launch_bounds__(256, 2) void stem7_kernel(Stem7Params p) {
  __shared__ __attribute__((aligned(16))) unsigned char patch[2 * ST_PLANEB];
  __shared__ __attribute__((aligned(16))) unsigned char tiles[4 * 32 * ST_RS];
  auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 1, rp = wave >> 1;
  const int n = lane & 31, h = lane >> 5;

  // weights of this wave's output-channel tile: registers for the whole kernel
  bf16x8 ah[7][2], al[7][2];
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint16_t* wp = p.w + ((((size_t)ky * 2 + s) * 2 + ct) * 2) * 512 + lane * 8;
      ah[ky][s] = *reinterpret_cast<const bf16x8*>(wp);
      al[ky][s] = *reinterpret_cast<const bf16x8*>(wp + 512);
    }
  f32x4 bv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(p.bias + ct * 32 + 8 * g + 4 * h);

  const size_t plane = (size_t)p.H * p.W;
  float pre[ST_PPT][3];
  auto fetch = [&](int tile) {                   // the tile's input pixels -> registers (zeros outside the image)
    const int nimg = tile / (p.n_ty * p.n_tx);
    const int rem = tile - nimg * p.n_ty * p.n_tx;
    const int ty = rem / p.n_tx, tx = rem - ty * p.n_tx;
    const int iy0 = 2 * ty * ST_TR - 3, ix0 = 2 * tx * 32 - 3;
    const float* img = p.x + (size_t)nimg * 3 * plane;
#pragma unroll
    for (int j = 0; j < ST_PPT; ++j) {
      const int slot = tid + 256 * j;
      const int row = slot / ST_PW, pc = slot - row * ST_PW;
      const int iy = iy0 + row, ix = ix0 + pc;
      const bool ok = slot < ST_NPIX && pc < ST_PW - 1 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const float* src = img + (size_t)(ok ? iy : 0) * p.W + (ok ? ix : 0);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = src[c * plane];
        pre[j][c] = ok ? v : 0.f;
      }
    }
  };
  auto commit = [&]() {                          // registers -> (hi | lo) NHWC4 rows in LDS
#pragma unroll
    for (int j = 0; j < ST_PPT; ++j) {
      const int slot = tid + 256 * j;
      if (slot < ST_NPIX) {
        ushort4 hv, lv;
        const f32x4 px = {pre[j][0], pre[j][1], pre[j][2], 0.f};
        split_bf16_4(px, hv, lv);
        *reinterpret_cast<ushort4*>(patch + slot * 8) = hv;
        *reinterpret_cast<ushort4*>(patch + ST_PLANEB + slot * 8) = lv;
      }
    }
  };

  int tile = blockIdx.x;
  if (tile < p.n_tiles) fetch(tile);
  for (; tile < p.n_tiles; tile += gridDim.x) {
    const int nimg = tile / (p.n_ty * p.n_tx);
    const int rem = tile - nimg * p.n_ty * p.n_tx;
    const int ty = rem / p.n_tx, tx = rem - ty * p.n_tx;
    const int y0 = ty * ST_TR, x0 = tx * 32;
    commit();
    lds_barrier();                                 // (not __syncthreads: that would also wait for the previous tile's stores)
    const int next = tile + gridDim.x;
    if (next < p.n_tiles) fetch(next);           // lands while this tile multiplies

    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int off = (2 * (2 * rp + b) + ky) * ST_ROWB + 16 * n + 32 * s + 16 * h;
          const bf16x8 xh = *reinterpret_cast<const bf16x8*>(patch + off);
          const bf16x8 xl = *reinterpret_cast<const bf16x8*>(patch + ST_PLANEB + off);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ky][s], xh, acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ky][s], xh, acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ky][s], xl, acc[b], 0, 0, 0);
        }
    lds_barrier();                                // patch free for the next tile's commit

    // ---- epilogue: bias + ReLU, transposed through a wave-private LDS tile: 128-byte rows per pixel
    unsigned char* tw = tiles + wave * (32 * ST_RS);
    const int mv_row = lane >> 3, mv_col = (lane & 7) * 16;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int y = y0 + 2 * rp + b;
      if (y >= p.Ho) continue;                    // wave-uniform
      f32x4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v[g] = {acc[b][4 * g + 0] + bv[g].x, acc[b][4 * g + 1] + bv[g].y, acc[b][4 * g + 2] + bv[g].z, acc[b][4 * g + 3] + bv[g].w};
        if (p.relu) {
          v[g].x = fmaxf(v[g].x, 0.f); v[g].y = fmaxf(v[g].y, 0.f); v[g].z = fmaxf(v[g].z, 0.f); v[g].w = fmaxf(v[g].w, 0.f);
        }
      }
      if (p.y_f32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(tw + n * ST_RS + (8 * g + 4 * h) * 4) = v[g];
        wave_sync();
        unsigned char* dst = reinterpret_cast<unsigned char*>(p.y_f32 + (((size_t)nimg * p.Ho + y) * p.Wo + x0) * 64 + ct * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i * 8 + mv_row;
          if (x0 + row < p.Wo)
            *reinterpret_cast<uint4*>(dst + (size_t)row * 256 + mv_col) = *reinterpret_cast<const uint4*>(tw + row * ST_RS + mv_col);
        }
        wave_sync();
      }
      if (p.y_split) {
        if (p.out_fmt == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 x = v[g];
            ushort4 hv, lv;
            split_bf16_4(x, hv, lv);
            unsigned char* o = tw + n * ST_RS + (8 * g + 4 * h) * 2;
            *reinterpret_cast<ushort4*>(o) = hv;
            *reinterpret_cast<ushort4*>(o + 64) = lv;
          }
        } else {
          bool ovf = false;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint2 hw, lw;
            uint32_t l8, h8;
            split_f16_4(v[g], p.out_scale, hw, l8, h8, lw, ovf);
            unsigned char* o = tw + n * ST_RS;
            *reinterpret_cast<uint2*>(o + (8 * g + 4 * h) * 2) = hw;
            *reinterpret_cast<uint32_t*>(o + 64 + 8 * g + 4 * h) = l8;
            *reinterpret_cast<uint32_t*>(o + 96 + 8 * g + 4 * h) = h8;
          }
          if (__builtin_amdgcn_ballot_w64(ovf && x0 + n < p.Wo) != 0ull && lane == 0) atomicOr(p.overflow, 1);
        }
        wave_sync();
        const size_t pix0 = ((size_t)nimg * p.Hop + (y + 1)) * p.Wop + (x0 + 1);
        unsigned char* dst = reinterpret_cast<unsigned char*>(p.y_split) + (pix0 * 2 + ct) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i * 8 + mv_row;
          if (x0 + row < p.Wo)
            *reinterpret_cast<uint4*>(dst + (size_t)row * 256 + mv_col) = *reinterpret_cast<const uint4*>(tw + row * ST_RS + mv_col);
        }
        wave_sync();
      }
    }
  }
}

int stem7_launch(const float* x, const uint16_t* w, const float* bias, uint16_t* y_split, float* y_f32, int N, int H, int W,
                 int Ho, int Wo, int Hop, int Wop, int relu, int out_fmt, int out_scale_log2, int* overflow, hipStream_t s) {
  Stem7Params p;
  p.out_fmt = out_fmt; p.out_scale = ldexpf(1.0f, out_scale_log2); p.overflow = overflow;
  p.x = x; p.w = w; p.bias = bias; p.y_split = y_split; p.y_f32 = y_f32;
  p.N = N; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.Hop = Hop; p.Wop = Wop; p.relu = relu;
  p.n_ty = cdiv(Ho, ST_TR); p.n_tx = cdiv(Wo, 32);
  const long long tiles = (long long)p.n_ty * p.n_tx * N;
  if (tiles >= (1ll << 31)) {
    set_error("fgvc_stem7_split_f32: too many tiles");
    return FGVC_ERR_UNSUPPORTED;
  }
  p.n_tiles = (int)tiles;
  const int grid = (int)(tiles < 512 ? tiles : 512);       // persistent: two workgroups per CU
  stem7_kernel<<<grid, 256, 0, s>>>(p);
  FGVC_CHECK_LAUNCH("fgvc_stem7_split_f32");
  return FGVC_OK;
}

}  // namespace fgvc
