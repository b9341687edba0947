// 64 -> 64 channel 3x3 stride-1 convolutions (ResNet layer 1: four per clip, on the largest activations of the trunk) with
// the folded weights RESIDENT IN REGISTERS.  Same arithmetic and tensors as conv_split.hip (split-bf16 operands, three
// partial products, f32 accumulate; padded split NHWC in; bias / residual / ReLU epilogue; dense NHWC f32 and / or padded
// split NHWC out).  Why a second kernel: in conv_split's form every one of the 6720 workgroups of such a layer streams the
// whole 147 KB weight tensor through its LDS ring -- 1 GB of L2 reads per launch, and 0.12 of the layer's 0.25-0.30 ms are
// that skeleton (conv_debug 6).  Here 256 persistent workgroups load their weights once: a wave's 32 output channels x
// 576 K x (hi, lo) are 72 MFMA operands = 288 VGPRs, so a wave owns a SIMD (4 waves per CU) and only pixels move:
// the (4+2) x 40 pixel patch of the NEXT tile is LDS-DMA'd into the second buffer while the current one multiplies,
// residuals are prefetched into registers, stores drain under the next tile.
// ARITH 1 (round 3): the f16 + fp8 arithmetic of conv_split.hip's wide layers (x -> h = f16(s x), e4m3 forms of h and of its
// residual; the main product on two K-16 f16 MFMAs, both cross sums in ONE K-64 fp8 MFMA with E8M0 block scales): 72 f16 + 36 fp8
// MFMAs per tile = 4608 pipe cycles instead of 216 bf16 MFMAs = 6912, the same 288 weight registers (144 of f16 fragments, 144 of
// fp8), the same 128-byte rows in the patch ([h 64 B | l8 32 B | h8 32 B] per pixel and 32-channel chunk).
#include "common.hpp"

namespace fgvc {

__device__ long long g_c64_probe[32];   // variant & 8: per wave of workgroup 77, cycles summed over its tiles: barrier, multiply loop, DMA wait, epilogue, tiles
static int g_conv64_variant = 0;
void set_conv64_variant(int v) { g_conv64_variant = v; }
int conv64_probe_read(long long* out32) {
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_c64_probe), 32 * sizeof(long long)) == hipSuccess ? 0 : -1;
}

struct Conv64Params {
  const uint16_t* x;       // padded split NHWC [N][Hp][Wp][2][64]
  const uint16_t* w;       // [2 cout tiles][9 taps][2 chunks] x 4 KiB in MFMA-operand order: ARITH 0 [2 k-steps][hi | lo][64 lanes][16 B]
                           // (ops.prepare_conv64), ARITH 1 [f16 k-step 0 | f16 k-step 1][64 lanes][16 B] then [64 lanes][h8 16 B | l8 16 B]
                           // (ops.prepare_conv64_f16)
  const float* bias;       // [64]
  const float* residual;   // optional, dense NHWC f32 [N][H][W][64]
  const uint16_t* res_split;   // optional (instead of `residual`): the residual as a padded split NHWC tensor of x's geometry; hi + lo
                               // is the value the next convolution sees anyway, and the producer need not write an f32 copy
  uint16_t* y_split;       // optional, padded split NHWC
  float* y_f32;            // optional, dense NHWC f32
  int N, H, W, Hp, Wp, relu;
  int n_ty, n_tx, n_tiles;
  float acc_scale;         // ARITH 1: 1 / (s_x s_w), the accumulator's scale (a power of two)
  float out_scale;         // out_fmt 1: s_y of the split output
  int out_fmt;             // split output: 0 = (hi, lo) bf16, 1 = f16 + fp8 (FGVC_ACT_F16F8)
  int* overflow;           // out_fmt 1: OR-ed with 1 when |s_y y| leaves the f16 range
  int variant;             // option "conv64_variant": 8 = s_memtime probe of workgroup 77 (fgvc_conv64_probe), 16 = the f16 + fp8 form on conv64_kernel<1>
};

__device__ __forceinline__ void c64_lds_dma_16(const void* src_lane, uint32_t lds_uniform) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src_lane), "s"(lds_uniform) : "memory");
}
__device__ __forceinline__ uint32_t c64_lds_addr(const void* p) {
  return (uint32_t)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
__device__ __forceinline__ int c64_swz(int row, int s) { return row * 128 + ((s ^ ((row >> 1) & 7)) << 4); }

constexpr int C64_TR = 4;                                  // output rows per tile
constexpr int C64_PW = 40;                                 // staged patch width (34 needed; DMA moves 8 pixels at a time)
constexpr int C64_CHUNKB = (C64_TR + 2) * C64_PW * 128;    // one 32-channel chunk of a patch
constexpr int C64_PATCHB = 2 * C64_CHUNKB;
constexpr int C64_RS = 144;                                // epilogue tile row stride (bytes)
constexpr int C64_PIECES = 2 * (C64_TR + 2) * 5;           // 1-KiB DMA pieces per patch

#define C64_READ(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
// address = scalar row base + lane part, as an opaque instruction: left visible, the compiler computes each of a tile's 48 distinct
// sums once and keeps them alive from their first to their last use (spills in the f16 + fp8 form)
#define C64_ADDR(DST, SBASE, LANE) asm volatile("v_add_u32 %0, %1, %2" : "=v"(DST) : "s"(SBASE), "v"(LANE))
#define C64_MFMA_A(ACC, W, X) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "a"(W), "v"(X))
#define C64_MFMA_V(ACC, W, X) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(W), "v"(X))
// first product of a tile: SrcC is the inline constant 0, the accumulator an early-clobber OUTPUT.  Zeroing it with vector moves
// put a VALU write right in front of a matrix instruction that reads it as SrcC -- a hazard the compiler covers for its own MFMAs
// and cannot see in an assembly statement (the first build of the f16 + fp8 form read stale operand bits that way).
#define C64_MFMA_A0(ACC, W, X) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(ACC) : "a"(W), "v"(X))
#define C64_MFMA_F0(ACC, W, X) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(ACC) : "a"(W), "v"(X))
#define C64_MFMA_F(ACC, W, X) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "a"(W), "v"(X))
#define C64_MFMA_XA(ACC, W, X, SA, SB) \
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(ACC) : "a"(W), "v"(X), "v"(SA), "v"(SB))
#define C64_MFMA_XV(ACC, W, X, SA, SB) \
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(ACC) : "v"(W), "v"(X), "v"(SA), "v"(SB))

template <int ARITH>
__global__ __launch_bounds__(256, 1) void conv64_kernel(Conv64Params p) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) unsigned char patches[2 * C64_PATCHB];
  __shared__ __attribute__((aligned(16))) unsigned char tiles[4 * 32 * C64_RS];
  __shared__ __attribute__((aligned(16))) float bias_s[64];
  auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 1, rg = wave >> 1;
  const int n = lane & 31, h = lane >> 5;
  const int d_row = lane >> 3, d_slot = lane & 7;

  // this wave's weights, registers for the whole kernel.  ARITH 0: 9 taps x 2 chunks x 2 k-steps x (hi, lo) bf16 fragments;
  // ARITH 1: 9 x 2 x (two f16 fragments + one 8-register fp8 operand: h8 of the weights in the first scale block, l8 in the second)
  bf16x8 wh[9][2][2], wl[9][2][2];
  f16x8 wf[9][2][2];
  i32x8 wx[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint16_t* wp = p.w + ((((size_t)ct * 9 + t) * 2 + c) * 4) * 512;
      if constexpr (ARITH == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          wh[t][c][s] = *reinterpret_cast<const bf16x8*>(wp + (s * 2) * 512 + lane * 8);
          wl[t][c][s] = *reinterpret_cast<const bf16x8*>(wp + (s * 2 + 1) * 512 + lane * 8);
        }
      } else {
        wf[t][c][0] = *reinterpret_cast<const f16x8*>(wp + lane * 8);
        wf[t][c][1] = *reinterpret_cast<const f16x8*>(wp + 512 + lane * 8);
        wx[t][c] = *reinterpret_cast<const i32x8*>(wp + 1024 + lane * 16);
      }
    }
  // E8M0 scales of this lane half's block of the fp8 operands (127 + log2, every byte): lanes 0-31 carry the first 32 K elements
  // (weights h8 = h / 2^AW against activations l8 = l 2^BX), lanes 32-63 the second (l8 = l 2^BW against h8 = h / 2^AX)
  const uint32_t sa_ = h ? (uint32_t)(127 - F8_BW) : (uint32_t)(127 + F8_AW), sb_ = h ? (uint32_t)(127 + F8_AX) : (uint32_t)(127 - F8_BX);
  const int scale_a = (int)(sa_ * 0x01010101u), scale_b = (int)(sb_ * 0x01010101u);
  if (tid < 64) bias_s[tid] = p.bias[tid];                 // read back per tile (registers are for the weights)

  // pieces i, i + 4, ... of a patch belong to this wave; piece -> (chunk, patch row, 8-pixel column group).  Scalar base +
  // 32-bit lane offset form of the DMA: the lane part is (row in the piece) * 256 + swizzled slot * 16, and since the first
  // patch pixel of a piece is a multiple of 8 the swizzle key (P >> 1) & 7 is (d_row >> 1) + {0, 4} = (d_row >> 1) ^ {0, 4}:
  // one lane constant, XORed with 64 for every other piece.
  const uint32_t lane_off0 = (uint32_t)(d_row * 256 + ((d_slot ^ (d_row >> 1)) << 4));
  auto stage_piece = [&](int nimg, int y0, int x0, int buf, int i) {
    const int chunk = i / ((C64_TR + 2) * 5), r2 = i - chunk * ((C64_TR + 2) * 5);
    const int prow = r2 / 5, pc0 = (r2 - prow * 5) * 8;
    const int key4 = ((prow * C64_PW + pc0) >> 1) & 4;       // wave-uniform
    const size_t gpix = ((size_t)nimg * p.Hp + (y0 + prow)) * p.Wp + (x0 + pc0);
    const unsigned char* base = reinterpret_cast<const unsigned char*>(p.x) + gpix * 256 + chunk * 128;
    const uint32_t off = lane_off0 ^ (uint32_t)(key4 << 4);
    const uint32_t dst = c64_lds_addr(patches + buf * C64_PATCHB + chunk * C64_CHUNKB + (prow * C64_PW + pc0) * 128);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory");
  };
  // Tile order: workgroup b takes tiles b, b + G, ... in raster order.  (An XCD-aware order -- the 32 workgroups of an XCD on 32
  // consecutive tiles of one image column, so that the patch rows vertical neighbours share meet in one L2 -- measured 0.7 % SLOWER
  // end to end on the same box, profiles/r03_bv_xcd.log.)
  const int G = gridDim.x;
  auto tile_at = [&](int k) { return k * G + (int)blockIdx.x; };
  auto tile_origin = [&](int tile, int& nimg, int& y0, int& x0) {
    nimg = tile / (p.n_ty * p.n_tx);
    const int rem = tile - nimg * p.n_ty * p.n_tx;
    const int ty = rem / p.n_tx;
    y0 = ty * C64_TR;
    x0 = (rem - ty * p.n_tx) * 32;
  };

  // lane parts of the B-operand addresses: pixel P = r * 40 + n + dx of the patch lives at P * 128 + ((slot ^ key(P)) << 4) with
  // key(P) = (P >> 1) & 7 = (((n + dx) >> 1) & 7) ^ (4 (r & 1))  (40 r / 2 = 20 r = 4 r mod 8); slot = 2 s + h for the hi part,
  // 4 + 2 s + h for the lo part (= the hi address with bit 6 flipped)
  uint32_t lane_b[3][2][2][2];                              // [dx][k-step][row parity][hi | lo]
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const uint32_t key = (uint32_t)((((n + dx) >> 1) & 7) ^ (4 * par));
        const uint32_t a0 = (uint32_t)((n + dx) * 128) + ((((uint32_t)(2 * s2 + h)) ^ key) << 4);
        lane_b[dx][s2][par][0] = a0;
        lane_b[dx][s2][par][1] = a0 ^ 64u;
      }
  const bool probe = (p.variant & 8) && blockIdx.x == 77;
  long long pb = 0, pm = 0, pw = 0, pe = 0, pn = 0;
  int it = 0, tile = tile_at(0), buf = 0;
  if (tile < p.n_tiles) {
    int ni, ya, xa;
    tile_origin(tile, ni, ya, xa);
    for (int i = wave; i < C64_PIECES; i += 4) stage_piece(ni, ya, xa, 0, i);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (; tile < p.n_tiles; tile = tile_at(++it), buf ^= 1) {       // tile_at(k) grows with k: the first tile beyond the end is the last
    int nimg, y0, x0;
    tile_origin(tile, nimg, y0, x0);
    const long long t0 = probe ? __builtin_amdgcn_s_memtime() : 0;
    lds_barrier();                                          // patch `buf` complete (every wave waited for its DMAs), buffer buf^1 free
    // the next tile's patch goes into the other buffer while this one multiplies: its 15 DMA pieces per wave are issued
    // BETWEEN the MFMA groups (a wave owns its SIMD: whatever it issues outside the MFMA stream is exposed)
    const long long t1 = probe ? __builtin_amdgcn_s_memtime() : 0;
    const int next = tile_at(it + 1);
    const bool has_next = next < p.n_tiles;
    int nimg_n = 0, y0_n = 0, x0_n = 0;
    if (has_next) tile_origin(next, nimg_n, y0_n, x0_n);

    f32x4 res[2][4];
    fgvc_i32x4 resw[2][4];                                  // ... or its raw (hi, lo) bf16 words, converted in the epilogue

    f32x16 acc[2];                                          // (written first by the zero-SrcC products of group 0)
    // B operands of group g = (tap, chunk, k-step): [row][hi | lo], read one group ahead of the multiplies (left to itself
    // hipcc reads each operand right before its first use: 144 exposed LDS round trips per tile, 3x the MFMA time).  The reads are
    // assembly with explicit addresses: address = (row base: scalar) + (lane part: one of 24 registers set up once) + (chunk:
    // immediate).  As C++ loads the compiler kept the lane part of all 144 addresses of a tile in registers across the tile loop,
    // and spilled them.  The LDS returns a wave's reads in order and nothing else of this wave uses the LDS inside the loop: one
    // lgkmcnt(0) at the top of a group covers the operands read during the group before.
    const uint32_t pbase = c64_lds_addr(patches) + (uint32_t)(buf * C64_PATCHB);
    auto side_work = [&](int k, int k_res) {    // what rides between the MFMA groups: DMA piece k - 1 of the next patch (k = 1..15), the residual loads
      if (k >= 1 && k <= 15 && has_next) stage_piece(nimg_n, y0_n, x0_n, buf ^ 1, wave + 4 * (k - 1));
      if (k == k_res && p.residual) {   // residual rows of this wave in accumulator layout (pixel on the lane, 4 consecutive channels per register group)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int y = imin(y0 + 2 * rg + b, p.H - 1), x = imin(x0 + n, p.W - 1);
          const float* rp = p.residual + (((size_t)nimg * p.H + y) * p.W + x) * 64 + ct * 32 + 4 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q) res[b][q] = *reinterpret_cast<const f32x4*>(rp + 8 * q);
        }
      }
      if (ARITH == 0 && k == k_res && p.res_split) {  // the same from the bf16 split form: channel c of a pixel's 32-channel chunk is hi at byte 2 c, lo at 64 + 2 c
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int y = imin(y0 + 2 * rg + b, p.H - 1), x = imin(x0 + n, p.W - 1);
          const unsigned char* rp = reinterpret_cast<const unsigned char*>(p.res_split) +
                                    ((((size_t)nimg * p.Hp + (y + 1)) * p.Wp + (x + 1)) * 2 + ct) * 128 + 8 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // the RAW words ride in the registers until the epilogue (round 4): converted here, the first shift waited for the loads --
            // vmcnt(0) in the middle of the multiplies of the SIMD's only wave, once per tile (round 3: 0.07 ms per launch slower)
            const uint2 hv = *reinterpret_cast<const uint2*>(rp + 16 * q);
            const uint2 lv = *reinterpret_cast<const uint2*>(rp + 64 + 16 * q);
            resw[b][q] = fgvc_i32x4{(int)hv.x, (int)hv.y, (int)lv.x, (int)lv.y};
          }
        }
      }
    };
    // The MFMAs are inline assembly so that the weight operands can be PINNED: 256 of the 288 weight registers in the accumulation
    // registers (used by the matrix instruction directly), 32 in vector registers.  Left to the compiler the weights were spread
    // over both files with the accumulation half as SPILL space: 300 v_accvgpr_read copies per tile in the only wave of the SIMD,
    // and the multiply loop ran at 65 cycles per MFMA (s_memtime probe, profiles/r03_probe_conv64_before.log, _after.log).
    if constexpr (ARITH == 0) {
      bf16x8 bc[4], bn[4];
      auto load_b = [&](bf16x8* d, int g) {
        const int t = g >> 2, c = (g >> 1) & 1, s = g & 1;
        const int dy = t / 3, dx = t % 3;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int r = 2 * rg + b + dy;                                  // patch row (wave-uniform)
          const uint32_t rowbase = pbase + (uint32_t)(r * C64_PW * 128);
          uint32_t ah, al;
          C64_ADDR(ah, rowbase, lane_b[dx][s][r & 1][0]);
          C64_ADDR(al, rowbase, lane_b[dx][s][r & 1][1]);
          if (c == 0) {
            C64_READ(d[b * 2 + 0], ah, 0);
            C64_READ(d[b * 2 + 1], al, 0);
          } else {
            C64_READ(d[b * 2 + 0], ah, C64_CHUNKB);
            C64_READ(d[b * 2 + 1], al, C64_CHUNKB);
          }
        }
      };
      load_b(bc, 0);
#pragma unroll
      for (int g = 0; g < 36; ++g) {
        const int t = g >> 2, c = (g >> 1) & 1, s = g & 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // this group's operands (read during the group before)
        if (g + 1 < 36) load_b(bn, g + 1);
        side_work((g & 1) == 0 ? g / 2 : -1, 12);                         // pieces in groups 2, 4, .., 30; residual in group 24
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1" ::: "memory");
        {   // same order per accumulator as ever (hi hi, lo hi, hi lo), the two pixel rows interleaved; lo parts of taps 7-8 from vector registers
          const bf16x8 xh0 = bc[0], xl0 = bc[1], xh1 = bc[2], xl1 = bc[3];
          if (g == 0) {
            C64_MFMA_A0(acc[0], wh[t][c][s], xh0);
            C64_MFMA_A0(acc[1], wh[t][c][s], xh1);
          } else {
            C64_MFMA_A(acc[0], wh[t][c][s], xh0);
            C64_MFMA_A(acc[1], wh[t][c][s], xh1);
          }
          if (t < 7) {
            C64_MFMA_A(acc[0], wl[t][c][s], xh0);
            C64_MFMA_A(acc[1], wl[t][c][s], xh1);
          } else {
            C64_MFMA_V(acc[0], wl[t][c][s], xh0);
            C64_MFMA_V(acc[1], wl[t][c][s], xh1);
          }
          C64_MFMA_A(acc[0], wh[t][c][s], xl0);
          C64_MFMA_A(acc[1], wh[t][c][s], xl1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) bc[q] = bn[q];
      }
    } else {
      // group g = (tap, chunk): per pixel row the two f16 fragments (slots h, 2 + h of the row) and the fp8 operand (slot 4 + h: l8 of
      // the activations, first scale block; slot 6 + h: h8, second block).  fp8 operands of taps 7-8 from vector registers.
      f16x8 fc[2][2], fn[2][2];
      i32x4 xc[2][2], xn[2][2];
      auto load_b = [&](f16x8 (*df)[2], i32x4 (*dxp)[2], int g) {
        const int t = g >> 1, c = g & 1;
        const int dy = t / 3, dx = t % 3;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int r = 2 * rg + b + dy;
          const uint32_t rowbase = pbase + (uint32_t)(r * C64_PW * 128);
          uint32_t a0, a1, a2, a3;
          C64_ADDR(a0, rowbase, lane_b[dx][0][r & 1][0]);
          C64_ADDR(a1, rowbase, lane_b[dx][1][r & 1][0]);
          C64_ADDR(a2, rowbase, lane_b[dx][0][r & 1][1]);
          C64_ADDR(a3, rowbase, lane_b[dx][1][r & 1][1]);
          if (c == 0) {
            C64_READ(df[b][0], a0, 0); C64_READ(df[b][1], a1, 0); C64_READ(dxp[b][0], a2, 0); C64_READ(dxp[b][1], a3, 0);
          } else {
            C64_READ(df[b][0], a0, C64_CHUNKB); C64_READ(df[b][1], a1, C64_CHUNKB);
            C64_READ(dxp[b][0], a2, C64_CHUNKB); C64_READ(dxp[b][1], a3, C64_CHUNKB);
          }
        }
      };
      load_b(fc, xc, 0);
#pragma unroll
      for (int g = 0; g < 18; ++g) {
        const int t = g >> 1, c = g & 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (g + 1 < 18) load_b(fn, xn, g + 1);
        side_work(g, 12);                                                 // pieces in groups 1..15, residual in group 12
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1" ::: "memory");
        {
          const i32x8 x0v = __builtin_shufflevector(xc[0][0], xc[0][1], 0, 1, 2, 3, 4, 5, 6, 7);
          const i32x8 x1v = __builtin_shufflevector(xc[1][0], xc[1][1], 0, 1, 2, 3, 4, 5, 6, 7);
          if (g == 0) {
            C64_MFMA_F0(acc[0], wf[t][c][0], fc[0][0]);
            C64_MFMA_F0(acc[1], wf[t][c][0], fc[1][0]);
          } else {
            C64_MFMA_F(acc[0], wf[t][c][0], fc[0][0]);
            C64_MFMA_F(acc[1], wf[t][c][0], fc[1][0]);
          }
          C64_MFMA_F(acc[0], wf[t][c][1], fc[0][1]);
          C64_MFMA_F(acc[1], wf[t][c][1], fc[1][1]);
          if (t < 7) {
            C64_MFMA_XA(acc[0], wx[t][c], x0v, scale_a, scale_b);
            C64_MFMA_XA(acc[1], wx[t][c], x1v, scale_a, scale_b);
          } else {
            C64_MFMA_XV(acc[0], wx[t][c], x0v, scale_a, scale_b);
            C64_MFMA_XV(acc[1], wx[t][c], x1v, scale_a, scale_b);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int q = 0; q < 2; ++q) { fc[b][q] = fn[b][q]; xc[b][q] = xn[b][q]; }
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMAs' results before vector instructions read them (the compiler
                                                             // does not see matrix instructions in the assembly statements)
    // the next patch's DMAs (and the residual loads) are older than anything the epilogue issues: one wait covers them and
    // leaves this tile's stores in flight across the barrier
    const long long t2 = probe ? __builtin_amdgcn_s_memtime() : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t3 = probe ? __builtin_amdgcn_s_memtime() : 0;

    // ---- epilogue: bias (+ residual) (+ ReLU), transposed through a wave-private LDS tile: 128-byte rows per pixel
    unsigned char* tw = tiles + wave * (32 * C64_RS);
    const int mv_row = lane >> 3, mv_col = (lane & 7) * 16;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int y = y0 + 2 * rg + b;
      if (y >= p.H) continue;                               // wave-uniform
      f32x4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + ct * 32 + 8 * g + 4 * h);
        if constexpr (ARITH == 0)
          v[g] = {acc[b][4 * g + 0] + bv.x, acc[b][4 * g + 1] + bv.y, acc[b][4 * g + 2] + bv.z, acc[b][4 * g + 3] + bv.w};
        else          // the accumulator holds s_x s_w times the convolution (powers of two: the product below is exact)
          v[g] = {fmaf(acc[b][4 * g + 0], p.acc_scale, bv.x), fmaf(acc[b][4 * g + 1], p.acc_scale, bv.y),
                  fmaf(acc[b][4 * g + 2], p.acc_scale, bv.z), fmaf(acc[b][4 * g + 3], p.acc_scale, bv.w)};
        if (p.residual) v[g] += res[b][g];
        if (ARITH == 0 && p.res_split) {                    // (hi, lo) bf16 pairs -> f32: hi + lo (bf16 tensors only: capi)
          const uint32_t h0 = (uint32_t)resw[b][g][0], h1 = (uint32_t)resw[b][g][1], l0 = (uint32_t)resw[b][g][2], l1 = (uint32_t)resw[b][g][3];
          v[g] += f32x4{__builtin_bit_cast(float, h0 << 16) + __builtin_bit_cast(float, l0 << 16),
                        __builtin_bit_cast(float, h0 & 0xffff0000u) + __builtin_bit_cast(float, l0 & 0xffff0000u),
                        __builtin_bit_cast(float, h1 << 16) + __builtin_bit_cast(float, l1 << 16),
                        __builtin_bit_cast(float, h1 & 0xffff0000u) + __builtin_bit_cast(float, l1 & 0xffff0000u)};
        }
        if (p.relu) {
          v[g].x = fmaxf(v[g].x, 0.f); v[g].y = fmaxf(v[g].y, 0.f); v[g].z = fmaxf(v[g].z, 0.f); v[g].w = fmaxf(v[g].w, 0.f);
        }
      }
      if (p.y_f32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(tw + n * C64_RS + (8 * g + 4 * h) * 4) = v[g];
        wave_sync();
        unsigned char* dst = reinterpret_cast<unsigned char*>(p.y_f32 + (((size_t)nimg * p.H + y) * p.W + x0) * 64 + ct * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i * 8 + mv_row;
          if (x0 + row < p.W)
            *reinterpret_cast<uint4*>(dst + (size_t)row * 256 + mv_col) = *reinterpret_cast<const uint4*>(tw + row * C64_RS + mv_col);
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
            unsigned char* o = tw + n * C64_RS + (8 * g + 4 * h) * 2;
            *reinterpret_cast<ushort4*>(o) = hv;
            *reinterpret_cast<ushort4*>(o + 64) = lv;
          }
        } else {                                    // the f16 + fp8 form the next layer reads: [h 64 B | l8 32 B | h8 32 B]
          bool ovf = false;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint2 hw, lw;
            uint32_t l8, h8;
            split_f16_4(v[g], p.out_scale, hw, l8, h8, lw, ovf);
            unsigned char* o = tw + n * C64_RS;
            *reinterpret_cast<uint2*>(o + (8 * g + 4 * h) * 2) = hw;
            *reinterpret_cast<uint32_t*>(o + 64 + 8 * g + 4 * h) = l8;
            *reinterpret_cast<uint32_t*>(o + 96 + 8 * g + 4 * h) = h8;
          }
          if (__builtin_amdgcn_ballot_w64(ovf && x0 + n < p.W) != 0ull && lane == 0) atomicOr(p.overflow, 1);
        }
        wave_sync();
        const size_t pix0 = ((size_t)nimg * p.Hp + (y + 1)) * p.Wp + (x0 + 1);
        unsigned char* dst = reinterpret_cast<unsigned char*>(p.y_split) + (pix0 * 2 + ct) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i * 8 + mv_row;
          if (x0 + row < p.W)
            *reinterpret_cast<uint4*>(dst + (size_t)row * 256 + mv_col) = *reinterpret_cast<const uint4*>(tw + row * C64_RS + mv_col);
        }
        wave_sync();
      }
    }
    if (probe) {
      const long long t4 = __builtin_amdgcn_s_memtime();
      pb += t1 - t0; pm += t2 - t1; pw += t3 - t2; pe += t4 - t3; pn += 1;
    }
  }
  if (probe && lane == 0) {
    g_c64_probe[wave * 8 + 0] = pb; g_c64_probe[wave * 8 + 1] = pm; g_c64_probe[wave * 8 + 2] = pw; g_c64_probe[wave * 8 + 3] = pe;
    g_c64_probe[wave * 8 + 4] = pn;
  }
}

// ---- round 5: the f16 + fp8 form with the K dimension split across the waves ------------------------------------------------------
// conv64_kernel's wave (ct, rg) multiplies 32 output channels x 2 pixel rows over the whole K = 576: every activation operand it reads
// from the LDS feeds ONE matrix instruction, the two waves of a row pair read the same bytes, and a patch row is read again for every
// kernel row it serves: 576 KiB of LDS reads per tile = 4608 cycles at 128 B/clock -- exactly the tile's 4608 MFMA cycles, on top of
// the patch DMA (60 KiB) and the epilogue's transposition (128 KiB).  The LDS, not the matrix pipe, paced the multiply loop
// (11 000 of its 14 500 cycles per tile).  Here wave (kh, rg) owns ALL 64 output channels x 2 pixel rows over ONE 32-channel chunk of the input
// (the same 288 weight registers), and walks the 4 patch rows x 3 column shifts of its row pair once: an operand (4 reads) feeds up
// to 12 matrix instructions -- both channel tiles, and both pixel rows it is a kernel-row neighbour of.  48 KiB of reads per wave and
// tile instead of 144.  The two chunk halves of an accumulator meet through the LDS: each wave hands the channel tile it does not
// finish to its partner (8 KiB), adds what it receives (f32 addition commutes: the sum does not depend on who adds) and runs the old
// epilogue on channel tile kh.  Accumulation order: per chunk as before (taps ascending; f16 k-steps, then the fp8 cross terms), the two
// chunk sums added last (conv64_kernel interleaves the chunks tap by tap: last-bit differences, both within the test bounds).
constexpr int C64_XCH = 8192;                              // per wave: 2 pixel rows x 16 accumulator registers x 64 lanes x 4 B

__global__ __launch_bounds__(256, 1) void conv64k_kernel(Conv64Params p) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) unsigned char patches[2 * C64_PATCHB];
  __shared__ __attribute__((aligned(16))) unsigned char xch[4 * C64_XCH];     // exchange of partial sums, then the epilogue's tiles
  __shared__ __attribute__((aligned(16))) float bias_s[64];
  auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wave & 1, rg = wave >> 1;                  // input chunk of the multiplies = output channel tile of the epilogue; row pair
  const int n = lane & 31, h = lane >> 5;
  const int d_row = lane >> 3, d_slot = lane & 7;

  // weights: index 0 of the channel-tile dimension = the tile this wave FINISHES (ct = kh), index 1 = the one it hands to its partner
  f16x8 wf[9][2][2];                                        // [tap][own | other][k-step]
  i32x8 wx[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint16_t* wp = p.w + ((((size_t)(j ^ kh) * 9 + t) * 2 + kh) * 4) * 512;
      wf[t][j][0] = *reinterpret_cast<const f16x8*>(wp + lane * 8);
      wf[t][j][1] = *reinterpret_cast<const f16x8*>(wp + 512 + lane * 8);
      wx[t][j] = *reinterpret_cast<const i32x8*>(wp + 1024 + lane * 16);
    }
  const uint32_t sa_ = h ? (uint32_t)(127 - F8_BW) : (uint32_t)(127 + F8_AW), sb_ = h ? (uint32_t)(127 + F8_AX) : (uint32_t)(127 - F8_BX);
  const int scale_a = (int)(sa_ * 0x01010101u), scale_b = (int)(sb_ * 0x01010101u);
  if (tid < 64) bias_s[tid] = p.bias[tid];

  // The patch DMA.  A wave issues ONE instruction per four cycles (the sequencer visits a SIMD every fourth cycle and this wave is the
  // SIMD's only one), so every scalar instruction of the address arithmetic costs what a vector instruction costs: conv64_kernel's
  // per-piece origin arithmetic (divisions by constants, 64-bit multiplies: ~27 instructions x 15 pieces) was 1 600 of a tile's 14 500
  // cycles.  Here lane k of one register holds what piece wave + 4 k needs -- lanes 0-14 the byte offset of its 8 pixels from the
  // tile's first patch pixel, lanes 16-30 its LDS offset in a patch buffer, lanes 32-46 its swizzle key (0 or 64: the XOR of the lanes'
  // source offsets) --, three v_readlane per piece.
  const uint32_t lane_off0 = (uint32_t)(d_row * 256 + ((d_slot ^ (d_row >> 1)) << 4));
  int piece_tab;
  {
    const int k = lane & 15, i = wave + 4 * k;              // (k = 15: unused)
    const int chunk = i / ((C64_TR + 2) * 5), r2 = i - chunk * ((C64_TR + 2) * 5);
    const int prow = r2 / 5, pc0 = (r2 - prow * 5) * 8;
    const int goff = (prow * p.Wp + pc0) * 256 + chunk * 128;
    const int ldst = chunk * C64_CHUNKB + (prow * C64_PW + pc0) * 128;
    const int key = (((prow * C64_PW + pc0) >> 1) & 4) << 4;
    piece_tab = lane < 16 ? goff : lane < 32 ? ldst : key;
  }
  const uint32_t patches_lds = c64_lds_addr(patches);
  auto stage_piece = [&](const unsigned char* tile_base, int buf, int k) {     // k: compile-time
    const uint32_t goff = (uint32_t)__builtin_amdgcn_readlane(piece_tab, k);
    const uint32_t dst = patches_lds + (uint32_t)(buf * C64_PATCHB) + (uint32_t)__builtin_amdgcn_readlane(piece_tab, 16 + k);
    const uint32_t off = lane_off0 ^ (uint32_t)__builtin_amdgcn_readlane(piece_tab, 32 + k);
    const unsigned char* base = tile_base + goff;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory");
  };
  const int G = gridDim.x;
  auto tile_at = [&](int k) { return k * G + (int)blockIdx.x; };
  auto tile_origin = [&](int tile, int& nimg, int& y0, int& x0) {
    nimg = tile / (p.n_ty * p.n_tx);
    const int rem = tile - nimg * p.n_ty * p.n_tx;
    const int ty = rem / p.n_tx;
    y0 = ty * C64_TR;
    x0 = (rem - ty * p.n_tx) * 32;
  };
  auto patch_base = [&](int nimg, int y0, int x0) {
    return reinterpret_cast<const unsigned char*>(p.x) + (((size_t)nimg * p.Hp + y0) * p.Wp + x0) * 256;
  };
  // lane part of the operand addresses: pixel n + dx of a patch row, slot h of its 128-byte row in this wave's chunk; [dx][patch row
  // parity].  Slots 2 + h (f16 k-step 1), 4 + h (l8), 6 + h (h8) are this address with bits 5 / 6 / 5-6 flipped (everything else in it is
  // a multiple of 128).
  uint32_t lane_b[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const uint32_t key = (uint32_t)((((n + dx) >> 1) & 7) ^ (4 * par));
      lane_b[dx][par] = (uint32_t)((n + dx) * 128 + kh * C64_CHUNKB) + ((((uint32_t)h) ^ key) << 4);
    }
  // the f32 residual as a raw buffer (range-checked: conv64_launch keeps this kernel to tensors below 4 GiB)
  const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.residual), 0, p.residual ? (int)((unsigned)p.N * (unsigned)p.H * (unsigned)p.W * 256u) : 0, 0x00020000);
  const int res_voff = (lane >> 3) * 256 + (lane & 7) * 16;
  const bool probe = (p.variant & 8) && blockIdx.x == 77;
  long long pb = 0, pm = 0, pw = 0, pe = 0, pn = 0;
  int it = 0, tile = tile_at(0), buf = 0;
  if (tile < p.n_tiles) {
    int ni, ya, xa;
    tile_origin(tile, ni, ya, xa);
    const unsigned char* tb = patch_base(ni, ya, xa);
#pragma unroll
    for (int k = 0; k < 15; ++k) stage_piece(tb, 0, k);
  }
  // (the BUILTIN, not an assembly statement: the compiler's own wait-count pass must see that the weight loads above are complete.  It
  // did not see conv64_kernel's assembly wait, guarded the first use of every weight register INSIDE the tile loop with a counted
  // vmcnt(N) of its own -- needed in the first tile only, executed in all -- and, blind to the assembly DMAs in flight, each of those
  // waited for recent DMA pieces to land: most of what the residual's loads seemed to cost, 2 800-7 000 cycles per tile)
  __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
  for (; tile < p.n_tiles; tile = tile_at(++it), buf ^= 1) {
    int nimg, y0, x0;
    tile_origin(tile, nimg, y0, x0);
    const long long t0 = probe ? __builtin_amdgcn_s_memtime() : 0;
    lds_barrier();                                          // patch `buf` complete, buffer buf ^ 1 and the exchange region free
    const long long t1 = probe ? __builtin_amdgcn_s_memtime() : 0;
    const int next = tile_at(it + 1);
    const bool has_next = next < p.n_tiles;
    int nimg_n = 0, y0_n = 0, x0_n = 0;
    if (has_next) tile_origin(next, nimg_n, y0_n, x0_n);
    const unsigned char* next_base = patch_base(nimg_n, y0_n, x0_n);

    i32x4 res[2][4];                                       // residual rows in the STORE layout: [pixel row b][pixels 8 i + lane / 8], 16 B at (lane & 7) * 16
    f32x16 acc[2][2];                                       // [own | other channel tile][pixel row of the pair]
    const uint32_t pbase = patches_lds + (uint32_t)(buf * C64_PATCHB) + (uint32_t)(2 * rg * C64_PW * 128);
    f16x8 fc[2], fn[2];
    i32x4 xc[2], xn[2];
    auto load_b = [&](f16x8* df, i32x4* dxp, int g) {       // group g = (patch row r of the wave's four, column shift dx)
      const int r = g / 3, dx = g % 3;
      const uint32_t rowbase = pbase + (uint32_t)(r * C64_PW * 128);
      uint32_t a0;
      C64_ADDR(a0, rowbase, lane_b[dx][r & 1]);
      const uint32_t a1 = a0 ^ 32u, a2 = a0 ^ 64u, a3 = a0 ^ 96u;
      C64_READ(df[0], a0, 0); C64_READ(df[1], a1, 0); C64_READ(dxp[0], a2, 0); C64_READ(dxp[1], a3, 0);
    };
    // What rides in the shadow of matrix instruction j of group g (a 32x32x16 f16 product covers 7 more issue slots, the fp8 one 15):
    // the next group's operand reads behind the first, the next patch's DMA pieces (15 per wave) behind the first two fp8 products,
    // the residual rows' loads late in the tile.
    auto side = [&](int g, int j, int M) {
      if (j == 0 && g + 1 < 12) load_b(fn, xn, g + 1);
      if (has_next) {
        if (j == M - 2 && 2 * g < 15) stage_piece(next_base, buf ^ 1, 2 * g);
        if (j == M - 1 && 2 * g + 1 < 15) stage_piece(next_base, buf ^ 1, 2 * g + 1);
      }
      if (p.residual && (g == 7 || g == 8) && j >= 2 && j < 6) {
        // eight loads of 8 pixels x 128 B: scalar row origin + the lane's (pixel, 16-byte column) + an immediate; pixels beyond the row's end
        // read what follows in the tensor (never stored), beyond the tensor's end the buffer's range check returns zeros
        const int b = g - 7, i = j - 2;
        const int y = imin(y0 + 2 * rg + b, p.H - 1);
        const int soff = ((nimg * p.H + y) * p.W + x0) * 256 + kh * 128;
        res[b][i] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, res_voff + i * 2048, soff, 0);
      }
    };
    load_b(fc, xc, 0);
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      const int r = g / 3, dx = g % 3;
      const int M = (r == 0 || r == 3) ? 6 : 12;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this group's operands (read during the group before)
      const i32x8 xv = __builtin_shufflevector(xc[0], xc[1], 0, 1, 2, 3, 4, 5, 6, 7);
      int j = 0;
      // pixel row b of the pair sees patch row r as kernel row dy = r - b
#pragma unroll
      for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int dy = r - b;
          if (dy < 0 || dy > 2) continue;
          const int t = dy * 3 + dx;
          const bool first = (dy == 0 && dx == 0 && part == 0);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            if (part < 2) {
              if (first) C64_MFMA_F0(acc[ct][b], wf[t][ct][part], fc[part]);
              else C64_MFMA_F(acc[ct][b], wf[t][ct][part], fc[part]);
            } else if (t < 7) {
              C64_MFMA_XA(acc[ct][b], wx[t][ct], xv, scale_a, scale_b);
            } else {
              C64_MFMA_XV(acc[ct][b], wx[t][ct], xv, scale_a, scale_b);
            }
            side(g, j, M);
            ++j;
          }
        }
      fc[0] = fn[0]; fc[1] = fn[1]; xc[0] = xn[0]; xc[1] = xn[1];
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMAs' results before vector instructions read them
    const long long t2 = probe ? __builtin_amdgcn_s_memtime() : 0;
    // ---- the other channel tile's partial sums to the partner wave (same row pair, other chunk)
    {
      unsigned char* xw = xch + wave * C64_XCH + lane * 16;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(xw + (b * 4 + q) * 1024) = f32x4{acc[1][b][4 * q + 0], acc[1][b][4 * q + 1], acc[1][b][4 * q + 2], acc[1][b][4 * q + 3]};
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): the next patch's DMAs and the residual loads (older than anything the epilogue issues)
    const long long t3 = probe ? __builtin_amdgcn_s_memtime() : 0;
    lds_barrier();
    unsigned char* tw = xch + (wave ^ 1) * C64_XCH;         // what the partner left; afterwards this wave's transposition tile (only this wave
                                                            // reads the region, and the partner writes it again behind the next tile's barrier)
    f32x4 fin[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(tw + lane * 16 + (b * 4 + q) * 1024);
        fin[b][q] = f32x4{acc[0][b][4 * q + 0], acc[0][b][4 * q + 1], acc[0][b][4 * q + 2], acc[0][b][4 * q + 3]} + o;
      }
    wave_sync();

    // ---- epilogue on channel tile kh: scale + bias (+ residual) (+ ReLU), the output formats, 128-byte rows per pixel through the wave's
    // LDS tile.  Without a residual (a block's first convolution) conv64_kernel's order: everything in the accumulator layout (pixel on the
    // lane), the finished rows transposed.  With one: the sums are transposed FIRST and the rest happens in the store layout (8 lanes per
    // pixel, 4 channels each) -- there the residual's loads are the coalesced ones prefetched above.  In the accumulator layout they were
    // 64 lanes x 16 B from 32 cache lines per instruction, eight per tile: 2 800 cycles of a tile's 17 000 (the wave waits at the ISSUE of
    // such a load, and nothing else runs on its SIMD).
    const int mv_row = lane >> 3, mv_col = (lane & 7) * 16;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int y = y0 + 2 * rg + b;
      if (y >= p.H) continue;                               // wave-uniform
      f32x4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + kh * 32 + 8 * g + 4 * h);
        v[g] = {fmaf(fin[b][g].x, p.acc_scale, bv.x), fmaf(fin[b][g].y, p.acc_scale, bv.y), fmaf(fin[b][g].z, p.acc_scale, bv.z),
                fmaf(fin[b][g].w, p.acc_scale, bv.w)};
      }
      const size_t pix0 = ((size_t)nimg * p.Hp + (y + 1)) * p.Wp + (x0 + 1);
      unsigned char* dst_s = reinterpret_cast<unsigned char*>(p.y_split) + (pix0 * 2 + kh) * 128;
      unsigned char* dst_f = reinterpret_cast<unsigned char*>(p.y_f32 + (((size_t)nimg * p.H + y) * p.W + x0) * 64 + kh * 32);
      if (p.residual) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(tw + n * C64_RS + (8 * g + 4 * h) * 4) = v[g];
        wave_sync();
        f32x4 u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = *reinterpret_cast<const f32x4*>(tw + (i * 8 + mv_row) * C64_RS + mv_col);
        wave_sync();
        bool ovf = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i * 8 + mv_row;
          u[i] += __builtin_bit_cast(f32x4, res[b][i]);
          if (p.relu) {
            u[i].x = fmaxf(u[i].x, 0.f); u[i].y = fmaxf(u[i].y, 0.f); u[i].z = fmaxf(u[i].z, 0.f); u[i].w = fmaxf(u[i].w, 0.f);
          }
          if (p.y_f32 && x0 + row < p.W) *reinterpret_cast<f32x4*>(dst_f + (size_t)row * 256 + mv_col) = u[i];
          if (p.y_split) {                                  // this lane's 4 channels of the pixel's split row: [h 64 B | l8 32 B | h8 32 B] or (hi, lo) bf16
            unsigned char* o = tw + row * C64_RS;
            if (p.out_fmt == 0) {
              ushort4 hv, lv;
              split_bf16_4(u[i], hv, lv);
              *reinterpret_cast<ushort4*>(o + (mv_col >> 1)) = hv;
              *reinterpret_cast<ushort4*>(o + 64 + (mv_col >> 1)) = lv;
            } else {
              uint2 hw, lw;
              uint32_t l8, h8;
              bool of1 = false;
              split_f16_4(u[i], p.out_scale, hw, l8, h8, lw, of1);
              ovf |= of1 && x0 + row < p.W;
              *reinterpret_cast<uint2*>(o + (mv_col >> 1)) = hw;
              *reinterpret_cast<uint32_t*>(o + 64 + (mv_col >> 2)) = l8;
              *reinterpret_cast<uint32_t*>(o + 96 + (mv_col >> 2)) = h8;
            }
          }
        }
        if (p.y_split) {
          if (p.out_fmt != 0 && __builtin_amdgcn_ballot_w64(ovf) != 0ull && lane == 0) atomicOr(p.overflow, 1);
          wave_sync();
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + mv_row;
            if (x0 + row < p.W)
              *reinterpret_cast<uint4*>(dst_s + (size_t)row * 256 + mv_col) = *reinterpret_cast<const uint4*>(tw + row * C64_RS + mv_col);
          }
          wave_sync();
        }
        continue;
      }
      if (p.relu) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          v[g].x = fmaxf(v[g].x, 0.f); v[g].y = fmaxf(v[g].y, 0.f); v[g].z = fmaxf(v[g].z, 0.f); v[g].w = fmaxf(v[g].w, 0.f);
        }
      }
      if (p.y_f32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(tw + n * C64_RS + (8 * g + 4 * h) * 4) = v[g];
        wave_sync();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i * 8 + mv_row;
          if (x0 + row < p.W)
            *reinterpret_cast<uint4*>(dst_f + (size_t)row * 256 + mv_col) = *reinterpret_cast<const uint4*>(tw + row * C64_RS + mv_col);
        }
        wave_sync();
      }
      if (p.y_split) {
        if (p.out_fmt == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            ushort4 hv, lv;
            split_bf16_4(v[g], hv, lv);
            unsigned char* o = tw + n * C64_RS + (8 * g + 4 * h) * 2;
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
            unsigned char* o = tw + n * C64_RS;
            *reinterpret_cast<uint2*>(o + (8 * g + 4 * h) * 2) = hw;
            *reinterpret_cast<uint32_t*>(o + 64 + 8 * g + 4 * h) = l8;
            *reinterpret_cast<uint32_t*>(o + 96 + 8 * g + 4 * h) = h8;
          }
          if (__builtin_amdgcn_ballot_w64(ovf && x0 + n < p.W) != 0ull && lane == 0) atomicOr(p.overflow, 1);
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i * 8 + mv_row;
          if (x0 + row < p.W)
            *reinterpret_cast<uint4*>(dst_s + (size_t)row * 256 + mv_col) = *reinterpret_cast<const uint4*>(tw + row * C64_RS + mv_col);
        }
        wave_sync();
      }
    }
    if (probe) {
      const long long t4 = __builtin_amdgcn_s_memtime();
      pb += t1 - t0; pm += t2 - t1; pw += t3 - t2; pe += t4 - t3; pn += 1;
    }
  }
  if (probe && lane == 0) {
    g_c64_probe[wave * 8 + 0] = pb; g_c64_probe[wave * 8 + 1] = pm; g_c64_probe[wave * 8 + 2] = pw; g_c64_probe[wave * 8 + 3] = pe;
    g_c64_probe[wave * 8 + 4] = pn;
  }
}

// ---- round 5: the f16 + fp8 form as ONE instruction stream per tile ---------------------------------------------------------------
// What the s_memtime probes of conv64_kernel<1> and of the K-split experiment above showed: the LDS was never the limit.  A wave that
// owns its SIMD issues one instruction per ~4 cycles, in order; a matrix instruction occupies the pipe for 32 (f16 32x32x16) or 64
// cycles (fp8 32x32x64) and only what is issued RIGHT BEHIND it -- ~5 resp. ~13 instructions -- is free.  conv64_kernel put the
// operand reads, the DMA's address arithmetic (27 scalar instructions per piece) and the residual's loads BETWEEN its groups of six
// matrix instructions and the whole epilogue behind the last: ~2 400 of a tile's ~3 600 instructions ran with the pipe idle, 14 500
// cycles per tile against 4 608 of matrix work.  The compiler's own s_waitcnt vmcnt(N) -- counted without the assembly DMAs, guarding
// the first use of each weight register INSIDE the loop because it had not seen the assembly wait in front of it -- stalled on recent
// DMA pieces on top of that.
// Here: conv64_kernel's decomposition (wave = 32 output channels x 2 pixel rows, all of K; the same accumulation order, bit for bit),
// but (1) an operand read from the LDS once per patch row serves both pixel rows it is a kernel-row neighbour of (96 reads per tile
// instead of 144); (2) the DMA reads its per-piece constants from a lane table (10 instructions per piece); (3) the residual rows are
// loaded in the STORE layout (8 lanes x 16 B per pixel: coalesced) and every global access of the loop is a raw-buffer instruction whose
// descriptor ends where the tile's row ends -- no predicates, no branches, and every one is issued every tile, so (4) each s_waitcnt
// vmcnt(N) is exact; (5) the epilogue of tile i - 1 runs in the gaps of tile i's matrix instructions, step by step, in the order
// tools/gen_conv64p_sched.py deals out (conv64p_sched_*.inc).
#define C64P_SB __builtin_amdgcn_sched_barrier(0);
#define MARK(S) __builtin_amdgcn_sched_barrier(0); asm volatile("; @@" S); __builtin_amdgcn_sched_barrier(0);      // (a comment in the ISA: tools/conv64p_costs.py counts a step's instructions between two of them)
#define C64P_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

template <bool RES, bool F32OUT, int OUT_FMT>
__global__ __launch_bounds__(256, 1) void conv64p_kernel(Conv64Params p) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) unsigned char patches[2 * C64_PATCHB];
  __shared__ __attribute__((aligned(16))) unsigned char tiles[4 * 32 * C64_RS];
  __shared__ __attribute__((aligned(16))) float bias_s[64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 1, rg = wave >> 1;
  const int n = lane & 31, h = lane >> 5;
  const int d_row = lane >> 3, d_slot = lane & 7;

  f16x8 wf[9][2][2];                                        // [tap][chunk][k-step]
  i32x8 wx[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint16_t* wp = p.w + ((((size_t)ct * 9 + t) * 2 + c) * 4) * 512;
      wf[t][c][0] = *reinterpret_cast<const f16x8*>(wp + lane * 8);
      wf[t][c][1] = *reinterpret_cast<const f16x8*>(wp + 512 + lane * 8);
      wx[t][c] = *reinterpret_cast<const i32x8*>(wp + 1024 + lane * 16);
    }
  const uint32_t sa_ = h ? (uint32_t)(127 - F8_BW) : (uint32_t)(127 + F8_AW), sb_ = h ? (uint32_t)(127 + F8_AX) : (uint32_t)(127 - F8_BX);
  const int scale_a = (int)(sa_ * 0x01010101u), scale_b = (int)(sb_ * 0x01010101u);
  if (tid < 64) bias_s[tid] = p.bias[tid];

  // DMA: lane k of piece_tab holds what piece wave + 4 k needs (conv64k_kernel)
  const uint32_t lane_off0 = (uint32_t)(d_row * 256 + ((d_slot ^ (d_row >> 1)) << 4));
  int piece_tab;
  {
    const int k = lane & 15, i = wave + 4 * k;
    const int chunk = i / ((C64_TR + 2) * 5), r2 = i - chunk * ((C64_TR + 2) * 5);
    const int prow = r2 / 5, pc0 = (r2 - prow * 5) * 8;
    const int goff = (prow * p.Wp + pc0) * 256 + chunk * 128;
    const int ldst = chunk * C64_CHUNKB + (prow * C64_PW + pc0) * 128;
    const int key = (((prow * C64_PW + pc0) >> 1) & 4) << 4;
    piece_tab = lane < 16 ? goff : lane < 32 ? ldst : key;
  }
  const uint32_t patches_lds = c64_lds_addr(patches);
  auto stage_piece = [&](const unsigned char* tile_base, uint32_t buf_lds, int k) {
    const uint32_t goff = (uint32_t)__builtin_amdgcn_readlane(piece_tab, k);
    const uint32_t dst = buf_lds + (uint32_t)__builtin_amdgcn_readlane(piece_tab, 16 + k);
    const uint32_t off = lane_off0 ^ (uint32_t)__builtin_amdgcn_readlane(piece_tab, 32 + k);
    const unsigned char* base = tile_base + goff;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory");
  };
  const int G = gridDim.x;
  auto tile_origin = [&](int tile, int& nimg, int& y0, int& x0) {
    nimg = tile / (p.n_ty * p.n_tx);
    const int rem = tile - nimg * p.n_ty * p.n_tx;
    const int ty = rem / p.n_tx;
    y0 = ty * C64_TR;
    x0 = (rem - ty * p.n_tx) * 32;
  };
  auto patch_base = [&](int nimg, int y0, int x0) {
    return reinterpret_cast<const unsigned char*>(p.x) + (uint32_t)(((nimg * p.Hp + y0) * p.Wp + x0) * 256);
  };
  // operand addresses: [dx][patch row parity][f16 k-step 0 | k-step 1 | l8 | h8] in the CURRENT patch buffer, rows of this wave's pair;
  // patch row r and chunk c are immediates of the reads, the buffer is flipped (flip()) behind an address's last read of a tile
  uint32_t lane_a[3][2][4];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const uint32_t key = (uint32_t)((((n + dx) >> 1) & 7) ^ (4 * par));
#pragma unroll
      for (int k = 0; k < 4; ++k)
        lane_a[dx][par][k] = patches_lds + (uint32_t)(2 * rg * C64_PW * 128 + (n + dx) * 128) + ((((uint32_t)(2 * k + h)) ^ key) << 4);
    }
  int flip_delta = C64_PATCHB;                              // added to an address when the tile behind this one reads the other buffer
  // epilogue constants
  unsigned char* tw = tiles + wave * (32 * C64_RS);
  const int mv_row = lane >> 3, mv_col = (lane & 7) * 16;
  const int st_goff = mv_row * 256 + mv_col, st_goff2 = st_goff + 4096;
  const float relu_lo = p.relu ? 0.f : -INFINITY;
  const float acc_scale = p.acc_scale, out_scale = p.out_scale;
  constexpr int RSRC3 = 0x00020000;

  // ---- state carried from tile to tile
  f32x4 fin[2][4];                                          // the previous tile's sums, accumulator layout
  i32x4 res[2][4];                                          // its residual rows, store layout
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) { fin[b][g] = f32x4{0.f, 0.f, 0.f, 0.f}; res[b][g] = i32x4{0, 0, 0, 0}; }
  int nimg_p = 0, y0_p = 0, x0_p = p.W;                     // ... and where it lies (x0 = W: nothing of it is stored -- the first tile has no predecessor)
  i32x4 d_f32[2], d_spl[2], d_res[2];                       // raw-buffer descriptors of the rows: previous tile's outputs, this tile's residual
  f32x4 bv[4], u[4];
  i32x4 rw[4];
  float amax[4];
  f32x4 sp_c, sp_hf, sp_lf, sp_t;
  uint2 sp_hw, sp_lw;
  uint32_t sp_h8, sp_l8;

  const bool probe = (p.variant & 8) && blockIdx.x == 77;
  long long pb = 0, pm = 0, pw = 0, pe = 0, pn = 0;
  // Tile order.  The layer is bound by HBM now (624 MB per 8-frame launch at 5 TB/s), a third of it patch rows that vertical neighbours
  // share -- fetched twice when the neighbours run on different XCDs (raster order: tiles t and t + n_tx on workgroups 14 apart, L2 hit
  // rate 0.30).  With the full grid of 256 persistent workgroups (dispatched round-robin over the 8 XCDs) the tiles are walked COLUMN-major
  // (y fastest) and the 32 workgroups of an XCD take 32 consecutive ones of a round: one image column's strip in one L2.
  // (option conv64_variant & 128: raster order.  conv64_kernel tried this order in round 3, issue-bound then: 0.7 % slower.)
  const bool colmajor = G == 256 && !(p.variant & 128);
  int tile = colmajor ? ((int)blockIdx.x & 7) * 32 + ((int)blockIdx.x >> 3) : (int)blockIdx.x, buf = 0;
  const int n_fast = colmajor ? p.n_ty : p.n_tx, n_slow = colmajor ? p.n_tx : p.n_ty;
  const int g_img = G / (p.n_ty * p.n_tx), g_rem = G - g_img * (p.n_ty * p.n_tx), g_slow = g_rem / n_fast, g_fast = g_rem - g_slow * n_fast;
  int nimg, y0, x0;
  uint32_t f_c, s_c;                                        // the tile's coordinates along the fast and the slow axis of the walk
  {
    nimg = tile / (p.n_ty * p.n_tx);
    const int rem = tile - nimg * p.n_ty * p.n_tx;
    s_c = (uint32_t)(rem / n_fast);
    f_c = (uint32_t)rem - s_c * (uint32_t)n_fast;
    y0 = (int)(colmajor ? f_c : s_c) * C64_TR;
    x0 = (int)(colmajor ? s_c : f_c) * 32;
  }
  {
    const unsigned char* tb = patch_base(nimg, y0, x0);
#pragma unroll
    for (int k = 0; k < 15; ++k) stage_piece(tb, patches_lds, k);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0), as the BUILTIN: the compiler's wait-count pass must know the weights have landed
  lds_barrier();

  // (byte offsets within a tensor fit 32 bits: conv64_launch keeps larger tensors on conv64_kernel -- scalar arithmetic throughout; a
  // 64-bit product would run on the vector unit)
  auto row_desc = [&](const void* base, uint32_t byte_off, int nrec) {
    const size_t a = (size_t)base + byte_off;
    return i32x4{__builtin_amdgcn_readfirstlane((int)(uint32_t)a), __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)),
                 __builtin_amdgcn_readfirstlane(nrec), RSRC3};
  };
  auto buf_store = [&](const i32x4& d, int voff, const i32x4& v, int imm) {   // imm: 0 or 2048
    if (imm == 0) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(d) : "memory");
    else asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen offset:2048\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(d) : "memory");
  };

  // ---- the steps of the previous tile's epilogue (called from the generated schedule with literal arguments)
  // raw-buffer descriptors of the rows [pixel x0 .. min(x0 + 32, W)) of image row y: the previous tile's outputs, this tile's residual
  auto row_records = [&](int x0_, int y) {
    const int nv = imin(imax(p.W - x0_, 0), 32);
    return (y < p.H && nv > 0) ? nv * 256 - 128 : 0;
  };
  auto e_desc_s = [&](int b) {
    const int y = y0_p + 2 * rg + b;
    d_spl[b] = row_desc(p.y_split, (uint32_t)((((nimg_p * p.Hp + (y + 1)) * p.Wp + (x0_p + 1)) * 2 + ct) * 128), row_records(x0_p, y));
  };
  auto e_desc_f = [&](int b) {
    const int y = y0_p + 2 * rg + b;
    if (F32OUT) d_f32[b] = row_desc(p.y_f32, (uint32_t)((((nimg_p * p.H + y) * p.W + x0_p) * 2 + ct) * 128), row_records(x0_p, y));
  };
  auto e_desc_r = [&](int b) {                              // (loaded late in this tile, used in the next)
    const int y = y0 + 2 * rg + b;
    if (RES) d_res[b] = row_desc(p.residual, (uint32_t)((((nimg * p.H + y) * p.W + x0) * 2 + ct) * 128), row_records(x0, y));
  };
  auto resload = [&](int b, int i) {
    if (!RES) return;
#define C64P_RL(B, I, REG, VOFF, IMM) \
  if (b == B && i == I) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" IMM : "=" REG(res[B][I]) : "v"(VOFF), "s"(d_res[B]) : "memory");
    C64P_RL(0, 0, "{v[64:67]}", st_goff, "") C64P_RL(0, 1, "{v[68:71]}", st_goff, " offset:2048")
    C64P_RL(0, 2, "{v[72:75]}", st_goff2, "") C64P_RL(0, 3, "{v[76:79]}", st_goff2, " offset:2048")
    C64P_RL(1, 0, "{v[80:83]}", st_goff, "") C64P_RL(1, 1, "{v[84:87]}", st_goff, " offset:2048")
    C64P_RL(1, 2, "{v[88:91]}", st_goff2, "") C64P_RL(1, 3, "{v[92:95]}", st_goff2, " offset:2048")
#undef C64P_RL
  };
  auto e_bias = [&]() {
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(bias_s + ct * 32 + 8 * g + 4 * h);
  };
  auto e_fma = [&](int b, int g) {
    const f32x4 f = fin[b][g];
    f32x4 v = {fmaf(f.x, acc_scale, bv[g].x), fmaf(f.y, acc_scale, bv[g].y), fmaf(f.z, acc_scale, bv[g].z), fmaf(f.w, acc_scale, bv[g].w)};
    if (RES) *reinterpret_cast<f32x4*>(tw + n * C64_RS + (8 * g + 4 * h) * 4) = v;
    else u[g] = v;
  };
  auto e_uread = [&](int b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = *reinterpret_cast<const f32x4*>(tw + (i * 8 + mv_row) * C64_RS + mv_col);
  };
  auto e_res = [&](int b, int i) { u[i] += __builtin_bit_cast(f32x4, res[b][i]); };
  auto e_relu = [&](int i) {
    u[i] = f32x4{fmaxf(u[i].x, relu_lo), fmaxf(u[i].y, relu_lo), fmaxf(u[i].z, relu_lo), fmaxf(u[i].w, relu_lo)};
  };
  auto e_stf = [&](int b, int i) {
    if (F32OUT) buf_store(d_f32[b], i < 2 ? st_goff : st_goff2, __builtin_bit_cast(i32x4, u[i]), (i & 1) * 2048);
  };
  // split_f16_4 / split_bf16_4 (common.hpp) on u[j], the same operations in the same order, in pieces of 4-6 instructions (what fits
  // behind an f16 matrix instruction)
  auto e_sp1 = [&](int j) {
    if (OUT_FMT == 0) {
      const uint32_t h0 = f2bf_pk(u[j].x, u[j].y), h1 = f2bf_pk(u[j].z, u[j].w);
      sp_hw = uint2{h0, h1};
    } else {
      sp_c = u[j] * out_scale;
    }
  };
  auto e_sp2 = [&](int j) {
    if (OUT_FMT == 0) {
      sp_hf = f32x4{__builtin_bit_cast(float, sp_hw.x << 16), __builtin_bit_cast(float, sp_hw.x & 0xFFFF0000u), __builtin_bit_cast(float, sp_hw.y << 16),
                    __builtin_bit_cast(float, sp_hw.y & 0xFFFF0000u)};
    } else {
      amax[j] = fmaxf(fmaxf(fabsf(sp_c.x), fabsf(sp_c.y)), fmaxf(fabsf(sp_c.z), fabsf(sp_c.w)));
    }
  };
  auto e_sp3 = [&](int j) {
    if (OUT_FMT == 0) {
      sp_lf = u[j] - sp_hf;
    } else {
      sp_c = f32x4{__builtin_amdgcn_fmed3f(sp_c.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(sp_c.y, -65504.f, 65504.f),
                   __builtin_amdgcn_fmed3f(sp_c.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(sp_c.w, -65504.f, 65504.f)};
    }
  };
  auto e_sp4 = [&](int j) {
    if (OUT_FMT == 0) {
      sp_lw = uint2{f2bf_pk(sp_lf.x, sp_lf.y), f2bf_pk(sp_lf.z, sp_lf.w)};
    } else {
      const f16x4 hh = {(_Float16)sp_c.x, (_Float16)sp_c.y, (_Float16)sp_c.z, (_Float16)sp_c.w};
      sp_hw = __builtin_bit_cast(uint2, hh);
    }
  };
  auto e_sp5 = [&](int j) {
    const f16x4 hh = __builtin_bit_cast(f16x4, sp_hw);
    sp_hf = f32x4{(float)hh.x, (float)hh.y, (float)hh.z, (float)hh.w};
  };
  auto e_sp6 = [&](int j) { sp_lf = sp_c - sp_hf; };
  auto e_sp7 = [&](int j) {
    constexpr float SA = 1.0f / (float)(1 << F8_AX);
    sp_t = sp_hf * SA;
  };
  auto e_sp8 = [&](int j) {
    int a = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(sp_t.x, -448.f, 448.f), __builtin_amdgcn_fmed3f(sp_t.y, -448.f, 448.f), 0, false);
    a = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(sp_t.z, -448.f, 448.f), __builtin_amdgcn_fmed3f(sp_t.w, -448.f, 448.f), a, true);
    sp_h8 = (uint32_t)a;
  };
  auto e_sp9 = [&](int j) {
    constexpr float SBX = (float)(1 << F8_BX);
    const f32x4 t = sp_lf * SBX;
    int bq = __builtin_amdgcn_cvt_pk_fp8_f32(t.x, t.y, 0, false);
    bq = __builtin_amdgcn_cvt_pk_fp8_f32(t.z, t.w, bq, true);
    sp_l8 = (uint32_t)bq;
  };
  auto e_spw = [&](int j) {                               // RES: j = pixel group i of the store layout; else: j = channel group g of the accumulator layout
    unsigned char* o;
    int c16;                                              // first of the lane's 4 channels within the 32-channel tile
    if (RES) { o = tw + (j * 8 + mv_row) * C64_RS; c16 = mv_col >> 2; }
    else { o = tw + n * C64_RS; c16 = 8 * j + 4 * h; }
    if (OUT_FMT == 0) {
      *reinterpret_cast<uint2*>(o + c16 * 2) = sp_hw;
      *reinterpret_cast<uint2*>(o + 64 + c16 * 2) = sp_lw;
    } else {
      *reinterpret_cast<uint2*>(o + c16 * 2) = sp_hw;
      *reinterpret_cast<uint32_t*>(o + 64 + c16) = sp_l8;
      *reinterpret_cast<uint32_t*>(o + 96 + c16) = sp_h8;
    }
  };
  auto e_ovf = [&](int b) {
    if (OUT_FMT == 0) return;
    const int nv = p.W - x0_p;
    const float m = fmaxf(fmaxf(amax[0], amax[1]), fmaxf(amax[2], amax[3]));
    bool of = m > 57344.f;
    if (nv < 32) {                                          // (the image's last tile column: pixels beyond the row's end do not count)
      if (RES) of = (amax[0] > 57344.f && mv_row < nv) || (amax[1] > 57344.f && 8 + mv_row < nv) || (amax[2] > 57344.f && 16 + mv_row < nv) ||
                    (amax[3] > 57344.f && 24 + mv_row < nv);
      else of = of && n < nv;
    }
    if (y0_p + 2 * rg + b >= p.H) of = false;
    if (__builtin_amdgcn_ballot_w64(of) != 0ull && lane == 0) atomicOr(p.overflow, 1);
  };
  auto e_rread = [&](int b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rw[i] = *reinterpret_cast<const i32x4*>(tw + (i * 8 + mv_row) * C64_RS + mv_col);
  };
  auto e_sts = [&](int b, int i) { buf_store(d_spl[b], i < 2 ? st_goff : st_goff2, rw[i], (i & 1) * 2048); };


  for (;;) {
    const long long t1 = probe ? __builtin_amdgcn_s_memtime() : 0;
    int nimg_n, y0_n, x0_n;
    uint32_t f_n, s_n;
    const unsigned char* next_base;
    const uint32_t pbuf = (uint32_t)(buf * C64_PATCHB);     // (only the DMA target needs it: the operand addresses carry the buffer)
    const uint32_t next_lds = patches_lds + (uint32_t)((buf ^ 1) * C64_PATCHB);
    (void)pbuf;
    f32x16 acc[2];
    f16x8 fa0, fa1, fb0, fb1;
    i32x4 xa0, xa1, xb0, xb1;

    // ---- the steps (called from the generated schedule with literal arguments)
    // Everything an assembly statement writes ASYNCHRONOUSLY -- the accumulators, the operand buffers, the residual rows -- sits in NAMED
    // registers: to the compiler an assembly output is complete when the statement ends, and a copy it makes afterwards (it moved one
    // accumulator to other registers between two matrix instructions of the first build, under register pressure) reads what has not
    // landed yet.  A value that every statement wants in the same physical registers is never moved.
    //   acc[0] v[0:15]  acc[1] v[16:31]   buffer a: f16 k-steps v[32:35] v[36:39], fp8 operand v[40:47]   buffer b: v[48:51] v[52:55] v[56:63]
    //   residual rows v[64:79] (b = 0), v[80:95] (b = 1)
    auto opread = [&](int q) {
      const int r = q / 6, dx = (q % 6) / 2, c = q % 2;
      const uint32_t* a = lane_a[dx][r & 1];
      // (offsets: patch row r, chunk c -- at most 3 * 5120 + 30720 < 65536)
      if ((q & 1) == 0) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "={v[32:35]}"(fa0) : "v"(a[0]), "i"(r * C64_PW * 128 + c * C64_CHUNKB) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "={v[36:39]}"(fa1) : "v"(a[1]), "i"(r * C64_PW * 128 + c * C64_CHUNKB) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "={v[40:43]}"(xa0) : "v"(a[2]), "i"(r * C64_PW * 128 + c * C64_CHUNKB) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "={v[44:47]}"(xa1) : "v"(a[3]), "i"(r * C64_PW * 128 + c * C64_CHUNKB) : "memory");
      } else {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "={v[48:51]}"(fb0) : "v"(a[0]), "i"(r * C64_PW * 128 + c * C64_CHUNKB) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "={v[52:55]}"(fb1) : "v"(a[1]), "i"(r * C64_PW * 128 + c * C64_CHUNKB) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "={v[56:59]}"(xb0) : "v"(a[2]), "i"(r * C64_PW * 128 + c * C64_CHUNKB) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "={v[60:63]}"(xb1) : "v"(a[3]), "i"(r * C64_PW * 128 + c * C64_CHUNKB) : "memory");
      }
    };
    auto tilenext_a = [&]() {                               // the tile behind this one -- or this one again (its patch is loaded a second
      // time, into the free buffer, and never read: every tile issues the same instructions).  G tiles further = (g_img, g_ty, g_tx)
      // further with carries: no division in the loop (tile_origin's two cost 45 scalar instructions)
      uint32_t fn = f_c + (uint32_t)g_fast, sn = s_c + (uint32_t)g_slow, nin = (uint32_t)nimg + (uint32_t)g_img;
      const uint32_t c1 = fn >= (uint32_t)n_fast ? 1u : 0u;
      fn -= c1 ? (uint32_t)n_fast : 0u;
      sn += c1;
      const uint32_t c2 = sn >= (uint32_t)n_slow ? 1u : 0u;
      sn -= c2 ? (uint32_t)n_slow : 0u;
      nin += c2;
      const bool has = tile + G < p.n_tiles;
      f_n = has ? fn : f_c; s_n = has ? sn : s_c; nimg_n = has ? (int)nin : nimg;
      y0_n = (int)((colmajor ? f_n : s_n) * C64_TR); x0_n = (int)((colmajor ? s_n : f_n) * 32);
    };
    auto tilenext_b = [&]() { next_base = patch_base(nimg_n, y0_n, x0_n); };
    const unsigned char* dma_base;
    uint32_t dma_dst, dma_off;
    auto dma_a = [&](int k) {
      dma_base = next_base + (uint32_t)__builtin_amdgcn_readlane(piece_tab, k);
      dma_dst = next_lds + (uint32_t)__builtin_amdgcn_readlane(piece_tab, 16 + k);
      dma_off = lane_off0 ^ (uint32_t)__builtin_amdgcn_readlane(piece_tab, 32 + k);
    };
    auto dma_b = [&](int k) {
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(dma_off), "s"(dma_base), "s"(dma_dst) : "memory");
    };
    auto flip = [&](int dx, int par, int k) { lane_a[dx][par][k] += (uint32_t)flip_delta; };
#define GB(Q) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define C64P_RACC_0 "{v[0:15]}"
#define C64P_RACC_1 "{v[16:31]}"
#define C64P_RF_a0 "{v[32:35]}"
#define C64P_RF_a1 "{v[36:39]}"
#define C64P_RF_b0 "{v[48:51]}"
#define C64P_RF_b1 "{v[52:55]}"
#define C64P_RX_a "{v[40:47]}"
#define C64P_RX_b "{v[56:63]}"
#define MF_F0(B, T, C, P, BUF) \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&" C64P_RACC_##B(acc[B]) : "a"(wf[T][C][P]), C64P_RF_##BUF##P(f##BUF##P));
#define MF_F(B, T, C, P, BUF) \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+" C64P_RACC_##B(acc[B]) : "a"(wf[T][C][P]), C64P_RF_##BUF##P(f##BUF##P));
#define MF_XA(B, T, C, P, BUF)                                                                                                    \
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"                                          \
               : "+" C64P_RACC_##B(acc[B])                                                                                          \
               : "a"(wx[T][C]), C64P_RX_##BUF(__builtin_shufflevector(x##BUF##0, x##BUF##1, 0, 1, 2, 3, 4, 5, 6, 7)), "v"(scale_a), "v"(scale_b));
#define MF_XV(B, T, C, P, BUF)                                                                                                    \
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"                                          \
               : "+" C64P_RACC_##B(acc[B])                                                                                          \
               : "v"(wx[T][C]), C64P_RX_##BUF(__builtin_shufflevector(x##BUF##0, x##BUF##1, 0, 1, 2, 3, 4, 5, 6, 7)), "v"(scale_a), "v"(scale_b));
// the last matrix instructions' results before vector instructions read them.  The accumulators are OPERANDS of the pause: the copy into
// `fin` below cannot be scheduled in front of it (hoisted to the accumulator's last matrix instruction it read sums still in the pipe)
#define C64P_ACC_DONE asm volatile("s_nop 15\n\ts_nop 15" : "+{v[0:15]}"(acc[0]), "+{v[16:31]}"(acc[1])::"memory");
#define SB C64P_SB
#define LB asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define C64P_N(A, F) ((A) + (F32OUT ? (F) : 0))
// (the residual registers are operands of the wait: what reads them cannot be scheduled in front of it)
#define WAITRES(B)                                                                                                                  \
  if ((B) == 0) asm volatile("s_waitcnt vmcnt(%4)" : "+{v[64:67]}"(res[0][0]), "+{v[68:71]}"(res[0][1]), "+{v[72:75]}"(res[0][2]), "+{v[76:79]}"(res[0][3]) : "n"(C64P_VM_RES0) : "memory"); \
  else asm volatile("s_waitcnt vmcnt(%4)" : "+{v[80:83]}"(res[1][0]), "+{v[84:87]}"(res[1][1]), "+{v[88:91]}"(res[1][2]), "+{v[92:95]}"(res[1][3]) : "n"(C64P_VM_RES1) : "memory");
    opread(0);
    if constexpr (!RES) {
#define C64P_SECTION 0
#include "conv64p_sched_plain.inc"
#undef C64P_SECTION
#define C64P_SECTION 1
#include "conv64p_sched_plain.inc"
#undef C64P_SECTION
      C64P_ACC_DONE
      C64P_WAIT_VM(C64P_VM_END);                            // the next patch has landed (this wave's pieces)
#undef C64P_VM_END
    } else if constexpr (OUT_FMT == 1) {
#define C64P_SECTION 0
#include "conv64p_sched_res.inc"
#undef C64P_SECTION
#define C64P_SECTION 1
#include "conv64p_sched_res.inc"
#undef C64P_SECTION
      C64P_ACC_DONE
      C64P_WAIT_VM(C64P_VM_END);
#undef C64P_VM_END
#undef C64P_VM_RES0
#undef C64P_VM_RES1
    } else {
#define C64P_SECTION 0
#include "conv64p_sched_res_bf16.inc"
#undef C64P_SECTION
#define C64P_SECTION 1
#include "conv64p_sched_res_bf16.inc"
#undef C64P_SECTION
      C64P_ACC_DONE
      C64P_WAIT_VM(C64P_VM_END);
#undef C64P_VM_END
#undef C64P_VM_RES0
#undef C64P_VM_RES1
    }
    const long long t2 = probe ? __builtin_amdgcn_s_memtime() : 0;
    // this tile becomes the previous one
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) fin[b][g] = f32x4{acc[b][4 * g + 0], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]};
    nimg_p = nimg; y0_p = y0; x0_p = x0;
    flip_delta = -flip_delta;
    buf ^= 1;
    if (probe) { pm += t2 - t1; pn += 1; }
    tile += G;
    if (tile >= p.n_tiles) break;
    nimg = nimg_n; y0 = y0_n; x0 = x0_n; f_c = f_n; s_c = s_n;
    const long long t3 = probe ? __builtin_amdgcn_s_memtime() : 0;
    lds_barrier();                                          // every wave's pieces of the next patch have landed; the other buffer is free
    if (probe) pb += __builtin_amdgcn_s_memtime() - t3;
  }
  // ---- the last tile's epilogue
  {
    const long long t3 = probe ? __builtin_amdgcn_s_memtime() : 0;
#define DRAIN_BEGIN __builtin_amdgcn_s_waitcnt(0x0F70);
#define C64P_SECTION 2
    if constexpr (!RES) {
#include "conv64p_sched_plain.inc"
    } else if constexpr (OUT_FMT == 1) {
#include "conv64p_sched_res.inc"
    } else {
#include "conv64p_sched_res_bf16.inc"
    }
#undef C64P_SECTION
    if (probe) pe += __builtin_amdgcn_s_memtime() - t3;
  }
#undef GB
#undef MF_F0
#undef MF_F
#undef MF_XA
#undef MF_XV
#undef SB
#undef C64P_ACC_DONE
#undef LB
#undef C64P_N
#undef WAITRES
#undef DRAIN_BEGIN
  if (probe && lane == 0) {
    g_c64_probe[wave * 8 + 0] = pb; g_c64_probe[wave * 8 + 1] = pm; g_c64_probe[wave * 8 + 2] = pw; g_c64_probe[wave * 8 + 3] = pe;
    g_c64_probe[wave * 8 + 4] = pn;
  }
}

int conv64_launch(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual, const uint16_t* res_split,
                  uint16_t* y_split, float* y_f32, int N, int H, int W, int Hp, int Wp, int relu, int in_fmt, int in_scale_log2,
                  int out_fmt, int out_scale_log2, int* overflow, hipStream_t s) {
  Conv64Params p;
  p.x = x; p.w = w; p.bias = bias; p.residual = residual; p.res_split = res_split; p.y_split = y_split; p.y_f32 = y_f32;
  p.N = N; p.H = H; p.W = W; p.Hp = Hp; p.Wp = Wp; p.relu = relu;
  p.acc_scale = ldexpf(1.0f, -in_scale_log2); p.out_scale = ldexpf(1.0f, out_scale_log2); p.out_fmt = out_fmt; p.overflow = overflow;
  p.n_ty = cdiv(H, C64_TR); p.n_tx = cdiv(W, 32);
  const long long tiles = (long long)p.n_ty * p.n_tx * N;
  if (tiles >= (1ll << 31)) {
    set_error("fgvc_conv64_split_f32: too many tiles");
    return FGVC_ERR_UNSUPPORTED;
  }
  p.n_tiles = (int)tiles;
  p.variant = g_conv64_variant;
  const int grid = (int)(tiles < 256 ? tiles : 256);        // persistent: one workgroup per CU (a wave owns a SIMD's registers)
  // conv64p_kernel: the forms the encoder launches (a block's first convolution; its second with the f32 residual, + f32 out and the f16 + fp8
  // split out, or the bf16 split out alone); every other combination runs on conv64_kernel.  option conv64_variant & 32: conv64k_kernel.
  const bool fits32 = (unsigned long long)N * Hp * Wp * 256ull < (1ull << 32);      // conv64p_kernel: 32-bit byte offsets within a tensor
  if (in_fmt == 1 && !(p.variant & (16 | 32)) && y_split && !res_split && fits32) {
    if (!residual && !y_f32 && out_fmt == 1) { conv64p_kernel<false, false, 1><<<grid, 256, 0, s>>>(p); FGVC_CHECK_LAUNCH("fgvc_conv64_split_f32"); return FGVC_OK; }
    if (residual && y_f32 && out_fmt == 1) { conv64p_kernel<true, true, 1><<<grid, 256, 0, s>>>(p); FGVC_CHECK_LAUNCH("fgvc_conv64_split_f32"); return FGVC_OK; }
    if (residual && !y_f32 && out_fmt == 1) { conv64p_kernel<true, false, 1><<<grid, 256, 0, s>>>(p); FGVC_CHECK_LAUNCH("fgvc_conv64_split_f32"); return FGVC_OK; }
    if (residual && !y_f32 && out_fmt == 0) { conv64p_kernel<true, false, 0><<<grid, 256, 0, s>>>(p); FGVC_CHECK_LAUNCH("fgvc_conv64_split_f32"); return FGVC_OK; }
  }
  const bool res_fits = !residual || (unsigned long long)N * H * W * 256ull < (1ull << 32);      // conv64k_kernel reads the residual as a 32-bit raw buffer
  if (in_fmt == 1 && (p.variant & 32) && res_fits) conv64k_kernel<<<grid, 256, 0, s>>>(p);      // option conv64_variant & 16: round 3's kernel (A/B)
  else if (in_fmt == 1) conv64_kernel<1><<<grid, 256, 0, s>>>(p);
  else conv64_kernel<0><<<grid, 256, 0, s>>>(p);
  FGVC_CHECK_LAUNCH("fgvc_conv64_split_f32");
  return FGVC_OK;
}

}  // namespace fgvc
