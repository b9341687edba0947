// Stride-2 convolutions of the encoder on the bf16 matrix pipe: the 3x3 / stride 2 / pad 1 convolution that opens a
// down-sampling stage and its 1x1 / stride 2 projection (resnet.py:54-76, 288-296), same arithmetic as conv_split.hip
// (activations and weights as (hi, lo) bf16 pairs, hi*hi + hi*lo + lo*hi accumulated in f32) and the same tensors on
// both sides: padded split NHWC in, padded split NHWC and / or dense NHWC f32 out (weights in MFMA-operand order).
// They are 5 % of the trunk's FLOPs; what this kernel removes is the f32 detour around them (MIOpen convolution + bias
// kernel + ReLU/split pass: 0.9 ms of kernel time per 8-frame 480p clip).
//
// A 512-thread workgroup owns 4 rows x 32 columns of OUTPUT pixels x 128 output channels: wave (ct, half) keeps output
// channels 32 ct.. and pixel rows 2 half, 2 half + 1 (two accumulator tiles); two workgroups share a CU.  The input
// patch of one 32-channel chunk at a time is staged by LDS-DMA with the columns DE-INTERLEAVED -- even padded columns,
// then odd ones -- so that the stride-2 walk of a tap becomes 32 consecutive 128-byte entries (a stride-2 read of
// 128-byte pixels is a 2-way bank conflict whatever the swizzle); entries are XOR-swizzled like conv_split's patch.
// Weights are not staged: every wave reads its A operands straight from the L2-resident weight tensor, one tap ahead
// of the multiplies.
// Measured on the layer-2 convolution of a 480p clip (tools/bench_conv_s2.py): 0.16 ms (MIOpen f32: 0.34 + bias and
// ReLU/split passes).  Staging, multiplies and epilogue cost about a third each and barely overlap, and under them lie
// 0.05 ms of weight traffic: every workgroup streams the whole 295 KB weight tensor from L2, 1 GB per launch (operand-
// shaped reads of the [Cout][64] layout -- 32 lines per instruction -- cost 0.20 ms; hence the operand-ordered copy).
// 2-row tiles with three workgroups per CU: 0.18 ms.
#include "common.hpp"

namespace fgvc {

struct ConvS2Params {
  const uint16_t* x;       // padded split NHWC [N][Hp][Wp][Cin/32][64]
  const uint16_t* w;       // [KS*KS][Cin/32][Cout/32][4][64][8]: MFMA-operand order (ops.prepare_conv_s2)
  const float* bias;       // [Cout]
  uint16_t* y_split;       // optional, padded split NHWC [N][Hop][Wop][Cout/32][64]
  float* y_f32;            // optional, dense NHWC f32 [N][Ho][Wo][Cout]
  int N, Hp, Wp, Cin, Cout, Ho, Wo, Hop, Wop, relu;
  int n_ty, n_tx;
  int debug;               // profiling ablations (results WRONG): 1 = no staging, 2 = no MFMA, 4 = no epilogue, 8 = no A loads
  float out_scale;         // out_fmt != 0: the split output stores s_out * y (conv_split.hip: the f16 forms)
  int out_fmt;             // format of y_split: 0 = (hi, lo) bf16, 1 = f16f8, 2 = (h, l) f16, 3 = f16f6
  int* overflow;           // out_fmt != 0: raised when |s_out * y| leaves the f16 range
};


__device__ __forceinline__ void s2_lds_dma_16(const void* src_lane, uint32_t lds_uniform) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src_lane), "s"(lds_uniform) : "memory");
}
__device__ __forceinline__ uint32_t s2_lds_addr(const void* p) {
  return (uint32_t)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
// physical byte offset of 16-byte slot `s` of 128-byte entry `e` (conv_split.hip: conflict-free ds_read_b128 of 32
// consecutive entries from any base)
__device__ __forceinline__ int s2_swz(int e, int s) { return e * 128 + ((s ^ ((e >> 1) & 7)) << 4); }

constexpr int S2_TR = 4;                        // output rows per workgroup
constexpr int S2_NW = 2 * S2_TR;                // waves per workgroup: 4 output-channel tiles x TR / 2 row pairs
constexpr int S2_CPG = 1;                       // input chunks staged together
constexpr int S2_RS = 144;                      // epilogue tile row stride (bytes)

template <int KS>
__global__ __launch_bounds__(64 * S2_NW, 2) void conv_s2_kernel(ConvS2Params p) {
  constexpr int T = KS * KS;
  constexpr int NR = KS == 3 ? 2 * S2_TR + 1 : S2_TR;   // staged input rows
  constexpr int NC = KS == 3 ? 65 : 32;                  // staged entries per row: 33 even + 32 odd columns | 32 odd
  constexpr int NE = NR * NC;
  constexpr int NG = (NE + 7) / 8;                       // 1-KiB DMA pieces per chunk
  constexpr int CHUNKB = NG * 8 * 128;
  constexpr int SMEMB = S2_CPG * CHUNKB > S2_NW * 32 * S2_RS ? S2_CPG * CHUNKB : S2_NW * 32 * S2_RS;
  static_assert(SMEMB <= 79 * 1024, "LDS: two workgroups per CU");
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEMB];
  auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 3, half = wave >> 2;
  const int n = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int nimg = bid / (p.n_ty * p.n_tx);
  bid -= nimg * p.n_ty * p.n_tx;
  const int ty = bid / p.n_tx, tx = bid - ty * p.n_tx;
  const int y0 = ty * S2_TR, x0 = tx * 32;
  const int co_w = blockIdx.y * 128 + ct * 32;           // first output channel of this wave
  const bool active = co_w < p.Cout;                      // wave-uniform (Cout % 32 == 0)
  const int nchunk = p.Cin / 32;
  const size_t pix_bytes_in = (size_t)nchunk * 128;
  const int d_row = lane >> 3, d_slot = lane & 7;

  f32x16 acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  // weights in MFMA-operand order: [tap][chunk][Cout/32][hi k0-15 | hi k16-31 | lo k0-15 | lo k16-31][lane][8]: a wave's A
  // operand is ONE contiguous KiB
  const int n_ct = p.Cout / 32;
  const uint16_t* wlane = p.w + (size_t)(active ? (co_w >> 5) : 0) * 2048 + lane * 8;   // + (t * nchunk + chunk) * n_ct * 2048
  // the lane's four bias vectors, loaded up front (read in the epilogue's row loop every load was followed by its own wait: eight L2 round
  // trips in a row per workgroup)
  f32x4 bias_v[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bias_v[g] = *reinterpret_cast<const f32x4*>(p.bias + (active ? co_w : 0) + 8 * g + 4 * h);

  for (int cg = 0; cg < nchunk; cg += S2_CPG) {
    const int ncg = imin(S2_CPG, nchunk - cg);
    // ---- stage the patch of chunks cg .. cg + ncg - 1
    for (int i = wave; i < ((p.debug & 1) ? 0 : ncg * NG); i += S2_NW) {
      const int cc = i / NG, g = i - cc * NG;
      const int e = imin(g * 8 + d_row, NE - 1);
      const int j = e / NC, m = e - j * NC;
      int prow, pcol;
      if (KS == 3) {
        prow = 2 * y0 + j;
        pcol = 2 * x0 + (m < 33 ? 2 * m : 2 * (m - 33) + 1);
      } else {
        prow = 2 * (y0 + j) + 1;
        pcol = 2 * (x0 + m) + 1;
      }
      if (prow >= p.Hp || pcol >= p.Wp) prow = pcol = 0;             // beyond the buffer: any border pixel (zero)
      const int sl = d_slot ^ ((e >> 1) & 7);
      const unsigned char* src = reinterpret_cast<const unsigned char*>(p.x) +
                                 (((size_t)nimg * p.Hp + prow) * p.Wp + pcol) * pix_bytes_in + (size_t)(cg + cc) * 128 + sl * 16;
      s2_lds_dma_16(src, s2_lds_addr(smem + cc * CHUNKB + g * 1024));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- taps: A operands from global memory one iteration ahead, B operands from the patch
    const int nit = ncg * T;
    bf16x8 a_cur[4], a_nxt[4];
    auto load_a = [&](bf16x8* a, int it) {
      const int cc = it / T, t = it - cc * T;
      const uint16_t* wp = wlane + ((size_t)t * nchunk + (cg + cc)) * n_ct * 2048;
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const bf16x8*>((p.debug & 8) ? wlane + q * 512 : wp + q * 512);   // hi s0, hi s1, lo s0, lo s1
    };
    load_a(a_cur, 0);
    for (int it = 0; it < nit; ++it) {
      if (it + 1 < nit) load_a(a_nxt, it + 1);
      __builtin_amdgcn_sched_barrier(0);        // keep the next tap's loads HERE (hipcc sinks them to their first use)
      const int cc = it / T, t = it - cc * T;
      const int dy = KS == 3 ? t / 3 : 0, dx = KS == 3 ? t - 3 * (t / 3) : 0;
      const int mbase = KS == 3 ? (dx == 0 ? 0 : dx == 1 ? 33 : 1) : 0;
      const unsigned char* patch = smem + cc * CHUNKB;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int r = half * 2 + b;
        const int e = (KS == 3 ? (2 * r + dy) : r) * NC + mbase + n;
        bf16x8 xh[2], xl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          xh[s] = *reinterpret_cast<const bf16x8*>(patch + s2_swz(e, 2 * s + h));
          xl[s] = *reinterpret_cast<const bf16x8*>(patch + s2_swz(e, 4 + 2 * s + h));
        }
        if (p.debug & 2) {
          acc[b][0] += (float)(xh[0][0] + xl[0][0] + xh[1][0] + xl[1][0] + a_cur[0][0] + a_cur[1][0] + a_cur[2][0] + a_cur[3][0]);
          continue;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[s], xh[s], acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[2 + s], xh[s], acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[s], xl[s], acc[b], 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) a_cur[q] = a_nxt[q];
    }
    __syncthreads();                                        // patch free: next chunk group or the epilogue tiles
  }
  if (!active || (p.debug & 4)) return;

  // ---- epilogue: bias (+ ReLU), transposed through a wave-private LDS tile so that every pixel leaves as one 128-byte
  // row: 32 channels of dense f32 and / or one (hi | lo) chunk of the padded split tensor
  unsigned char* tile = smem + wave * (32 * S2_RS);
  const int mv_row = lane >> 3, mv_col = (lane & 7) * 16;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int y = y0 + half * 2 + b;
    if (y >= p.Ho) continue;                                // wave-uniform
    f32x4 v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = bias_v[g];
      v[g] = {acc[b][4 * g + 0] + bv.x, acc[b][4 * g + 1] + bv.y, acc[b][4 * g + 2] + bv.z, acc[b][4 * g + 3] + bv.w};
      if (p.relu) {
        v[g].x = fmaxf(v[g].x, 0.f); v[g].y = fmaxf(v[g].y, 0.f); v[g].z = fmaxf(v[g].z, 0.f); v[g].w = fmaxf(v[g].w, 0.f);
      }
    }
    if (p.y_f32) {
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(tile + n * S2_RS + (8 * g + 4 * h) * 4) = v[g];
      wave_sync();
      unsigned char* dst = reinterpret_cast<unsigned char*>(p.y_f32 + (((size_t)nimg * p.Ho + y) * p.Wo + x0) * p.Cout + co_w);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + mv_row;
        if (x0 + row < p.Wo)
          *reinterpret_cast<uint4*>(dst + (size_t)row * p.Cout * 4 + mv_col) =
              *reinterpret_cast<const uint4*>(tile + row * S2_RS + mv_col);
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
          unsigned char* o = tile + n * S2_RS + (8 * g + 4 * h) * 2;
          *reinterpret_cast<ushort4*>(o) = hv;
          *reinterpret_cast<ushort4*>(o + 64) = lv;
        }
      } else if (p.out_fmt == 3) {                            // f16 + FP6 (common.hpp: split_f16f6_chunk)
        bool ovf = false;
        uint2 hw[4];
        fgvc_i32x4 main6, tail6;
        split_f16f6_chunk(v, p.out_scale, h, hw, main6, tail6, ovf);
        unsigned char* o = tile + n * S2_RS;
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<uint2*>(o + (8 * g + 4 * h) * 2) = hw[g];
        *reinterpret_cast<fgvc_i32x4*>(o + 80 - 16 * h) = main6;
        *reinterpret_cast<fgvc_i32x4*>(o + 112 - 16 * h) = tail6;
        if (__builtin_amdgcn_ballot_w64(ovf && x0 + n < p.Wo) != 0ull && lane == 0) atomicOr(p.overflow, 1);
      } else {                                                // the f16 forms (conv_split.hip): [h 64 B | l8 32 B | h8 32 B] or [h 64 B | l 64 B]
        bool ovf = false;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 hw, lw;
          uint32_t l8, h8;
          split_f16_4(v[g], p.out_scale, hw, l8, h8, lw, ovf);
          unsigned char* o = tile + n * S2_RS;
          *reinterpret_cast<uint2*>(o + (8 * g + 4 * h) * 2) = hw;
          if (p.out_fmt == 1) {
            *reinterpret_cast<uint32_t*>(o + 64 + 8 * g + 4 * h) = l8;
            *reinterpret_cast<uint32_t*>(o + 96 + 8 * g + 4 * h) = h8;
          } else {
            *reinterpret_cast<uint2*>(o + 64 + (8 * g + 4 * h) * 2) = lw;
          }
        }
        if (__builtin_amdgcn_ballot_w64(ovf && x0 + n < p.Wo) != 0ull && lane == 0) atomicOr(p.overflow, 1);
      }
      wave_sync();
      const size_t pix0 = ((size_t)nimg * p.Hop + (y + 1)) * p.Wop + (x0 + 1);
      unsigned char* dst = reinterpret_cast<unsigned char*>(p.y_split) + (pix0 * (p.Cout / 32) + (co_w >> 5)) * 128;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + mv_row;
        if (x0 + row < p.Wo)
          *reinterpret_cast<uint4*>(dst + (size_t)row * p.Cout * 4 + mv_col) =
              *reinterpret_cast<const uint4*>(tile + row * S2_RS + mv_col);
      }
      wave_sync();
    }
  }
}

static int g_conv_s2_debug = 0;
void set_conv_s2_debug(int v) { g_conv_s2_debug = v; }

int conv_s2_launch(const uint16_t* x, const uint16_t* w, const float* bias, uint16_t* y_split, float* y_f32, int N, int Hp,
                   int Wp, int Cin, int Cout, int KS, int Ho, int Wo, int Hop, int Wop, int relu, int out_fmt, int out_scale_log2,
                   int* overflow, hipStream_t s) {
  ConvS2Params p;
  p.out_fmt = out_fmt; p.out_scale = ldexpf(1.0f, out_scale_log2); p.overflow = overflow;
  p.x = x; p.w = w; p.bias = bias; p.y_split = y_split; p.y_f32 = y_f32;
  p.N = N; p.Hp = Hp; p.Wp = Wp; p.Cin = Cin; p.Cout = Cout; p.Ho = Ho; p.Wo = Wo; p.Hop = Hop; p.Wop = Wop; p.relu = relu;
  p.n_ty = cdiv(Ho, S2_TR); p.n_tx = cdiv(Wo, 32);
  p.debug = g_conv_s2_debug;
  dim3 grid(p.n_ty * p.n_tx * N, cdiv(Cout, 128));
  if (KS == 3) conv_s2_kernel<3><<<grid, 64 * S2_NW, 0, s>>>(p);
  else conv_s2_kernel<1><<<grid, 64 * S2_NW, 0, s>>>(p);
  FGVC_CHECK_LAUNCH("fgvc_conv_s2_split_f32");
  return FGVC_OK;
}

}  // namespace fgvc
