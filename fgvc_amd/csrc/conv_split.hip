// Convolution on the bf16 matrix pipe with f32-grade accuracy: the encoder's 3x3 / 1x1 stride-1 convolutions as an
// implicit GEMM over activations and weights that are both stored as (hi, lo) bf16 pairs (x = hi + lo up to 2^-18 |x|).
// Three partial products hi*hi + hi*lo + lo*hi accumulate in f32 (v_mfma_f32_32x32x16_bf16); the dropped lo*lo term is
// 2^-18 relative and sign-random here (activations and weights are unrelated), i.e. below f32 rounding noise.
// MIOpen's f32 Winograd kernel needs 2.4 ms for a 256->256 layer of a 480p clip; the bf16 pipe has ~16x the FLOP rate.
//
// Layouts (all chosen so that every LDS-DMA instruction moves 1 KiB that is contiguous in global memory):
//   activations  x[n][Hp][Wp][C/32][hi 32 ch | lo 32 ch] bf16 -- "padded split NHWC": the image sits at (1,1) inside a
//                zero border (Hp >= 8*ceil(H/8)+2, Wp >= 32*ceil(W/32)+8), so halos and ragged tiles need no predicates;
//   weights      w[tap][Cin/32][Cout][hi 32 ci | lo 32 ci] bf16, BatchNorm folded in on the host;
//   f32 side outputs / residuals: dense NHWC f32 [n][H][W][C] (= a channels_last NCHW tensor: MIOpen reads and writes it
//                without any layout conversion).
// Arithmetic forms (template parameter ARITH = format of the INPUT tensor and of the weights; all accumulate in f32):
//   0  bf16x3: (hi, lo) bf16 operands, hi*hi + hi*lo + lo*hi: three pipe units per product, ~2^-17 per term (above).
//   1  f16f8 : x -> h = f16(s x) (s a per-tensor power of two), l = s x - h (|l| <= 2^-11 |h|), h8 = e4m3(h 2^-a), l8 = e4m3(l 2^b).
//              The main product h*h runs on v_mfma_f32_32x32x16_f16 (exact products); BOTH cross sums of a 32-channel chunk run in ONE
//              v_mfma_scale_f32_32x32x64_f8f6f4: K = 64 = [h8_w . l8_x (32 channels) | l8_w . h8_x (the same 32 channels)], the two
//              32-element scale blocks carrying their own E8M0 scales -- two pipe units per product, ~2^-15.5 per term
//              (tools/sim_conv_formats.py: 1.0-1.7e-5 of max|y| per layer on real activations against 0.7-1.0e-5 for bf16x3; 1e-4 logit
//              on the final features).  Where the instruction takes a block's scale from was probed (tools/micro/probe_fp8_scaled.hip,
//              exact integers): lane (r, hh) holds 32 bytes; its FIRST 16 (registers 0-3) belong to scale block 0, whose scale is read
//              from lane (r, 0), its LAST 16 (registers 4-7) to block 1, scale from lane (r, 1) -- a block is 16 bytes of each lane half,
//              not one lane half's 32.  So registers 0-3 carry the first cross term and registers 4-7 the second:
//              rows keep their 128 bytes per (pixel | output channel, 32-channel chunk), activations [h 64 B | l8 32 B | h8 32 B],
//              weights [h 64 B | h8 32 B | l8 32 B], and lane half hh of either operand reads the 16-byte slots 4 + hh and 6 + hh.
//   2  f16x3 : h = f16(s x), l = f16(s x - h): h*h + h*l + l*h on the f16 pipe, three units, ~2^-22 per term.
//   3  f16f6 : (round 4) as f16f8 with the cross sums on FP6 (e2m3) operands with one E8M0 scale per (pixel | output channel, 32-channel
//              chunk): the same K = 64 instruction retires FP6 operands in HALF the cycles of fp8 (32 instead of 64: the rate of
//              v_mfma_f32_32x32x16_f16 per instruction) -- 1.5 pipe units per product.  Here a lane's scale applies to ITS OWN 32
//              elements (tools/micro/probe_fp6_32x32.hip), so lane half 0 carries h6_w . l6_x and lane half 1 l6_w . h6_x over all 32
//              channels; rows [h 64 B | 4 x 16 B: main pieces, tails + scale bytes] (common.hpp: split_f16f6_chunk), read as the same
//              two ds_read_b128 per operand as f16f8.  tools/sim_conv_formats.py: 1.40e-5 of the features against 1.26e-5 for f16f8.
// The epilogue writes the split output in the format the NEXT layer reads (out_fmt, out_scale), whatever this layer's own arithmetic.
// Work split (widest form): a 512-thread workgroup owns 8 rows x 32 columns of output pixels x 256 output channels; wave (pr, ch) owns
// pixel rows 2pr, 2pr+1 (two 32-pixel MFMA B operands) x channels ch*128..+128 (four 32-channel A operands): 8 accumulator
// tiles.  K loop: for every 32-channel input chunk the (8+2) x 40 pixel patch is staged once (swizzled 128-byte pixel
// rows: ds_read_b128 of a 3x3-shifted row segment is conflict-free for every shift), then the KS*KS taps stream their
// 256 x 32 weight slab through a 3-slot ring; one barrier per tap = per 96 MFMAs per SIMD.
#include "common.hpp"

namespace fgvc {

struct ConvSplitParams {
  const uint16_t* x;
  const uint16_t* w;
  const float* bias;       // [Cout]
  const float* residual;   // optional, dense NHWC f32 [N][H][W][Cout]
  uint16_t* y_split;       // optional, padded split NHWC
  float* y_f32;            // optional, dense NHWC f32
  int N, H, W, Hp, Wp, Cin, Cout, relu;
  int n_ty, n_tx;
  int debug;   // profiling ablations (results WRONG): 1 = patch staged once, 2 = no epilogue, 4 = no MFMA, 8 = s_memtime probe
  float acc_scale;   // accumulator -> convolution value: 1 / (s_x s_w) for the f16 forms, 1 for bf16x3
  float out_scale;   // out_fmt != 0: the split output stores s_out * y
  int out_fmt;       // format of y_split: 0 = (hi, lo) bf16, 1 = f16f8, 2 = (h, l) f16, 3 = f16f6
  int* overflow;     // out_fmt != 0: *overflow |= 1 when |s_out * y| leaves the f16 range (the scale must be re-calibrated)
  const uint16_t* x2;      // optional (the 256-channel 3 x 3 form only): a SECOND input, padded split NHWC of the same H x W geometry with
  const uint16_t* w2;      // Cin2 channels, and the [1][Cin2/32][Cout] rows of a 1 x 1 convolution over it that accumulates into the same
  int Cin2;                // sums: y = conv3x3(x, w) + conv1x1(x2, w2) + bias -- a block's projection shortcut folded into its second convolution
  unsigned char* y_bank;   // optional (Cout = 256 only): the L2-normalised pixels as rows of fgvc_split_f16f6p, [N][H*W][1024 B], INSTEAD of
  int bank_normalize;      // y_split / y_f32 (the trunk's last convolution writes the pair kernel's feature bank itself); 0: rows of the raw values
  int bank_row_bytes;      // 1024: fgvc_split_f16f6p rows; 2048: fgvc_split_f16f6x rows (+ the normalised f32 channels in the second KiB)
};

typedef fgvc_f16x8 f16x8;
typedef fgvc_i32x8 i32x8;
typedef fgvc_i32x4 i32x4;

__device__ __forceinline__ void conv_lds_dma_16(const void* src_lane, uint32_t lds_uniform) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src_lane), "s"(lds_uniform) : "memory");
}
// ... with a wave-uniform base in SGPRs and a 32-bit lane offset (one VGPR instead of a 64-bit address pair)
__device__ __forceinline__ void conv_lds_dma_16s(uint32_t lane_off, const void* base_uniform, uint32_t lds_uniform) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base_uniform), "s"(lds_uniform) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}

constexpr int CV_PW = 40;                      // staged patch width in pixels (34 needed; DMA moves 8 pixels at a time)
constexpr int CV_RINGB = 96 * 1024;             // weight ring: NSLOT slots of TG taps x COT output channels x 128 B

// physical byte offset of 16-byte slot `s` (0..7) of 128-byte row `row`: slot index XOR (row >> 1) & 7
__device__ __forceinline__ int cv_swz(int row, int s) { return row * 128 + ((s ^ ((row >> 1) & 7)) << 4); }

// COT = output channels per workgroup (256 / 128 / 64: wave (pr, ch) owns COT/2 of them = NA 32-channel A operands);
// a pipeline stage = TG taps of one 32-channel input chunk (TG = 1 at COT = 256; a whole row of three taps for the
// narrower layers, which would otherwise synchronise every 24 MFMAs per wave); NSLOT ring slots of TG*COT*128 bytes.
// NWR = waves along the pixel rows (4: an 8-row tile, 512 threads, one workgroup per CU; 2: a 4-row tile, 256 threads,
// small enough in LDS and registers for TWO workgroups per CU -- for the narrow layers, whose short main loop cannot hide
// its own prologue and epilogue, the second workgroup does).
// RPW = pixel rows per wave: 2 in every launched form.  (4 = a "tall" 16 x 32 x 128-channel tile of the f16f8 form, which moves 31 %
// fewer weight + patch bytes per MAC than 8 x 32 x 256; measured 0.67 against 0.49 ms on the 256 -> 256 layer -- register spills in
// the stage loop -- and not instantiated: docs/LAB_NOTES.md, round 3.)
template <int KS, int COT, int TG, int NSLOT, int NWR, bool PINNED = false, int ARITH = 0, int RPW = 2, bool BANK = false, bool X2 = false>
__global__ __launch_bounds__(NWR * 128, NWR == 2 ? 2 : 1) void conv_split_kernel(ConvSplitParams p) {
  static_assert(!X2 || (KS == 3 && TG == 1 && !PINNED), "the second input: extra one-tap stages of the 3 x 3 form with one tap per stage");
  static_assert(!BANK || (COT == 256 && NWR == 4 && RPW == 2 && !PINNED), "the bank epilogue: a pixel's 256 channels in one 8-wave workgroup");
  static_assert(RPW == 2 || RPW == 4, "pixel rows per wave");
  static_assert(!PINNED || RPW == 2, "the all-assembly stage has two pixel rows per wave");
  static_assert(!PINNED || ARITH == 0, "the all-assembly stage is the bf16x3 form");
  static_assert(ARITH >= 0 && ARITH <= 3, "ARITH");
  constexpr int T = KS * KS;
  constexpr int PADK = KS / 2;                  // 1 for 3x3, 0 for 1x1
  constexpr int NW = NWR * 2;                   // waves per workgroup
  constexpr int TROWS = NWR * RPW;              // output rows per workgroup tile
  constexpr int CV_PATCHB = (TROWS + 2) * CV_PW * 128;
  constexpr int NA = COT / 64;                  // A operands (32 output channels each) per wave
  constexpr int SPC = (T + TG - 1) / TG;        // stages per input chunk
  constexpr int CV_WSLOTB = TG * COT * 128;
  constexpr int PPW = TG * COT / 8 / NW;        // 1-KiB DMA pieces per wave per full stage
  static_assert(NSLOT * CV_WSLOTB <= CV_RINGB && (TG * COT / 8) % NW == 0, "ring");
  __shared__ __attribute__((aligned(16))) unsigned char smem[CV_PATCHB + NSLOT * CV_WSLOTB];
  unsigned char* patch = smem;
  unsigned char* wring = smem + CV_PATCHB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pr = wave % NWR, ch = wave / NWR;
  const int n = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;     // raster order over the XCDs round-robin (xcd_remap + column-major tiles: 0.7 % slower, profiles/r03_bv_xcd.log)
  const int nimg = bid / (p.n_ty * p.n_tx);
  bid -= nimg * p.n_ty * p.n_tx;
  const int ty = bid / p.n_tx, tx = bid - ty * p.n_tx;
  const int y0 = ty * TROWS, x0 = tx * 32;
  const int co_base = blockIdx.y * COT;
  const int nchunk = p.Cin / 32;
  const size_t pix_bytes_in = (size_t)nchunk * 128;
  const int nchunk2 = X2 ? p.Cin2 / 32 : 0;                    // chunks of the second input: one extra stage each (its only tap = the centre)
  const size_t pix_bytes_in2 = (size_t)nchunk2 * 128;

  // ---- staging helpers: lane L of a DMA instruction fills LDS bytes [16L, 16L+16) of a 1-KiB piece = 8 rows x 8 slots
  const int d_row = lane >> 3, d_slot = lane & 7;
  auto stage_patch = [&](int chunk) {
    // (8 + 2 PADK) rows x (32 + 8 PADK) pixels in pieces of 8 pixels; piece i -> wave i % 8
    constexpr int ROWS = TROWS + 2 * PADK, PPR = 4 + PADK;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x) + (size_t)chunk * 128;
    for (int i = wave; i < ROWS * PPR; i += NW) {
      const int prow = i / PPR, pc0 = (i - prow * PPR) * 8;
      const int P = prow * CV_PW + pc0 + d_row;                       // patch pixel of this lane
      const int sl = d_slot ^ ((P >> 1) & 7);                         // logical slot that lives at this physical slot
      // (rows below the buffer -- a 16-row tile over a buffer padded for 8-row tiles -- read its last row: zero border)
      const size_t gpix = ((size_t)nimg * p.Hp + imin(y0 + 1 - PADK + prow, p.Hp - 1)) * p.Wp + (x0 + 1 - PADK + pc0 + d_row);
      conv_lds_dma_16(xb + gpix * pix_bytes_in + sl * 16, lds_addr(patch + (prow * CV_PW + pc0) * 128));
    }
  };
  // The patches of chunks 1.. come through REGISTERS: global -> VGPRs is issued one stage before the chunk boundary and
  // lands while that stage multiplies; at the boundary only ds_writes remain (an LDS-DMA there exposed the whole memory
  // latency once per chunk, 15 % of the kernel; a second LDS patch buffer does not fit beside the weight ring).
  constexpr int NPIECE = (TROWS + 2 * PADK) * (4 + PADK), MAXP = (NPIECE + NW - 1) / NW;
  static_assert(MAXP <= 12, "patch pieces per wave");
  // seven named registers, not an array (as `uint4 pre[7]` touched from two lambdas hipcc left it in scratch memory), and
  // loaded by inline assembly: for loads it knows about, hipcc's own vmcnt wait before the ds_writes is computed without
  // the inline-assembly DMAs and comes out as vmcnt(0) -- draining the weight DMAs issued a moment before, i.e. exactly
  // the memory latency this was meant to hide.  The explicit counted wait below leaves those DMAs in flight.
  uint4 pre0 = {}, pre1 = {}, pre2 = {}, pre3 = {}, pre4 = {}, pre5 = {}, pre6 = {}, pre7 = {}, pre8 = {}, pre9 = {}, pre10 = {}, pre11 = {};
  const unsigned char* xb0 = reinterpret_cast<const unsigned char*>(p.x);
#define CV_PIECE_GEOM(J)                                                                              \
  const int pi_ = imin(wave + NW * (J), NPIECE - 1);                                                   \
  const int prow_ = pi_ / (4 + PADK), pc0_ = (pi_ - prow_ * (4 + PADK)) * 8;
  // piece J of this wave = 8 pixels of patch row prow_: a wave-uniform base (scalar) + one lane offset.  The swizzle key of the lane's
  // patch pixel, ((prow_ * 40 + pc0_ + d_row) >> 1) & 7, is (d_row >> 1) ^ 4 * ((prow_ + pc0_ / 8) & 1): one lane constant, XORed with 64
  // bytes for every other piece -- no per-piece 64-bit lane addresses (hipcc hoisted twelve of them out of the stage loop and spilled them)
  const uint32_t pf_lane_off = (uint32_t)(d_row * (int)pix_bytes_in + ((d_slot ^ (d_row >> 1)) << 4));
  const uint32_t pf_lane_off2 = (uint32_t)(d_row * (int)pix_bytes_in2 + ((d_slot ^ (d_row >> 1)) << 4));
  const unsigned char* xb2 = reinterpret_cast<const unsigned char*>(p.x2);
  // (CHUNK counts through both inputs: chunks nchunk .. nchunk + nchunk2 - 1 are the second input's)
#define CV_PREFETCH(J, CHUNK)                                                                         \
  if ((J) < MAXP) {                                                                                   \
    CV_PIECE_GEOM(J)                                                                                  \
    const size_t g_ = ((size_t)nimg * p.Hp + imin(y0 + 1 - PADK + prow_, p.Hp - 1)) * p.Wp + (x0 + 1 - PADK + pc0_); \
    const bool second_ = X2 && (CHUNK) >= nchunk;                                                     \
    const unsigned char* b_ = second_ ? xb2 + (size_t)((CHUNK) - nchunk) * 128 + g_ * pix_bytes_in2    \
                                      : xb0 + (size_t)(CHUNK) * 128 + g_ * pix_bytes_in;              \
    const uint32_t o_ = (second_ ? pf_lane_off2 : pf_lane_off) ^ (uint32_t)(((prow_ + (pc0_ >> 3)) & 1) << 6); \
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(pre##J) : "v"(o_), "s"(b_) : "memory");   \
  }
#define CV_COMMIT(J)                                                                                  \
  if ((J) < MAXP && wave + NW * (J) < NPIECE) {                                                        \
    CV_PIECE_GEOM(J)                                                                                  \
    /* the address is formed HERE (volatile): as loop-invariant C++ hipcc kept one address register per piece across the stage loop */ \
    uint32_t a_;                                                                                      \
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(a_) : "s"(lds_addr(patch) + (uint32_t)((prow_ * CV_PW + pc0_) * 128)), "v"(lane16)); \
    const i32x4 t_ = __builtin_bit_cast(i32x4, pre##J);                                               \
    asm volatile("ds_write_b128 %0, %1" ::"v"(a_), "v"(t_) : "memory");                               \
  }
  const uint32_t lane16 = (uint32_t)lane * 16u;
  const uint32_t w_lane_off = (uint32_t)(d_row * 128 + ((d_slot ^ (d_row >> 1)) << 4));
  // conv_debug & 256 (experiment, the f16f8 / f16f6 stage of the 8-wave forms): ALL weight DMAs of a stage are issued by the older wave of
  // every SIMD (waves 0 .. NW/2 - 1, 2 PPW pieces each) -- the wave that wins the arbitration for the matrix pipe and then idles at the
  // barrier -- and none by the younger
  const bool dma_older = (ARITH == 1 || ARITH == 3) && NW == 8 && (p.debug & 256) != 0;
  const bool dma_issuer = !dma_older || wave < NW / 2;
  // one 1-KiB piece (8 output channels of one tap) of the weight slab of stage q = chunk * SPC + tap group
  auto stage_weight_piece = [&](int q, int j) {
    const int chunk = q / SPC, tap0 = (q - chunk * SPC) * TG;
    unsigned char* dst = wring + (q % NSLOT) * CV_WSLOTB;
    const int piece = dma_older ? wave * (2 * PPW) + j : wave * PPW + j;   // always PPW pieces per wave (a short last group
    const int tg = piece / (COT / 8), c0 = (piece - tg * (COT / 8)) * 8;     // re-reads its last tap): the vmcnt
    const int tap = imin(tap0 + tg, T - 1);                            // arithmetic stays fixed
    // wave-uniform base (tap, chunk, first output channel of the piece) + lane offset: row d_row of the piece, slot d_slot swizzled by
    // the row's key ((c0 + d_row) >> 1) & 7 = (d_row >> 1) | ((c0 >> 1) & 4)  (c0 is a multiple of 8): one lane constant, XORed with 64
    // for every other piece
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.w) +
                              ((((size_t)tap * nchunk + chunk) * p.Cout + co_base) + c0) * 128;
    if constexpr (X2) {                                              // a stage of the second input: chunk q - nchunk * SPC of its one tap
      if (q >= nchunk * SPC)
        wb = reinterpret_cast<const unsigned char*>(p.w2) + (((size_t)(q - nchunk * SPC) * p.Cout + co_base) + c0) * 128;
    }
    uint32_t wo_;                                                    // (volatile: not another lane constant held across the stage loop)
    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(wo_) : "s"((uint32_t)((c0 & 8) << 3)), "v"(w_lane_off));
    conv_lds_dma_16s(wo_, wb, lds_addr(dst + (tg * COT + c0) * 128));
  };
  auto stage_weights = [&](int q) {
    if (!dma_issuer) return;
#pragma unroll
    for (int j = 0; j < 2 * PPW; ++j)
      if (j < PPW || dma_older) stage_weight_piece(q, j);
  };

  f32x16 acc[NA][RPW];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < RPW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int n_stage = nchunk * SPC + nchunk2;
  // static wave priorities (experiment switches, results unchanged): the two waves of a SIMD are arbitrated by priority, then age
  if (NW == 8) {
    if ((p.debug & 32) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if ((p.debug & 64) && wave < 4) __builtin_amdgcn_s_setprio(1);
    if ((p.debug & 128) && wave >= 4) __builtin_amdgcn_s_setprio(3);
  }
  constexpr int LA = NSLOT - 1;                  // stages in flight ahead of the one being multiplied
  stage_patch(0);
#pragma unroll
  for (int i = 0; i < LA; ++i)
    if (i < n_stage) stage_weights(i);
  const bool timing = (p.debug & 8) && blockIdx.x == 300 && blockIdx.y == 0;   // s_memtime probe of one workgroup
  long long t_wait = 0, t_issue = 0, t_mma = 0, t_bound = 0, t_start = 0;
  if (timing) t_start = __builtin_amdgcn_s_memtime();
  for (int q = 0; q < n_stage; ++q) {
    int chunk = q / SPC, sg = q - chunk * SPC;
    const bool second = X2 && q >= nchunk * SPC;   // a stage of the second input: its own chunk, the centre tap's pixels
    if (second) { chunk = nchunk + (q - nchunk * SPC); sg = T / 2; }
    long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (timing) c0 = __builtin_amdgcn_s_memtime();
    if (q + LA - 1 < n_stage) {
      if (dma_older) {
        if (dma_issuer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * 2 * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * PPW) : "memory");   // all but the stages after q have landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();
    if (timing) c1 = __builtin_amdgcn_s_memtime();
    const bool last_of_chunk = (second || sg == SPC - 1) && q + 1 < n_stage && (p.debug & 1) == 0;
    if (last_of_chunk) {                            // next chunk's patch: lands while this stage multiplies.  Issued BEFORE
      // this stage's weight DMAs: hipcc guards the reuse of these registers with a vmcnt wait that knows nothing of the
      // inline-assembly DMAs and would otherwise wait for the ones issued a moment ago
      CV_PREFETCH(0, chunk + 1) CV_PREFETCH(1, chunk + 1) CV_PREFETCH(2, chunk + 1) CV_PREFETCH(3, chunk + 1)
      CV_PREFETCH(4, chunk + 1) CV_PREFETCH(5, chunk + 1) CV_PREFETCH(6, chunk + 1) CV_PREFETCH(7, chunk + 1)
      CV_PREFETCH(8, chunk + 1) CV_PREFETCH(9, chunk + 1) CV_PREFETCH(10, chunk + 1) CV_PREFETCH(11, chunk + 1)
      __builtin_amdgcn_sched_barrier(0);            // keep the loads ahead of this stage's MFMAs
    }
    // the weight DMAs of stage q+LA (its slot held stage q-1, which every wave has finished) are issued between the MFMA
    // blocks below: issued up front they kept the matrix pipe idle for 500-900 cycles per stage (s_memtime probe)
    const bool stage_more = q + LA < n_stage;
    // (tried: let one wave of each SIMD issue its DMAs up front to shift it against its partner -- 10 % slower)
    if (timing) c2 = __builtin_amdgcn_s_memtime();
    const unsigned char* wslot = wring + (q % NSLOT) * CV_WSLOTB;
    if constexpr (PINNED && TG == 1 && NA == 4 && KS == 3) {
      // ---- the widest form's stage as volatile inline assembly in a fixed order (reads, counted waits, MFMAs).  Left to the compiler,
      // every weight fragment is read right before its first use (2-4 MFMAs of cover for a ~130-cycle LDS round trip; 24 registers for
      // fragments).  Here the pixel fragments of a K-16 step (16 registers) are read during the step before and the weight fragments
      // of channel block a + 2 right after block a's six MFMAs have issued: a block of cover for every read, 48 fragment registers --
      // which fit since the weight DMAs take a scalar base and one lane-offset register instead of 64-bit address pairs.  The LDS
      // returns a wave's reads in order, so the waits are counted: when a fragment is needed, only the reads issued after it may still
      // be in flight.  Bit-identical results; -1.6 % stand-alone, -0.7 % of the encoder (tools/experiments/try_conv_pinned.py): the partner wave
      // already hid most of the latency.
      const int tap = sg;
      const int dy = tap / KS, dx = tap - dy * KS;
      const int ka = (n >> 1) & 7;
      const uint32_t wbase = lds_addr(wslot) + (uint32_t)((ch * (COT / 2) + n) * 128);
      uint32_t pb[2], kb[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int P = (2 * pr + b + dy) * CV_PW + n + dx;
        pb[b] = lds_addr(patch) + (uint32_t)(P * 128);
        kb[b] = (uint32_t)((P >> 1) & 7);
      }
      bf16x8 A_h[2], A_l[2], B_h[2][2], B_l[2][2];
#define CVR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define CVM(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
      auto read_A = [&](int buf, int a, int st) {
        const uint32_t ah_ = wbase + (uint32_t)(((2 * st + h) ^ ka) << 4), al_ = wbase + (uint32_t)(((4 + 2 * st + h) ^ ka) << 4);
        switch (a) {
          case 0: CVR(A_h[buf], ah_, 0); CVR(A_l[buf], al_, 0); break;
          case 1: CVR(A_h[buf], ah_, 4096); CVR(A_l[buf], al_, 4096); break;
          case 2: CVR(A_h[buf], ah_, 8192); CVR(A_l[buf], al_, 8192); break;
          default: CVR(A_h[buf], ah_, 12288); CVR(A_l[buf], al_, 12288); break;
        }
      };
      auto read_B = [&](int buf, int st) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          CVR(B_h[buf][b], pb[b] + (((uint32_t)(2 * st + h) ^ kb[b]) << 4), 0);
          CVR(B_l[buf][b], pb[b] + (((uint32_t)(4 + 2 * st + h) ^ kb[b]) << 4), 0);
        }
      };
      read_B(0, 0);
      read_A(0, 0, 0);
      read_A(1, 1, 0);
#pragma unroll
      for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int ab = a & 1;
          if (a == 3 && st == 0) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
          else if (a == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
          CVM(acc[a][0], A_h[ab], B_h[st][0]);
          CVM(acc[a][1], A_h[ab], B_h[st][1]);
          CVM(acc[a][0], A_h[ab], B_l[st][0]);
          CVM(acc[a][1], A_h[ab], B_l[st][1]);
          CVM(acc[a][0], A_l[ab], B_h[st][0]);
          CVM(acc[a][1], A_l[ab], B_h[st][1]);
          if (a < 2) read_A(ab, a + 2, st);
          else if (st == 0) {
            read_A(ab, a - 2, 1);
            if (a == 2) read_B(1, 1);
          }
          if (stage_more && (a & 1)) {
            constexpr int NIT = 4;
            const int it = st * 2 + (a >> 1);
#pragma unroll
            for (int j = 0; j < PPW; ++j)
              if (j * NIT / PPW == it) stage_weight_piece(q + LA, j);
          }
        }
      }
#undef CVR
#undef CVM
    } else if constexpr (ARITH == 1 || ARITH == 3) {
      // ---- f16f8 / f16f6 (the X block: fp8 with two fixed scales, or FP6 with the scale bytes that travel in the rows).  Per tap and wave: NA "F" blocks (output-channel tile a: the two K-16 steps of the f16 main product against both
      // pixel rows, 4 MFMAs = 128 pipe cycles) and NA "X" blocks (the K-64 fp8 MFMA of both cross sums against both pixel rows, 2 MFMAs
      // = 128 cycles).  Left to itself hipcc keeps ONE set of weight-fragment registers and waits for every read right in front of the
      // MFMAs that use it (ds_read, s_waitcnt lgkmcnt(0), two MFMAs, ds_read, ...): the whole LDS round trip of every block lies open
      // and only the SIMD's other wave covers it (0.50 ms for the 256 -> 256 layer of an 8-frame 480p clip).  Here the blocks of a
      // stage are software-pipelined by hand: the fragments of block g + 2 are read right after block g's MFMAs have issued, the pixel
      // fragments of a phase during the phase before, and scheduling fences keep hipcc from folding the order back; its own counted
      // lgkmcnt waits then come out right (the LDS returns a wave's reads in order).
      static_assert(T % TG == 0, "whole tap groups");
      const uint32_t sa_ = h ? (uint32_t)(127 - F8_BW) : (uint32_t)(127 + F8_AW), sb_ = h ? (uint32_t)(127 + F8_AX) : (uint32_t)(127 - F8_BX);
      const int scale_a = (int)(sa_ * 0x01010101u), scale_b = (int)(sb_ * 0x01010101u);   // E8M0 scales of this lane's block (127 + log2), every byte
      constexpr int NRP = RPW / 2;                    // pixel-row pairs of this wave: a tap's blocks run pair by pair (the weight fragments
                                                      // are read once per pair, the pixel fragments of ONE pair are live at a time)
      constexpr int NB = 2 * NA;                      // blocks per (tap, row pair)
      constexpr int NBLK = TG * NRP * NB;             // blocks per stage
      constexpr int LOOK = RPW == 4 ? 1 : 2;          // weight fragments are read LOOK blocks ahead, into LOOK buffers (the tall tile's
                                                      // twelve patch-prefetch pieces leave registers for one buffer only)
      f16x8 af[LOOK][2];                              // [buffer][K-16 step]: F block g lives in buffer g % LOOK
      i32x8 ax[LOOK];                                 // X block g lives in buffer g % LOOK
      f16x8 bf[2][2];                                 // [pixel row][K-16 step] of the current tap
      i32x8 bx[2];                                    // [pixel row] fp8 operand of the current tap
      auto w_row = [&](int g) {                       // LDS row of block g's weight fragment for this lane
        const int tg = g / (NRP * NB), a = (g % NB) % NA;
        return wslot + tg * COT * 128 + (ch * (COT / 2) + a * 32 + n) * 128;
      };
      auto kx = [&](int g) { return ((ch * (COT / 2) + ((g % NB) % NA) * 32 + n) >> 1) & 7; };
      auto load_block = [&](int g) {
        if (g >= NBLK) return;
        const unsigned char* r = w_row(g);
        const int k = kx(g);
        if ((g % NB) < NA) {
          af[g % LOOK][0] = *reinterpret_cast<const f16x8*>(r + (((0 + h) ^ k) << 4));
          af[g % LOOK][1] = *reinterpret_cast<const f16x8*>(r + (((2 + h) ^ k) << 4));
        } else {
          const i32x4 u = *reinterpret_cast<const i32x4*>(r + (((4 + h) ^ k) << 4)), v = *reinterpret_cast<const i32x4*>(r + (((6 + h) ^ k) << 4));
          ax[g % LOOK] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
        }
      };
      auto load_bf = [&](int u_) {                    // u_ = tg * NRP + row pair
        const int tg = u_ / NRP, rp = u_ % NRP;
        const int tap = sg * TG + tg, dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int P = (RPW * pr + 2 * rp + b + dy) * CV_PW + n + dx;
          bf[b][0] = *reinterpret_cast<const f16x8*>(patch + cv_swz(P, 0 + h));
          bf[b][1] = *reinterpret_cast<const f16x8*>(patch + cv_swz(P, 2 + h));
        }
      };
      auto load_bx = [&](int u_) {
        const int tg = u_ / NRP, rp = u_ % NRP;
        const int tap = sg * TG + tg, dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int P = (RPW * pr + 2 * rp + b + dy) * CV_PW + n + dx;
          const i32x4 u = *reinterpret_cast<const i32x4*>(patch + cv_swz(P, 4 + h)), v = *reinterpret_cast<const i32x4*>(patch + cv_swz(P, 6 + h));
          bx[b] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
        }
      };
      load_bf(0);
#pragma unroll
      for (int g = 0; g < LOOK; ++g) load_block(g);
#pragma unroll
      for (int g = 0; g < NBLK; ++g) {
        const int u_ = g / NB, kb = g % NB, a = kb % NA, r0 = 2 * (u_ % NRP);
        __builtin_amdgcn_sched_barrier(0);
        if (kb < NA) {
#pragma unroll
          for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][r0 + b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[g % LOOK][s_], bf[b][s_], acc[a][r0 + b], 0, 0, 0);
        } else {
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            if constexpr (ARITH == 3)
              acc[a][r0 + b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ax[g % LOOK], bx[b], acc[a][r0 + b], 2, 2, 0, ax[g % LOOK][6], 0, bx[b][6]);
            else
              acc[a][r0 + b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ax[g % LOOK], bx[b], acc[a][r0 + b], 0, 0, 0, scale_a, 0, scale_b);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        load_block(g + LOOK);                         // into the buffer block g just left
        if (kb == (NA > 1 ? 1 : 0)) load_bx(u_);      // the pair's fp8 pixel operands: needed NA - 1 blocks from here
        if (kb == NB - 2 + (NA > 1 ? 0 : 1) && u_ + 1 < TG * NRP) load_bf(u_ + 1);   // the next (tap, pair)'s f16 pixel operands, during this one's X phase
        if (stage_more) {                             // this block's share of the PPW weight pieces, behind its MFMAs
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < PPW; ++j)
            if (!dma_older && j * NBLK / PPW == g) stage_weight_piece(q + LA, j);
          if (dma_older && dma_issuer) {
#pragma unroll
            for (int j = 0; j < 2 * PPW; ++j)
              if (j * NBLK / (2 * PPW) == g) stage_weight_piece(q + LA, j);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    } else
#pragma unroll
    for (int tg = 0; tg < TG; ++tg) {
      const int tap = sg * TG + tg;
      if (tap >= T) break;                          // wave-uniform: short last tap group
      const int dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
      for (int s = 0; s < 2; ++s) {                 // two k16 steps per 32-channel chunk
        bf16x8 ah[NA], al[NA], bh[RPW], bl[RPW];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          const int co = ch * (COT / 2) + a * 32 + n;
          ah[a] = *reinterpret_cast<const bf16x8*>(wslot + tg * COT * 128 + cv_swz(co, 2 * s + h));
          al[a] = *reinterpret_cast<const bf16x8*>(wslot + tg * COT * 128 + cv_swz(co, 4 + 2 * s + h));
        }
#pragma unroll
        for (int b = 0; b < RPW; ++b) {
          const int P = (RPW * pr + b + dy) * CV_PW + n + dx;
          bh[b] = *reinterpret_cast<const bf16x8*>(patch + cv_swz(P, 2 * s + h));
          bl[b] = *reinterpret_cast<const bf16x8*>(patch + cv_swz(P, 4 + 2 * s + h));
        }
        if ((p.debug & 4) == 0) {
#pragma unroll
          for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < RPW; ++b) {
              if constexpr (ARITH == 2) {             // (h, l) f16 operands: the same three products on the f16 form
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[a]), __builtin_bit_cast(f16x8, bh[b]), acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[a]), __builtin_bit_cast(f16x8, bl[b]), acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[a]), __builtin_bit_cast(f16x8, bh[b]), acc[a][b], 0, 0, 0);
              } else {
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
              }
            }
        }
        if (stage_more) {                           // this iteration's share of the PPW weight pieces, behind its MFMAs
          constexpr int NIT = TG * 2;
          const int it = tg * 2 + s;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < PPW; ++j)
            if (j * NIT / PPW == it) stage_weight_piece(q + LA, j);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (timing) {
      asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[NA - 1][RPW - 1][15]));
      c3 = __builtin_amdgcn_s_memtime();
      t_wait += c1 - c0; t_issue += c2 - c1; t_mma += c3 - c2;
    }
    if (last_of_chunk) {                            // chunk boundary (the registers live only inside this iteration)
      if (dma_older && q + LA < n_stage) {
        if (dma_issuer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else
      if (q + LA < n_stage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");   // prefetch landed; this stage's
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                               // weight DMAs stay in flight
      lds_barrier();                                // everyone is done reading this chunk's patch
      CV_COMMIT(0) CV_COMMIT(1) CV_COMMIT(2) CV_COMMIT(3) CV_COMMIT(4) CV_COMMIT(5) CV_COMMIT(6) CV_COMMIT(7)   // visible after the
      CV_COMMIT(8) CV_COMMIT(9) CV_COMMIT(10) CV_COMMIT(11)
                                                                                                  // next stage's barrier
      if (timing) t_bound += __builtin_amdgcn_s_memtime() - c3;
    }
  }
  if (timing) {
    if (lane == 0) {
      long long* o = reinterpret_cast<long long*>(p.y_split) + wave * 8;
      o[0] = t_wait; o[1] = t_issue; o[2] = t_mma; o[3] = t_bound; o[4] = __builtin_amdgcn_s_memtime() - t_start; o[5] = n_stage;
    }
    return;
  }

  // ---- epilogue: + bias [+ residual] [ReLU].  The C layout puts the pixel on the lane and four consecutive output
  //      channels in a register group: stored straight from there every instruction would scatter 64 pieces of 8-16
  //      bytes (measured: 23 % of the kernel).  Instead each wave transposes its 32-pixel x CW-channel tile through a
  //      private LDS region (rows padded by 16 B: conflict-free both ways) and moves whole pixel rows -- CW*4 contiguous
  //      bytes -- to and from global memory; the residual tile comes in the same way.
  if (p.debug & 2) {
    float sink = 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a) sink += acc[a][0][0] + acc[a][1][5];
    if (sink == 123.456f) p.y_split[0] = 1;
    return;
  }
  constexpr int CW = COT / 2;                     // output channels of this wave
  constexpr int RB = CW * 4;                      // bytes of one pixel row of the tile (f32, or hi+lo bf16)
  constexpr int RS = RB + 16;                     // padded LDS row stride
  constexpr int LPR = RB / 16;                    // lanes that move one row (16 B each)
  constexpr int RPI = 64 / LPR;                   // rows per wave instruction
  static_assert(NW * 32 * RS <= CV_PATCHB + NSLOT * CV_WSLOTB, "epilogue staging");
  __syncthreads();                                // patch and weight ring are dead: reuse them
  unsigned char* tile = smem + wave * (32 * RS);
  const int co_w = co_base + ch * CW;             // first output channel of this wave
  const int mv_row = lane / LPR, mv_col = (lane % LPR) * 16;
  auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  if constexpr (BANK) {
    {
      // ---- the feature bank straight from the accumulators (round 4): what normalize_f16f6p_kernel makes of this convolution's dense
      // f32 output, bit for bit, without that output ever reaching memory.  A workgroup holds all 256 channels of its pixels: wave
      // (pr, ch) channels 128 ch ..+128 of pixel rows 2 pr, 2 pr + 1.  Per pixel row: (1) the sum of squares in that kernel's own
      // association -- lane L of its wave-per-pixel layout holds channels 4 L ..+4 = here group (ch, a, g, h) with L = 32 ch + 8 a +
      // 2 g + h, and its xor-shuffle tree adds partners 32, 16, 8, 4, 2, 1 apart: the other wave's partial through LDS, then a ^ 2,
      // a ^ 1, g ^ 2, g ^ 1 in registers, then the other lane half; (2) x = v / |v|, h = f16(256 x), residual 256 (256 x - h);
      // (3) a scale block of the row format = channels 64 v + 16 m + 8 hi + i = groups a = 2 (v & 1) + (m >> 1), g = hi + 2 (m & 1) of
      // BOTH lane halves: the halves trade registers (v_permlane32_swap) so that lane (n, 0) holds the block's 32 h values and
      // lane (n, 1) its 32 residuals, one v_cvt_scalef32_2xpk16_fp6_f32 each (position 2 e <- S0[e], 2 e + 1 <- S1[e], round to nearest
      // even from f32: tools/micro/probe_cvt_fp6_f32.hip); (4) the pair of waves assembles the 1-KiB rows in LDS and stores them whole.
      constexpr int BK_RS = 1024 + 16;
      static_assert(NWR * 32 * BK_RS <= CV_PATCHB + NSLOT * CV_WSLOTB, "bank staging");
      unsigned char* rows = smem + pr * (32 * BK_RS);
      float* part = reinterpret_cast<float*>(rows);                 // [ch][a * 4 + g][lane]: aliases the rows, used before them
#pragma unroll
      for (int b = 0; b < RPW; ++b) {
        const int y = y0 + RPW * pr + b;
        const bool row_ok = y < p.H;                                // wave-uniform, the same for both waves of a pair; barriers are taken by all
        const size_t fpix0 = ((size_t)nimg * p.H + imin(y, p.H - 1)) * p.W + x0;
        if (p.residual) {
          const unsigned char* src = reinterpret_cast<const unsigned char*>(p.residual + fpix0 * p.Cout + co_w);
#pragma unroll
          for (int i = 0; i < 32 / RPI; ++i) {
            const int row = i * RPI + mv_row;
            if (x0 + row < p.W)
              *reinterpret_cast<uint4*>(tile + row * RS + mv_col) = *reinterpret_cast<const uint4*>(src + (size_t)row * p.Cout * 4 + mv_col);
          }
          wave_sync();
        }
        f32x4 v[NA][4];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int cw = a * 32 + 8 * g + 4 * h;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + co_w + cw);
            if constexpr (ARITH == 0)
              v[a][g] = {acc[a][b][4 * g + 0] + bv.x, acc[a][b][4 * g + 1] + bv.y, acc[a][b][4 * g + 2] + bv.z, acc[a][b][4 * g + 3] + bv.w};
            else
              v[a][g] = {fmaf(acc[a][b][4 * g + 0], p.acc_scale, bv.x), fmaf(acc[a][b][4 * g + 1], p.acc_scale, bv.y),
                         fmaf(acc[a][b][4 * g + 2], p.acc_scale, bv.z), fmaf(acc[a][b][4 * g + 3], p.acc_scale, bv.w)};
            if (p.residual) v[a][g] += *reinterpret_cast<const f32x4*>(tile + n * RS + cw * 4);
            if (p.relu) {
              v[a][g].x = fmaxf(v[a][g].x, 0.f); v[a][g].y = fmaxf(v[a][g].y, 0.f);
              v[a][g].z = fmaxf(v[a][g].z, 0.f); v[a][g].w = fmaxf(v[a][g].w, 0.f);
            }
          }
        __syncthreads();                                            // (1) the private residual tiles are read: the region becomes the pairs' buffers
        float pp[NA][4];
        {
#pragma clang fp contract(off)                                       // four rounded squares and three adds, as normalize_nhwc_kernel compiles
#pragma unroll
          for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float xx = v[a][g].x * v[a][g].x, yy = v[a][g].y * v[a][g].y, zz = v[a][g].z * v[a][g].z, ww = v[a][g].w * v[a][g].w;
              pp[a][g] = ((xx + yy) + zz) + ww;
              part[(ch * 16 + a * 4 + g) * 64 + lane] = pp[a][g];
            }
        }
        __syncthreads();                                            // (2)
        float ss;
        {
          float t[NA][4];
#pragma unroll
          for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) t[a][g] = pp[a][g] + part[((ch ^ 1) * 16 + a * 4 + g) * 64 + lane];       // partner 32 lanes away
          float u[2][4], w4[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) { u[0][g] = t[0][g] + t[2][g]; u[1][g] = t[1][g] + t[3][g]; }                // 16
#pragma unroll
          for (int g = 0; g < 4; ++g) w4[g] = u[0][g] + u[1][g];                                                 // 8
          const float z0 = w4[0] + w4[2], z1 = w4[1] + w4[3];                                                     // 4
          const float s1 = z0 + z1;                                                                               // 2
          ss = s1 + __shfl_xor(s1, 32);                                                                           // 1: the other lane half
        }
        __syncthreads();                                            // (3) the partials are read: rows may be written
        const float inv = p.bank_normalize ? 1.0f / fmaxf(sqrtf(ss), 1e-12f) : 1.0f;
        unsigned char* row = rows + n * BK_RS;
#pragma unroll
        for (int vl = 0; vl < 2; ++vl)
#pragma unroll
          for (int hi = 0; hi < 2; ++hi) {
            unsigned A_[16], B_[16];
            float mh = 0.f, ml = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const int a = 2 * vl + (m >> 1), g = hi + 2 * (m & 1);          // (every (a, g) group belongs to exactly one block)
              f32x4 x = v[a][g];
              x *= inv;
              typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
              f16x4 hv;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const float xs = x[k] * 256.f;
                const _Float16 hh_ = (_Float16)xs;
                hv[k] = hh_;
                const float fh = (float)hh_, fl = (xs - fh) * 256.f;
                mh = fmaxf(mh, fabsf(fh)); ml = fmaxf(ml, fabsf(fl));
                A_[4 * m + k] = __builtin_bit_cast(unsigned, fh);
                B_[4 * m + k] = __builtin_bit_cast(unsigned, fl);
              }
              *reinterpret_cast<f16x4*>(row + 2 * (128 * ch + 32 * a + 8 * g + 4 * h)) = hv;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {        // upper lanes of A_ <-> lower lanes of B_: A_ = the values of lane half 0, B_ of half 1
              const auto t2 = __builtin_amdgcn_permlane32_swap(A_[j], B_[j], false, false);
              A_[j] = t2[0]; B_[j] = t2[1];
            }
            const auto tm = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mh), __builtin_bit_cast(unsigned, ml), false, false);
            const unsigned mb = tm[0] > tm[1] ? tm[0] : tm[1];     // (as unsigned integers: see split_f16f6_chunk)
            int sexp = (int)(mb >> 23) - 129 + ((mb & 0x7fffffu) > 0x700000u ? 1 : 0);     // m / 2^s in (3.75, 7.5]: n6_scale_exp
            sexp = (mb == 0u || sexp < -40) ? -40 : sexp;
            fgvc_i32x16 S0, S1;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              S0[4 * m + 0] = (int)A_[4 * m + 0]; S0[4 * m + 1] = (int)A_[4 * m + 2]; S0[4 * m + 2] = (int)B_[4 * m + 0]; S0[4 * m + 3] = (int)B_[4 * m + 2];
              S1[4 * m + 0] = (int)A_[4 * m + 1]; S1[4 * m + 1] = (int)A_[4 * m + 3]; S1[4 * m + 2] = (int)B_[4 * m + 1]; S1[4 * m + 3] = (int)B_[4 * m + 3];
            }
            fgvc_i32x6 c6;
            const unsigned sc_bits = (unsigned)(sexp + 127) << 23;
            asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(c6) : "v"(S0), "v"(S1), "v"(sc_bits));
            const int vg = 2 * ch + vl;
            *reinterpret_cast<i32x4*>(row + (h ? 704 : 512) + 32 * vg + 16 * hi) = i32x4{c6[0], c6[1], c6[2], c6[3]};
            *reinterpret_cast<uint2*>(row + (h ? 832 : 640) + 32 * (vg >> 1) + 16 * hi + 8 * (vg & 1)) = uint2{(unsigned)c6[4], (unsigned)c6[5]};
            row[896 + 16 * hi + 4 * h + vg] = (unsigned char)(sexp + 127 - 4);
          }
        if (ch == 0) {                                              // the pads of the scale area and the row's zero tail
          if (h == 0) {
            *reinterpret_cast<unsigned long long*>(row + 904) = 0ull;
            *reinterpret_cast<unsigned long long*>(row + 920) = 0ull;
          } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) *reinterpret_cast<i32x4*>(row + 928 + 16 * i) = i32x4{0, 0, 0, 0};
          }
        }
        __syncthreads();                                            // (4) the rows are whole
        if (row_ok) {
          unsigned char* dst = p.y_bank + fpix0 * p.bank_row_bytes;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int r = 16 * ch + i;
            if (x0 + r < p.W)
              *reinterpret_cast<uint4*>(dst + (size_t)r * p.bank_row_bytes + 16 * lane) = *reinterpret_cast<const uint4*>(rows + r * BK_RS + 16 * lane);
          }
        }
        __syncthreads();                                            // (5) before the next pixel row's residual tiles
        if (p.bank_row_bytes > 1024) {                              // (uniform) round 5: the exact channels x themselves, second KiB of every row
#pragma unroll
          for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              f32x4 x = v[a][g];
              x *= inv;
              *reinterpret_cast<f32x4*>(row + 4 * (128 * ch + 32 * a + 8 * g + 4 * h)) = x;
            }
          __syncthreads();
          if (row_ok) {
            unsigned char* dst = p.y_bank + fpix0 * p.bank_row_bytes + 1024;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int r = 16 * ch + i;
              if (x0 + r < p.W)
                *reinterpret_cast<uint4*>(dst + (size_t)r * p.bank_row_bytes + 16 * lane) = *reinterpret_cast<const uint4*>(rows + r * BK_RS + 16 * lane);
            }
          }
          __syncthreads();
        }
      }
      return;
    }
  }
#pragma unroll
  for (int b = 0; b < RPW; ++b) {
    const int y = y0 + RPW * pr + b;
    if (y >= p.H) continue;                       // wave-uniform
    const size_t pix0 = ((size_t)nimg * p.Hp + (y + 1)) * p.Wp + (x0 + 1);   // pixel of lane n = 0 in the padded split tensor
    const size_t fpix0 = ((size_t)nimg * p.H + y) * p.W + x0;                 // ... and in the dense f32 tensors
    if (p.residual) {                             // rows of the residual tile -> LDS
      const unsigned char* src = reinterpret_cast<const unsigned char*>(p.residual + fpix0 * p.Cout + co_w);
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int row = i * RPI + mv_row;
        if (x0 + row < p.W)
          *reinterpret_cast<uint4*>(tile + row * RS + mv_col) =
              *reinterpret_cast<const uint4*>(src + (size_t)row * p.Cout * 4 + mv_col);
      }
      wave_sync();
    }
    f32x4 v[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cw = a * 32 + 8 * g + 4 * h;    // channel within the wave's CW: rows (r&3) + 8 (r>>2) + 4 h of the tile
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + co_w + cw);
        if constexpr (ARITH == 0)
          v[a][g] = {acc[a][b][4 * g + 0] + bv.x, acc[a][b][4 * g + 1] + bv.y, acc[a][b][4 * g + 2] + bv.z,
                     acc[a][b][4 * g + 3] + bv.w};
        else          // the accumulator holds s_x s_w times the convolution (powers of two: the product below is exact)
          v[a][g] = {fmaf(acc[a][b][4 * g + 0], p.acc_scale, bv.x), fmaf(acc[a][b][4 * g + 1], p.acc_scale, bv.y),
                     fmaf(acc[a][b][4 * g + 2], p.acc_scale, bv.z), fmaf(acc[a][b][4 * g + 3], p.acc_scale, bv.w)};
        if (p.residual) v[a][g] += *reinterpret_cast<const f32x4*>(tile + n * RS + cw * 4);
        if (p.relu) {
          v[a][g].x = fmaxf(v[a][g].x, 0.f); v[a][g].y = fmaxf(v[a][g].y, 0.f);
          v[a][g].z = fmaxf(v[a][g].z, 0.f); v[a][g].w = fmaxf(v[a][g].w, 0.f);
        }
      }
    if (p.y_f32) {
      wave_sync();                                // residual reads done before the region is overwritten
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(tile + n * RS + (a * 32 + 8 * g + 4 * h) * 4) = v[a][g];
      wave_sync();
      unsigned char* dst = reinterpret_cast<unsigned char*>(p.y_f32 + fpix0 * p.Cout + co_w);
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int row = i * RPI + mv_row;
        if (x0 + row < p.W)
          *reinterpret_cast<uint4*>(dst + (size_t)row * p.Cout * 4 + mv_col) =
              *reinterpret_cast<const uint4*>(tile + row * RS + mv_col);
      }
    }
    if (p.y_split) {
      wave_sync();
      if (p.out_fmt == 0) {
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 x = v[a][g];
            ushort4 hv, lv;
            split_bf16_4(x, hv, lv);
            unsigned char* o = tile + n * RS + a * 128 + (8 * g + 4 * h) * 2;    // [chunk a][hi 64 B | lo 64 B]
            *reinterpret_cast<ushort4*>(o) = hv;
            *reinterpret_cast<ushort4*>(o + 64) = lv;
          }
      } else if (p.out_fmt == 3) {                  // f16 + FP6: [h 64 B | l6, h6 main | l6, h6 tail + scale byte] (common.hpp)
        bool ovf = false;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          uint2 hw[4];
          i32x4 main6, tail6;
          split_f16f6_chunk(v[a], p.out_scale, h, hw, main6, tail6, ovf);
          unsigned char* o = tile + n * RS + a * 128;
#pragma unroll
          for (int g = 0; g < 4; ++g) *reinterpret_cast<uint2*>(o + (8 * g + 4 * h) * 2) = hw[g];
          *reinterpret_cast<i32x4*>(o + 80 - 16 * h) = main6;
          *reinterpret_cast<i32x4*>(o + 112 - 16 * h) = tail6;
        }
        if (__builtin_amdgcn_ballot_w64(ovf && x0 + n < p.W) != 0ull && lane == 0) atomicOr(p.overflow, 1);
      } else {                                      // the f16 forms the next layer reads: [h 64 B | l8 32 B | h8 32 B] or [h 64 B | l 64 B]
        bool ovf = false;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint2 hw, lw;
            uint32_t l8, h8;
            split_f16_4(v[a][g], p.out_scale, hw, l8, h8, lw, ovf);
            unsigned char* o = tile + n * RS + a * 128;
            *reinterpret_cast<uint2*>(o + (8 * g + 4 * h) * 2) = hw;
            if (p.out_fmt == 1) {
              *reinterpret_cast<uint32_t*>(o + 64 + 8 * g + 4 * h) = l8;
              *reinterpret_cast<uint32_t*>(o + 96 + 8 * g + 4 * h) = h8;
            } else {
              *reinterpret_cast<uint2*>(o + 64 + (8 * g + 4 * h) * 2) = lw;
            }
          }
        if (__builtin_amdgcn_ballot_w64(ovf && x0 + n < p.W) != 0ull && lane == 0) atomicOr(p.overflow, 1);
      }
      wave_sync();
      // in global memory the wave's CW channels of a pixel are CW/32 consecutive 128-byte chunks = RB contiguous bytes
      unsigned char* dst = reinterpret_cast<unsigned char*>(p.y_split) + (pix0 * (p.Cout / 32) + (co_w >> 5)) * 128;
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int row = i * RPI + mv_row;
        if (x0 + row < p.W)
          *reinterpret_cast<uint4*>(dst + (size_t)row * p.Cout * 4 + mv_col) =
              *reinterpret_cast<const uint4*>(tile + row * RS + mv_col);
      }
    }
    wave_sync();
  }
}

// f32 NCHW -> padded split NHWC (interior only; the border must already be zero) and/or dense NHWC f32
__global__ __launch_bounds__(256) void nchw_to_split_nhwc_kernel(const float* __restrict__ in, uint16_t* __restrict__ out,
                                                                  float* __restrict__ out_f32, int C, int H, int W, int Hp,
                                                                  int Wp) {
  // one thread = one pixel x 32-channel chunk; consecutive threads = consecutive x -> coalesced reads per channel
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y, chunk = blockIdx.z % (C / 32), nimg = blockIdx.z / (C / 32);
  if (x >= W) return;
  const float* src = in + (((size_t)nimg * C + chunk * 32) * H + y) * W + x;
  __attribute__((aligned(16))) uint16_t hv[32];
  __attribute__((aligned(16))) uint16_t lv[32];
  __attribute__((aligned(16))) float fv[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    const float v = src[(size_t)c * H * W];
    fv[c] = v;
    hv[c] = f2bf(v);
    lv[c] = f2bf(v - bf2f(hv[c]));
  }
  const size_t pix = ((size_t)nimg * Hp + (y + 1)) * Wp + (x + 1);
  if (out) {
    uint16_t* o = out + (pix * (C / 32) + chunk) * 64;
#pragma unroll
    for (int c = 0; c < 32; c += 8) {
      *reinterpret_cast<uint4*>(o + c) = *reinterpret_cast<const uint4*>(hv + c);
      *reinterpret_cast<uint4*>(o + 32 + c) = *reinterpret_cast<const uint4*>(lv + c);
    }
  }
  if (out_f32) {
    float* o = out_f32 + (((size_t)nimg * H + y) * W + x) * C + chunk * 32;
#pragma unroll
    for (int c = 0; c < 32; c += 4) *reinterpret_cast<f32x4*>(o + c) = *reinterpret_cast<const f32x4*>(fv + c);
  }
}

// dense NHWC f32 -> L2-normalised rows: [n][H*W][C] f32 (the layout of fgvc_normalize_chw_to_hwc_f32's output) and / or their
// (hi, lo) bf16 split [n][H*W][hi C | lo C] (fgvc_split_bf16's output: what fgvc_corr_volume_bf16x3 reads); one wave per pixel
template <int FMT>   // split format: 0 = (hi, lo) bf16 (fgvc_split_bf16), 1 = (h, l) f16 at scale 2^14 (fgvc_split_f16x2)
__global__ __launch_bounds__(256) void normalize_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              uint16_t* __restrict__ out_split, int C, int normalize,
                                                              long long npix) {
  const long long pixel = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (pixel >= npix) return;
  const float* src = in + (size_t)pixel * C;
  float ss = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
  const float inv = normalize ? 1.0f / fmaxf(sqrtf(ss), 1e-12f) : 1.0f;      // F.normalize: x / max(||x||, eps)
  for (int c = lane * 4; c < C; c += 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
    v *= inv;
    if (out) *reinterpret_cast<f32x4*>(out + pixel * C + c) = v;
    if (out_split) {
      ushort4 hv, lv;
      if constexpr (FMT == 0) {
        split_bf16_4(v, hv, lv);
      } else {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 h, l;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float xs = v[i] * 16384.f;
          h[i] = (_Float16)xs;
          l[i] = (_Float16)(xs - (float)h[i]);
        }
        hv = __builtin_bit_cast(ushort4, h);
        lv = __builtin_bit_cast(ushort4, l);
      }
      *reinterpret_cast<ushort4*>(out_split + pixel * 2 * C + c) = hv;
      *reinterpret_cast<ushort4*>(out_split + pixel * 2 * C + C + c) = lv;
    }
  }
}

// ... and straight into the rows of fgvc_split_f16f6p (csrc/pair_topk_v7.hpp: what fgvc_pair_topk_f16f6 reads), C = 256: one wave per
// pixel, lane = 4 consecutive channels.  A scale block of that format is the 32 channels 64 v + 16 m + 8 hi + i (m < 4, i < 8) = the 8
// lanes {16 v + 4 m + 2 hi + b}: block maxima by three xor-shuffles (1, 4, 8), a lane's four FP6 codes are 24 bits of its block's
// 192-bit string at bit 48 m + 24 b, and the lanes with piece index 2 m + b < 6 assemble one 32-bit word each from two pieces.
__device__ __forceinline__ int n6_scale_exp(float m) {          // (= p6_scale_exp of pair_topk_v7.hpp)
  if (!(m > 0.f)) return -40;
  int e;
  const float f = frexpf(m * (1.0f / 7.5f), &e);
  int s = (f > 0.5f) ? e : e - 1;
  if (m * exp2f((float)-s) > 7.5f) ++s;
  return imax(s, -40);
}
__device__ __forceinline__ unsigned n6_code(float y) {
  const float a = fabsf(y);
  const float inv_step = a < 2.f ? 8.f : (a < 4.f ? 4.f : 2.f);
  const float r = fminf(__builtin_rintf(a * inv_step) / inv_step, 7.5f);
  const float c = r < 2.f ? 8.f * r : (r < 4.f ? 8.f + 4.f * r : 16.f + 2.f * r);
  return (unsigned)c | (y < 0.f ? 32u : 0u);
}
__global__ __launch_bounds__(256) void normalize_f16f6p_kernel(const float* __restrict__ in, unsigned char* __restrict__ out, int normalize,
                                                                long long npix, int rowb) {
  const long long pixel = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (pixel >= npix) return;
  // (the norm exactly as normalize_nhwc_kernel computes it, expression for expression: the rows then equal fgvc_split_f16f6p of that
  // kernel's f32 rows bit for bit -- a sum contracted differently moves 1/|x| by an ulp and a few FP6 codes of the residual with it)
  const float* src = in + (size_t)pixel * 256;
  float ss;
  {
#pragma clang fp contract(off)      // (that kernel's sum compiles to four rounded squares and three adds: no fused multiply-add here either)
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * lane);
    const float xx = v.x * v.x, yy = v.y * v.y, zz = v.z * v.z, ww = v.w * v.w;
    ss = ((xx + yy) + zz) + ww;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
  const float inv = normalize ? 1.0f / fmaxf(sqrtf(ss), 1e-12f) : 1.0f;
  f32x4 x = *reinterpret_cast<const f32x4*>(src + 4 * lane);
  x *= inv;
  unsigned char* row = out + (size_t)pixel * rowb;
  if (rowb > 1024) *reinterpret_cast<f32x4*>(row + 1024 + 16 * lane) = x;      // fgvc_split_f16f6x rows: the exact channels in the second KiB
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  f16x4 hv;
  float hf[4], lf[4], mh = 0.f, ml = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float xs = x[i] * 256.f;
    const _Float16 h = (_Float16)xs;
    hv[i] = h;
    hf[i] = (float)h;
    lf[i] = (xs - hf[i]) * 256.f;
    mh = fmaxf(mh, fabsf(hf[i]));
    ml = fmaxf(ml, fabsf(lf[i]));
  }
  *reinterpret_cast<f16x4*>(row + 8 * lane) = hv;
#pragma unroll
  for (int m = 1; m <= 8; m <<= 1) {
    if (m == 2) continue;                                       // bit 1 of the lane is hi: another block
    mh = fmaxf(mh, __shfl_xor(mh, m));
    ml = fmaxf(ml, __shfl_xor(ml, m));
  }
  const int sh = n6_scale_exp(mh), sl = n6_scale_exp(ml);
  const float ih = exp2f((float)-sh), il = exp2f((float)-sl);
  unsigned ph = 0, pl = 0;                                      // this lane's four codes: 24 bits
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ph |= n6_code(hf[i] * ih) << (6 * i);
    pl |= n6_code(lf[i] * il) << (6 * i);
  }
  const int v = lane >> 4, hi = (lane >> 1) & 1, pi = ((lane >> 2) & 3) * 2 + (lane & 1);      // piece index inside the block
  // word w = pi (w < 6) of the block's string: pieces pa = 4 w / 3 (from bit oa = 32 w - 24 pa of it) and pa + 1
  const int w = pi < 6 ? pi : 0;
  const int pa = (4 * w) / 3, oa = 32 * w - 24 * pa;
  const int base = (lane & ~13) ;                                // lane of piece 0 of this block: 16 v + 2 hi
  const int la = base + 4 * (pa >> 1) + (pa & 1), lb = base + 4 * ((pa + 1) >> 1) + ((pa + 1) & 1);
  const unsigned ha = __shfl(ph, la), hb = __shfl(ph, lb), qa = __shfl(pl, la), qb = __shfl(pl, lb);
  const unsigned wh = (ha >> oa) | (hb << (24 - oa)), wl = (qa >> oa) | (qb << (24 - oa));
  if (pi < 4) {
    *reinterpret_cast<unsigned*>(row + 512 + 32 * v + 16 * hi + 4 * pi) = wh;
    *reinterpret_cast<unsigned*>(row + 704 + 32 * v + 16 * hi + 4 * pi) = wl;
  } else if (pi < 6) {
    *reinterpret_cast<unsigned*>(row + 640 + 32 * (v >> 1) + 16 * hi + 8 * (v & 1) + 4 * (pi - 4)) = wh;
    *reinterpret_cast<unsigned*>(row + 832 + 32 * (v >> 1) + 16 * hi + 8 * (v & 1) + 4 * (pi - 4)) = wl;
  } else if (pi == 6) {
    row[896 + 16 * hi + v] = (unsigned char)(sh + 127 - 4);
  } else {
    row[896 + 16 * hi + 4 + v] = (unsigned char)(sl + 127 - 4);
  }
  // the rest of the row: [904, 912), [920, 928) pads of the scale area, [928, 1024) zero
  if (lane < 2) *reinterpret_cast<unsigned long long*>(row + 904 + 16 * lane) = 0ull;
  else if (lane < 26) *reinterpret_cast<unsigned*>(row + 928 + 4 * (lane - 2)) = 0u;
}

static int g_conv_debug = 0;
void set_conv_debug(int v) { g_conv_debug = v; }
static int g_conv_narrow = 1;       // bit 0: 64-channel layers, bit 1: 128-channel 3x3 layers (no gain measured) -- 4-row tiles, two workgroups per CU (0: 8-row tiles)
void set_conv_narrow(int v) { g_conv_narrow = v; }
static int g_conv_cot_cap = 0;     // tuning knob: cap the output channels per workgroup (0 = widest that divides Cout)
void set_conv_cot_cap(int v) { g_conv_cot_cap = v; }

// ---- round 5: the 256 -> 256 channel 3 x 3 convolution (f16 + FP6) with ONE wave per SIMD, a fixed instruction stream per stage ------
// conv_split_kernel runs this layer with two waves per SIMD, 24 matrix instructions per wave and stage.  Round 4's s_memtime probe
// (profiles/r04_probe_conv_stage.log) had the older wave of a SIMD spend 1 600 cycles on them and the younger 2 150, one after the other:
// 768 cycles of matrix pipe each -- a wave's own stream is instruction-bound (~200 instructions beside the 24), and the second wave
// does not hide that, it queues behind it.  Round 5's conv64p_kernel (conv64.hip) showed what a wave that owns its SIMD does when its
// stream is laid out by hand: one instruction issues per ~4 cycles, a matrix instruction covers ~6 of them.
// Here: 4 waves per workgroup, the same 8 x 32 pixel x 256 channel tile, LDS layout, patch and weight ring; wave (ch, pr) owns output
// channels 128 ch ..+128 (four 32-channel tiles) x pixel rows 4 pr ..+4: 16 accumulator tiles = the 256 accumulation registers.  A stage
// (one tap of one 32-channel input chunk) = 48 matrix instructions = 1 536 cycles of pipe, in conv_split_kernel's order per accumulator
// (f16 k-step 0, 1, then the FP6 cross terms: bit-identical sums); beside them 16 weight-fragment reads (tile a + 1 during tile a),
// 16 pixel-fragment reads (the NEXT tap's, into the other buffer), 8 weight DMA pieces of the stage two ahead, and -- taps 5-7 of a
// chunk -- the next chunk's patch into registers, written to the LDS during tap 8 (whose own pixel fragments were read during tap 7).
// ~120 instructions per 48: every gap has room.  tools/gen_conv256p_sched.py deals them out and counts the vmcnt waits
// (conv256p_sched.inc).  Everything an assembly statement writes asynchronously sits in NAMED registers (conv64.hip has the story):
//   accumulators a[0:255] = [tile a][row r] x 16;  pixel fragments v[0:63] / v[64:127] (buffer x row x [f16 k0 | k1 | FP6 16 B | 16 B]);
//   weight fragments v[128:143] / v[144:159];  the patch in flight v[160:211].
// OUT_FMT / RES / F32OUT: the epilogue's form at compile time (-1 / the runtime fields when GENERIC).  With every form in one body, unrolled
// over the wave's four pixel rows, the epilogue was 13 000 instructions -- more than the instruction cache -- and took 88 000 cycles per tile
// beside the loop's 140 000.
// COT: output channels per workgroup, 256 (four 32-channel tiles per wave, 16 accumulators) or 128 (two tiles, 8 accumulators: the 128 -> 128
// layers; conv128p_loop.inc -- a stage has 24 matrix instructions there).
// CIN_TAG: no code depends on it -- the 128 -> 256 convolution of a stage's first block is launched as its own instance (1) so that a kernel
// trace tells its launches (half the loop) from the 256 -> 256 ones (round-5 review: one symbol mixed both, 123-374 us)
template <int COT, int OUT_FMT, bool RES, bool F32OUT, bool GENERIC, bool BANK = false, int CIN_TAG = 0>
__global__ __launch_bounds__(256, 1) void conv256p_kernel(ConvSplitParams p) {
  constexpr int NA = COT / 64, PPW = NA * 2;
  constexpr int TROWS = 8, PATCHB = (TROWS + 2) * CV_PW * 128, SLOTB = COT * 128, NPIECE = (TROWS + 2) * 5;
  __shared__ __attribute__((aligned(16))) unsigned char smem[PATCHB + 3 * SLOTB + COT * 4];
  unsigned char* patch = smem;
  unsigned char* wring = smem + PATCHB;
  float* bias_lds = reinterpret_cast<float*>(smem + PATCHB + 3 * SLOTB);      // the workgroup's COT bias values (written here, read in the epilogue:
                                                                              // as global loads under register pressure they were waited for one by one)
  const long long t_begin = __builtin_amdgcn_s_memtime();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = wave & 1, pr = wave >> 1;
  const int n = lane & 31, h = lane >> 5;
  const int d_row = lane >> 3, d_slot = lane & 7;
  int bid = blockIdx.x;
  const int nimg = bid / (p.n_ty * p.n_tx);
  bid -= nimg * p.n_ty * p.n_tx;
  const int ty = bid / p.n_tx, tx = bid - ty * p.n_tx;
  const int y0 = ty * TROWS, x0 = tx * 32;
  const int co_base = blockIdx.y * COT;
  const int nchunk = p.Cin / 32;
  const uint32_t pixb = (uint32_t)nchunk * 128u;

  // ---- lane parts and loop-invariant scalars
  const uint32_t w_lane_off = (uint32_t)(d_row * 128 + ((d_slot ^ (d_row >> 1)) << 4));
  const uint32_t wo[2] = {w_lane_off, w_lane_off ^ 64u};                     // piece j: first channel 8 (8 wave + j), swizzle key 4 (j & 1)
  const uint32_t ring_lds = lds_addr(wring), patch_lds = lds_addr(patch);
  const uint32_t wl = ring_lds + (uint32_t)((ch * (COT / 2) + n) * 128) + ((((uint32_t)h) ^ (uint32_t)((n >> 1) & 7)) << 4);
  uint32_t pl[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const uint32_t key = (uint32_t)((((n + dx) >> 1) & 7) ^ (4 * par));
      pl[dx][par] = patch_lds + (uint32_t)((4 * pr * CV_PW + n + dx) * 128) + ((((uint32_t)h) ^ key) << 4);
    }
  const uint32_t pf_lane_off = (uint32_t)d_row * pixb + (uint32_t)((d_slot ^ (d_row >> 1)) << 4);
  const uint32_t lane16 = (uint32_t)lane * 16u;
  int ptab;                                                 // lane k: patch piece wave + 4 k of this wave: source offset | LDS offset | swizzle key
  {
    const int k = lane & 15, i = imin(wave + 4 * k, NPIECE - 1);
    const int prow = i / 5, pc0 = (i - prow * 5) * 8;
    const int goff = (int)((uint32_t)(prow * p.Wp + pc0) * pixb);
    const int ldst = (prow * CV_PW + pc0) * 128;
    const int key = ((prow + (pc0 >> 3)) & 1) << 6;
    ptab = lane < 16 ? goff : lane < 32 ? ldst : key;
  }
  const unsigned char* xb_tile = reinterpret_cast<const unsigned char*>(p.x) + (uint32_t)((nimg * p.Hp + y0) * p.Wp + x0) * pixb;
  const unsigned char* wb_tile = reinterpret_cast<const unsigned char*>(p.w) + (uint32_t)(co_base * 128);
  const uint32_t piece_off = (uint32_t)(wave * PPW * 1024);                 // this wave's PPW pieces = output channels 8 PPW wave ..+8 PPW of a slab
  auto weight_base = [&](int chunk, int tap) { return wb_tile + (uint32_t)((tap * nchunk + chunk) * p.Cout * 128); };
  auto weight_piece = [&](const unsigned char* base, uint32_t dst, int j) {
    const unsigned char* src = base + piece_off + (uint32_t)(j * 1024);
    const uint32_t d = dst + piece_off + (uint32_t)(j * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(wo[j & 1]), "s"(src), "s"(d) : "memory");
  };

  if (tid < COT / 4) reinterpret_cast<f32x4*>(bias_lds)[tid] = reinterpret_cast<const f32x4*>(p.bias + co_base)[tid];
  // ---- prologue: the first chunk's patch (LDS-DMA), the weights of stages 0 and 1
  for (int i = wave; i < NPIECE; i += 4) {
    const int prow = i / 5, pc0 = (i - prow * 5) * 8;
    const int P = prow * CV_PW + pc0 + d_row;
    const int sl = d_slot ^ ((P >> 1) & 7);
    const size_t gpix = ((size_t)nimg * p.Hp + (y0 + prow)) * p.Wp + (x0 + pc0 + d_row);
    conv_lds_dma_16(reinterpret_cast<const unsigned char*>(p.x) + gpix * pixb + sl * 16, lds_addr(patch + (prow * CV_PW + pc0) * 128));
  }
#pragma unroll
  for (int j = 0; j < PPW; ++j) weight_piece(weight_base(0, 0), ring_lds, j);
#pragma unroll
  for (int j = 0; j < PPW; ++j) weight_piece(weight_base(0, 1), ring_lds + SLOTB, j);

  __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0), as the builtin (conv64.hip): prologue DMAs landed (this wave's)

  const long long t_loop0 = __builtin_amdgcn_s_memtime();
  // ---- the main loop: one assembly statement (tools/gen_conv256p_sched.py has the register map and the reasons)
  {
    const size_t wb64 = (size_t)wb_tile, xb64 = (size_t)xb_tile;
    const int s_tap = __builtin_amdgcn_readfirstlane(nchunk * p.Cout * 128), s_chunk = __builtin_amdgcn_readfirstlane(p.Cout * 128);
    const int s_nchunk = __builtin_amdgcn_readfirstlane(nchunk), s_ring = __builtin_amdgcn_readfirstlane((int)ring_lds);
    const int s_patch = __builtin_amdgcn_readfirstlane((int)patch_lds), s_piece = __builtin_amdgcn_readfirstlane((int)piece_off);
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    const i32x2 s_wb = {__builtin_amdgcn_readfirstlane((int)(uint32_t)wb64), __builtin_amdgcn_readfirstlane((int)(uint32_t)(wb64 >> 32))};
    const i32x2 s_xb = {__builtin_amdgcn_readfirstlane((int)(uint32_t)xb64), __builtin_amdgcn_readfirstlane((int)(uint32_t)(xb64 >> 32))};
    typedef int i32x2_t __attribute__((ext_vector_type(2)));
    i32x2_t loop_cycles;
    // the second input (p.x2 / p.w2: a block's 1 x 1 projection shortcut folded into this convolution): its chunks follow the main loop's
    const int nchunk2 = p.x2 ? p.Cin2 / 32 : 0;
    const uint32_t pixb2 = (uint32_t)nchunk2 * 128u;
    const uint32_t pf_lane_off2 = (uint32_t)d_row * pixb2 + (uint32_t)((d_slot ^ (d_row >> 1)) << 4);
    int ptab2;
    {
      const int k2 = lane & 15, i2 = imin(wave + 4 * k2, NPIECE - 1);
      const int prow = i2 / 5, pc0 = (i2 - prow * 5) * 8;
      const int goff = (int)((uint32_t)(prow * p.Wp + pc0) * pixb2);
      ptab2 = lane < 16 ? goff : lane < 32 ? (prow * CV_PW + pc0) * 128 : (((prow + (pc0 >> 3)) & 1) << 6);
    }
    const size_t w2b64 = (size_t)p.w2 + (uint32_t)(co_base * 128), x2b64 = (size_t)p.x2 + (size_t)((uint32_t)((nimg * p.Hp + y0) * p.Wp + x0) * pixb2);
    const i32x2 s_w2b = {__builtin_amdgcn_readfirstlane((int)(uint32_t)w2b64), __builtin_amdgcn_readfirstlane((int)(uint32_t)(w2b64 >> 32))};
    const i32x2 s_x2b = {__builtin_amdgcn_readfirstlane((int)(uint32_t)x2b64), __builtin_amdgcn_readfirstlane((int)(uint32_t)(x2b64 >> 32))};
    const int s_last2 = __builtin_amdgcn_readfirstlane(nchunk2 ? nchunk - 1 : -1), s_nchunk2 = __builtin_amdgcn_readfirstlane(nchunk2);
#define C256P_INPUTS                                                                                                              \
  "{v232}"(wl), "{v233}"(pl[0][0]), "{v234}"(pl[0][1]), "{v235}"(pl[1][0]), "{v236}"(pl[1][1]), "{v237}"(pl[2][0]), "{v238}"(pl[2][1]),     \
      "{v239}"(wo[0]), "{v240}"(wo[1]), "{v241}"(pf_lane_off), "{v242}"(lane16), "{v243}"(ptab), "{v244}"(pf_lane_off2), "{v245}"(ptab2),  \
      "{s[20:21]}"(s_wb), "{s22}"(s_tap), "{s23}"(s_chunk), "{s[24:25]}"(s_xb), "{s26}"(s_nchunk), "{s27}"(s_ring), "{s28}"(s_patch),      \
      "{s29}"(s_piece), "{s46}"(s_last2), "{s[48:49]}"(s_w2b), "{s[50:51]}"(s_x2b), "{s52}"(s_nchunk2)
    if constexpr (COT == 256) {
#include "conv256p_loop.inc"
    } else {
#include "conv128p_loop.inc"
    }
#undef C256P_INPUTS
    if ((p.debug & 8) && blockIdx.x == 300 && blockIdx.y == 0 && lane == 0 && p.y_split) {    // option conv_debug & 8: s_memtime of workgroup 300's loop (as conv_split_kernel)
      long long* o = reinterpret_cast<long long*>(p.y_split) + wave * 8;
      o[4] = (long long)(uint32_t)loop_cycles.x | ((long long)loop_cycles.y << 32);
      o[5] = nchunk * 9 + nchunk2;
      o[0] = t_loop0 - t_begin;
    }
  }
  f32x16 acc[NA][4];
  if constexpr (NA == 4)
    asm volatile(""
                 : "={a[0:15]}"(acc[0][0]), "={a[16:31]}"(acc[0][1]), "={a[32:47]}"(acc[0][2]), "={a[48:63]}"(acc[0][3]), "={a[64:79]}"(acc[1][0]),
                   "={a[80:95]}"(acc[1][1]), "={a[96:111]}"(acc[1][2]), "={a[112:127]}"(acc[1][3]), "={a[128:143]}"(acc[NA - 2][0]),
                   "={a[144:159]}"(acc[NA - 2][1]), "={a[160:175]}"(acc[NA - 2][2]), "={a[176:191]}"(acc[NA - 2][3]), "={a[192:207]}"(acc[NA - 1][0]),
                   "={a[208:223]}"(acc[NA - 1][1]), "={a[224:239]}"(acc[NA - 1][2]), "={a[240:255]}"(acc[NA - 1][3]));
  else
    asm volatile(""
                 : "={a[0:15]}"(acc[0][0]), "={a[16:31]}"(acc[0][1]), "={a[32:47]}"(acc[0][2]), "={a[48:63]}"(acc[0][3]), "={a[64:79]}"(acc[1][0]),
                   "={a[80:95]}"(acc[1][1]), "={a[96:111]}"(acc[1][2]), "={a[112:127]}"(acc[1][3]));

  // ---- epilogue (conv_split_kernel's, for 4 tiles x 4 rows per wave): + bias [+ residual] [ReLU], pixel rows through a wave-private LDS tile
  // The loop's statement clobbers every vector register but ten: whatever per-lane value the epilogue needs and the compiler had computed
  // BEFORE the loop (lane, pixel, lane half and what it derives from them: staging addresses, store columns) went to scratch around it --
  // 16-18 registers of the two hottest instances (round-5 review).  The lane id is taken again here, by an instruction the compiler cannot
  // move above the loop, and the epilogue derives its values from that.
  int lane_e;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
  {
  int lane = lane_e, n = lane & 31, h = lane >> 5;      // (not const: re-taken per pixel row below -- `row_fence`)
  // (the padded width too: the prologue reads p.Wp under a lane predicate, the compiler kept that VECTOR copy, sign-extended, for the
  // epilogue's pixel addresses -- two registers spilled around the loop in the plain instances once the conversion was rewritten in
  // round 6; a scalar copy taken as new here is what the rows multiply by)
  int Wp_e = __builtin_amdgcn_readfirstlane(p.Wp);
  asm volatile("" : "+s"(Wp_e));
  constexpr int RPW = 4, CW = COT / 2, RB = CW * 4, RS = RB + 16, LPR = RB / 16, RPI = 64 / LPR;
  static_assert(4 * 32 * RS <= PATCHB + 3 * SLOTB, "epilogue staging");
  __syncthreads();
  long long ts0 = __builtin_amdgcn_s_memtime(), ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;
  unsigned char* tile = smem + wave * (32 * RS);
  const int co_w = co_base + ch * CW;
  int mv_row = lane / LPR, mv_col = (lane % LPR) * 16;
  // per pixel row: the lane's values are taken as new (an empty statement that 'writes' them), so that the compiler computes a row's
  // addresses in that row instead of all four rows' up front -- 20 registers of the plain instances went to scratch for those
  auto row_fence = [&]() { asm volatile("" : "+v"(n), "+v"(h), "+v"(mv_row), "+v"(mv_col)); };
  auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  // ---- the f32 residual rows come by LDS-DMA, one pixel row of the wave ahead of their use.  (Read through registers where they are needed,
  // every load under its own predicate, the compiler waited for each by itself -- 16 memory latencies per pixel row, 18 000 of a row's
  // 26 000 cycles; batched into registers they do not fit beside 256 accumulators: scratch memory.)  Region: behind the four staging tiles,
  // 32 pixels x RB bytes per wave, 16-byte slot s of pixel p at slot s ^ (p mod slots): the reads in the accumulator layout (one pixel
  // per lane, the same logical slot) then fall on different banks.
  constexpr int RSLOTS = RB / 16, RPIECES = 32 * RB / 1024, RPPP = 1024 / RB;       // slots per pixel row, 1-KiB pieces per wave row, pixels per piece
  static_assert(4 * 32 * RS + 4 * 32 * RB <= PATCHB + 3 * SLOTB, "residual tiles");
  unsigned char* rtile = smem + 4 * (32 * RS) + wave * (32 * RB);
  const bool has_res = GENERIC || BANK ? p.residual != nullptr : RES;
  auto residual_dma = [&](int b_) {
    const int y_ = y0 + RPW * pr + b_;
    if (!has_res || b_ >= RPW || y_ >= p.H) return;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(p.residual + (((size_t)nimg * p.H + y_) * p.W + x0) * p.Cout + co_w);
    const int pmax = p.W - 1 - x0;                          // (pixels beyond the row's end read its last pixel: never stored)
#pragma unroll
    for (int j = 0; j < RPIECES; ++j) {
      const int pp = j * RPPP + (lane * 16) / RB, ps = ((lane * 16) % RB) / 16;
      const uint32_t voff = (uint32_t)(imin(pp, pmax) * p.Cout * 4 + ((ps ^ (pp & (RSLOTS - 1))) << 4));
      conv_lds_dma_16s(voff, src, lds_addr(rtile + j * 1024));
    }
  };
  // the row's DMA pieces are older than the previous row's stores (issued behind them): a counted wait lets those stores drain on their
  // own -- when the tile is interior, i.e. every store instruction was issued (a predicated one whose pixels are all outside is skipped)
  auto residual_wait = [&](int stores_behind) {
    if (x0 + 32 > p.W || stores_behind == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (stores_behind == 32 / RPI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(32 / RPI) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (32 / RPI)) : "memory");
  };
  auto residual_at = [&](int cw) {                          // the lane's pixel n, channels cw ..+4 of the wave's CW
    return *reinterpret_cast<const f32x4*>(rtile + n * RB + ((((cw * 4) >> 4) ^ (n & (RSLOTS - 1))) << 4));
  };
  residual_dma(0);
  // the staged pixel rows to global memory; a tile that lies wholly inside the image row (all but the last tile column) stores without
  // per-pixel predicates
  const bool tile_full = x0 + 32 <= p.W;
  auto store_rows = [&](unsigned char* dst) {
    unsigned char* d0 = dst + (size_t)mv_row * p.Cout * 4 + mv_col;
    const unsigned char* t0 = tile + mv_row * RS + mv_col;
    if (tile_full) {
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) *reinterpret_cast<uint4*>(d0 + (size_t)(i * RPI) * p.Cout * 4) = *reinterpret_cast<const uint4*>(t0 + i * RPI * RS);
    } else {
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i)
        if (x0 + i * RPI + mv_row < p.W) *reinterpret_cast<uint4*>(d0 + (size_t)(i * RPI) * p.Cout * 4) = *reinterpret_cast<const uint4*>(t0 + i * RPI * RS);
    }
  };
  if constexpr (BANK) {
    {
      // ---- the feature bank straight from the accumulators (conv_split_kernel's BANK epilogue, here for 4 rows per wave and two row groups): what normalize_f16f6p_kernel makes of this convolution's dense
      // f32 output, bit for bit, without that output ever reaching memory.  A workgroup holds all 256 channels of its pixels: wave
      // (pr, ch) channels 128 ch ..+128 of pixel rows 2 pr, 2 pr + 1.  Per pixel row: (1) the sum of squares in that kernel's own
      // association -- lane L of its wave-per-pixel layout holds channels 4 L ..+4 = here group (ch, a, g, h) with L = 32 ch + 8 a +
      // 2 g + h, and its xor-shuffle tree adds partners 32, 16, 8, 4, 2, 1 apart: the other wave's partial through LDS, then a ^ 2,
      // a ^ 1, g ^ 2, g ^ 1 in registers, then the other lane half; (2) x = v / |v|, h = f16(256 x), residual 256 (256 x - h);
      // (3) a scale block of the row format = channels 64 v + 16 m + 8 hi + i = groups a = 2 (v & 1) + (m >> 1), g = hi + 2 (m & 1) of
      // BOTH lane halves: the halves trade registers (v_permlane32_swap) so that lane (n, 0) holds the block's 32 h values and
      // lane (n, 1) its 32 residuals, one v_cvt_scalef32_2xpk16_fp6_f32 each (position 2 e <- S0[e], 2 e + 1 <- S1[e], round to nearest
      // even from f32: tools/micro/probe_cvt_fp6_f32.hip); (4) the pair of waves assembles the 1-KiB rows in LDS and stores them whole.
      constexpr int BK_RS = 1024 + 16;
      static_assert(COT == 256 && 2 * 32 * BK_RS <= PATCHB + 3 * SLOTB, "bank staging");
      unsigned char* rows = smem + pr * (32 * BK_RS);
      float* part = reinterpret_cast<float*>(rows);                 // [ch][a * 4 + g][lane]: aliases the rows, used before them
      long long bts[6] = {}, bt_begin = 0;
#pragma unroll
      for (int b = 0; b < RPW; ++b) {
        row_fence();
        const int y = y0 + RPW * pr + b;
        const bool row_ok = y < p.H;                                // wave-uniform, the same for both waves of a pair; barriers are taken by all
        if (b == 1) bt_begin = __builtin_amdgcn_s_memtime();
        if (b == 2) bts[4] = __builtin_amdgcn_s_memtime();
        const size_t fpix0 = ((size_t)nimg * p.H + imin(y, p.H - 1)) * p.W + x0;
        if (has_res && row_ok) residual_wait(b == 0 || 32 / RPI != 16 ? 0 : (p.bank_row_bytes > 1024 ? 32 : 16));
        f32x4 v[NA][4];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int cw = a * 32 + 8 * g + 4 * h;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + ch * CW + cw);
            v[a][g] = {fmaf(acc[a][b][4 * g + 0], p.acc_scale, bv.x), fmaf(acc[a][b][4 * g + 1], p.acc_scale, bv.y),
                         fmaf(acc[a][b][4 * g + 2], p.acc_scale, bv.z), fmaf(acc[a][b][4 * g + 3], p.acc_scale, bv.w)};
            if (has_res && row_ok) v[a][g] += residual_at(cw);
            if (p.relu) {
              v[a][g].x = fmaxf(v[a][g].x, 0.f); v[a][g].y = fmaxf(v[a][g].y, 0.f);
              v[a][g].z = fmaxf(v[a][g].z, 0.f); v[a][g].w = fmaxf(v[a][g].w, 0.f);
            }
          }
        if (has_res) { wave_sync(); residual_dma(b + 1); }
        if (b == 1) bts[0] = __builtin_amdgcn_s_memtime();
        __syncthreads();                                            // (1) the private residual tiles are read: the region becomes the pairs' buffers
        float pp[NA][4];
        {
#pragma clang fp contract(off)                                       // four rounded squares and three adds, as normalize_nhwc_kernel compiles
#pragma unroll
          for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float xx = v[a][g].x * v[a][g].x, yy = v[a][g].y * v[a][g].y, zz = v[a][g].z * v[a][g].z, ww = v[a][g].w * v[a][g].w;
              pp[a][g] = ((xx + yy) + zz) + ww;
              part[(ch * 16 + a * 4 + g) * 64 + lane] = pp[a][g];
            }
        }
        __syncthreads();                                            // (2)
        float ss;
        {
          float t[NA][4];
#pragma unroll
          for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) t[a][g] = pp[a][g] + part[((ch ^ 1) * 16 + a * 4 + g) * 64 + lane];       // partner 32 lanes away
          float u[2][4], w4[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) { u[0][g] = t[0][g] + t[2][g]; u[1][g] = t[1][g] + t[3][g]; }                // 16
#pragma unroll
          for (int g = 0; g < 4; ++g) w4[g] = u[0][g] + u[1][g];                                                 // 8
          const float z0 = w4[0] + w4[2], z1 = w4[1] + w4[3];                                                     // 4
          const float s1 = z0 + z1;                                                                               // 2
          ss = s1 + __shfl_xor(s1, 32);                                                                           // 1: the other lane half
        }
        __syncthreads();                                            // (3) the partials are read: rows may be written
        if (b == 1) bts[1] = __builtin_amdgcn_s_memtime();
        const float inv = p.bank_normalize ? 1.0f / fmaxf(sqrtf(ss), 1e-12f) : 1.0f;
        unsigned char* row = rows + n * BK_RS;
#pragma unroll
        for (int vl = 0; vl < 2; ++vl)
#pragma unroll
          for (int hi = 0; hi < 2; ++hi) {
            unsigned A_[16], B_[16];
            float mh = 0.f, ml = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const int a = 2 * vl + (m >> 1), g = hi + 2 * (m & 1);          // (every (a, g) group belongs to exactly one block)
              f32x4 x = v[a][g];
              x *= inv;
              typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
              f16x4 hv;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const float xs = x[k] * 256.f;
                const _Float16 hh_ = (_Float16)xs;
                hv[k] = hh_;
                const float fh = (float)hh_, fl = (xs - fh) * 256.f;
                mh = fmaxf(mh, fabsf(fh)); ml = fmaxf(ml, fabsf(fl));
                A_[4 * m + k] = __builtin_bit_cast(unsigned, fh);
                B_[4 * m + k] = __builtin_bit_cast(unsigned, fl);
              }
              *reinterpret_cast<f16x4*>(row + 2 * (128 * ch + 32 * a + 8 * g + 4 * h)) = hv;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {        // upper lanes of A_ <-> lower lanes of B_: A_ = the values of lane half 0, B_ of half 1
              const auto t2 = __builtin_amdgcn_permlane32_swap(A_[j], B_[j], false, false);
              A_[j] = t2[0]; B_[j] = t2[1];
            }
            const auto tm = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mh), __builtin_bit_cast(unsigned, ml), false, false);
            const unsigned mb = tm[0] > tm[1] ? tm[0] : tm[1];     // (as unsigned integers: see split_f16f6_chunk)
            int sexp = (int)(mb >> 23) - 129 + ((mb & 0x7fffffu) > 0x700000u ? 1 : 0);     // m / 2^s in (3.75, 7.5]: n6_scale_exp
            sexp = (mb == 0u || sexp < -40) ? -40 : sexp;
            fgvc_i32x16 S0, S1;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              S0[4 * m + 0] = (int)A_[4 * m + 0]; S0[4 * m + 1] = (int)A_[4 * m + 2]; S0[4 * m + 2] = (int)B_[4 * m + 0]; S0[4 * m + 3] = (int)B_[4 * m + 2];
              S1[4 * m + 0] = (int)A_[4 * m + 1]; S1[4 * m + 1] = (int)A_[4 * m + 3]; S1[4 * m + 2] = (int)B_[4 * m + 1]; S1[4 * m + 3] = (int)B_[4 * m + 3];
            }
            fgvc_i32x6 c6;
            const unsigned sc_bits = (unsigned)(sexp + 127) << 23;
            asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(c6) : "v"(S0), "v"(S1), "v"(sc_bits));
            const int vg = 2 * ch + vl;
            *reinterpret_cast<i32x4*>(row + (h ? 704 : 512) + 32 * vg + 16 * hi) = i32x4{c6[0], c6[1], c6[2], c6[3]};
            *reinterpret_cast<uint2*>(row + (h ? 832 : 640) + 32 * (vg >> 1) + 16 * hi + 8 * (vg & 1)) = uint2{(unsigned)c6[4], (unsigned)c6[5]};
            row[896 + 16 * hi + 4 * h + vg] = (unsigned char)(sexp + 127 - 4);
          }
        if (ch == 0) {                                              // the pads of the scale area and the row's zero tail
          if (h == 0) {
            *reinterpret_cast<unsigned long long*>(row + 904) = 0ull;
            *reinterpret_cast<unsigned long long*>(row + 920) = 0ull;
          } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) *reinterpret_cast<i32x4*>(row + 928 + 16 * i) = i32x4{0, 0, 0, 0};
          }
        }
        if (b == 1) bts[2] = __builtin_amdgcn_s_memtime();
        __syncthreads();                                            // (4) the rows are whole
        if (row_ok) {
          unsigned char* dst = p.y_bank + fpix0 * p.bank_row_bytes;
          if (x0 + 32 <= p.W) {                                     // (a tile inside the image row: no per-pixel predicates)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int r = 16 * ch + i;
              *reinterpret_cast<uint4*>(dst + (size_t)r * p.bank_row_bytes + 16 * lane) = *reinterpret_cast<const uint4*>(rows + r * BK_RS + 16 * lane);
            }
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int r = 16 * ch + i;
              if (x0 + r < p.W)
                *reinterpret_cast<uint4*>(dst + (size_t)r * p.bank_row_bytes + 16 * lane) = *reinterpret_cast<const uint4*>(rows + r * BK_RS + 16 * lane);
            }
          }
        }
        __syncthreads();                                            // (5) before the next pixel row's residual tiles
        if (b == 1) bts[3] = __builtin_amdgcn_s_memtime();
        if (p.bank_row_bytes > 1024) {                              // (uniform) round 5: the exact channels x themselves, second KiB of every row
#pragma unroll
          for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              f32x4 x = v[a][g];
              x *= inv;
              *reinterpret_cast<f32x4*>(row + 4 * (128 * ch + 32 * a + 8 * g + 4 * h)) = x;
            }
          __syncthreads();
          if (row_ok) {
            unsigned char* dst = p.y_bank + fpix0 * p.bank_row_bytes + 1024;
            if (x0 + 32 <= p.W) {
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                const int r = 16 * ch + i;
                *reinterpret_cast<uint4*>(dst + (size_t)r * p.bank_row_bytes + 16 * lane) = *reinterpret_cast<const uint4*>(rows + r * BK_RS + 16 * lane);
              }
            } else {
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                const int r = 16 * ch + i;
                if (x0 + r < p.W)
                  *reinterpret_cast<uint4*>(dst + (size_t)r * p.bank_row_bytes + 16 * lane) = *reinterpret_cast<const uint4*>(rows + r * BK_RS + 16 * lane);
              }
            }
          }
          __syncthreads();
        }
      }
      if ((p.debug & 8) && blockIdx.x == 300 && blockIdx.y == 0 && lane == 0) {
        long long* o_ = reinterpret_cast<long long*>(p.y_bank) + wave * 8;
        o_[0] = bts[0] - bt_begin; o_[1] = bts[1] - bts[0]; o_[2] = bts[2] - bts[1]; o_[3] = bts[3] - bts[2]; o_[4] = bts[4] - bts[3]; o_[5] = __builtin_amdgcn_s_memtime() - t_begin;
      }
      return;
    }
  }
  // (Round 5 kept the lane's 16 bias vectors in 64 registers across the wave's four rows where no residual needs the room.  With them the
  // plain instances spill 36-45 registers into scratch inside the rows; from the LDS copy (16 ds_read_b128 per row) they spill none, at
  // +0.3 % of the launch -- tools/ab_bench.sh, same box.  Off.)
  constexpr bool HOIST_BIAS = false;
  // round 6: where the split form is the only reader of the row (no f32 copy), the ReLU is the lower bound of the conversion's clamp
  constexpr bool FUSE_RELU = !GENERIC && !F32OUT && OUT_FMT == 3;
  const float clamp_lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(FUSE_RELU && p.relu ? 0 : (int)0xC77FE000u));    // 0 or -65504
  f32x4 bvv[HOIST_BIAS ? NA : 1][4];
  if constexpr (HOIST_BIAS) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int g = 0; g < 4; ++g) bvv[a][g] = *reinterpret_cast<const f32x4*>(bias_lds + ch * CW + a * 32 + 8 * g + 4 * h);
  }
#pragma unroll
  for (int b = 0; b < RPW; ++b) {
    row_fence();
    const int y = y0 + RPW * pr + b;
    if (y >= p.H) continue;                       // wave-uniform
    const size_t pix0 = ((size_t)nimg * p.Hp + (y + 1)) * Wp_e + (x0 + 1);
    const size_t fpix0 = ((size_t)nimg * p.H + y) * p.W + x0;
    if (has_res) residual_wait(b == 0 ? 0 : ((GENERIC ? p.y_f32 != nullptr : F32OUT) ? 32 / RPI : 0) + (p.y_split ? 32 / RPI : 0));
    f32x4 v[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cw = a * 32 + 8 * g + 4 * h;
        const f32x4 bv = HOIST_BIAS ? bvv[HOIST_BIAS ? a : 0][g] : *reinterpret_cast<const f32x4*>(bias_lds + ch * CW + cw);
        v[a][g] = {fmaf(acc[a][b][4 * g + 0], p.acc_scale, bv.x), fmaf(acc[a][b][4 * g + 1], p.acc_scale, bv.y),
                   fmaf(acc[a][b][4 * g + 2], p.acc_scale, bv.z), fmaf(acc[a][b][4 * g + 3], p.acc_scale, bv.w)};
        if (has_res) v[a][g] += residual_at(cw);
        if (p.relu && !FUSE_RELU) {
          v[a][g].x = fmaxf(v[a][g].x, 0.f); v[a][g].y = fmaxf(v[a][g].y, 0.f);
          v[a][g].z = fmaxf(v[a][g].z, 0.f); v[a][g].w = fmaxf(v[a][g].w, 0.f);
        }
      }
    if (has_res) { wave_sync(); residual_dma(b + 1); }                  // this row's residual is in registers: the next row's may land
    if (b == 1) { asm volatile("s_nop 0" ::"v"(v[0][0]), "v"(v[3][3])); ts1 = __builtin_amdgcn_s_memtime(); }
    if (GENERIC ? p.y_f32 != nullptr : F32OUT) {
      wave_sync();
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(tile + n * RS + (a * 32 + 8 * g + 4 * h) * 4) = v[a][g];
      wave_sync();
      store_rows(reinterpret_cast<unsigned char*>(p.y_f32 + fpix0 * p.Cout + co_w));
    }
    if (p.y_split) {
      wave_sync();
      const int out_fmt = GENERIC ? p.out_fmt : OUT_FMT;
      if (out_fmt == 0) {
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            ushort4 hv, lv;
            split_bf16_4(v[a][g], hv, lv);
            unsigned char* o = tile + n * RS + a * 128 + (8 * g + 4 * h) * 2;
            *reinterpret_cast<ushort4*>(o) = hv;
            *reinterpret_cast<ushort4*>(o + 64) = lv;
          }
      } else if (out_fmt == 3) {
        bool ovf = false;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          uint2 hw[4];
          i32x4 main6, tail6;
          split_f16f6_chunk(v[a], p.out_scale, h, hw, main6, tail6, ovf, clamp_lo);
          unsigned char* o = tile + n * RS + a * 128;
#pragma unroll
          for (int g = 0; g < 4; ++g) *reinterpret_cast<uint2*>(o + (8 * g + 4 * h) * 2) = hw[g];
          *reinterpret_cast<i32x4*>(o + 80 - 16 * h) = main6;
          *reinterpret_cast<i32x4*>(o + 112 - 16 * h) = tail6;
        }
        if (__builtin_amdgcn_ballot_w64(ovf && x0 + n < p.W) != 0ull && lane == 0) atomicOr(p.overflow, 1);
      } else {
        bool ovf = false;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint2 hw, lw;
            uint32_t l8, h8;
            split_f16_4(v[a][g], p.out_scale, hw, l8, h8, lw, ovf);
            unsigned char* o = tile + n * RS + a * 128;
            *reinterpret_cast<uint2*>(o + (8 * g + 4 * h) * 2) = hw;
            if (out_fmt == 1) {
              *reinterpret_cast<uint32_t*>(o + 64 + 8 * g + 4 * h) = l8;
              *reinterpret_cast<uint32_t*>(o + 96 + 8 * g + 4 * h) = h8;
            } else {
              *reinterpret_cast<uint2*>(o + 64 + (8 * g + 4 * h) * 2) = lw;
            }
          }
        if (__builtin_amdgcn_ballot_w64(ovf && x0 + n < p.W) != 0ull && lane == 0) atomicOr(p.overflow, 1);
      }
      wave_sync();
      if (b == 1) ts2 = __builtin_amdgcn_s_memtime();
      store_rows(reinterpret_cast<unsigned char*>(p.y_split) + (pix0 * (p.Cout / 32) + (co_w >> 5)) * 128);
    }
    wave_sync();
    if (b == 0) ts4 = __builtin_amdgcn_s_memtime();
    if (b == 1) ts3 = __builtin_amdgcn_s_memtime();
  }
  if ((p.debug & 8) && blockIdx.x == 300 && blockIdx.y == 0 && lane == 0 && p.y_split) {
    long long* o_ = reinterpret_cast<long long*>(p.y_split) + wave * 8;
    o_[2] = ts4 - ts0; o_[3] = ts1 - ts4; o_[6] = ts2 - ts1; o_[7] = ts3 - ts2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    reinterpret_cast<long long*>(p.y_split)[wave * 8 + 1] = __builtin_amdgcn_s_memtime() - t_begin;
  }
  }      // (the epilogue's scope: its own lane, n, h)
}

template <int ARITH>
static void conv_split_dispatch(const ConvSplitParams& p, dim3 grid, int KS, int cot_eff, bool narrow, hipStream_t s) {
  if (KS == 3) {
    if (cot_eff == 256 && p.y_bank) conv_split_kernel<3, 256, 1, 3, 4, false, ARITH, 2, true><<<grid, 512, 0, s>>>(p);      // the bank epilogue
    else if (cot_eff == 256 && p.x2) conv_split_kernel<3, 256, 1, 3, 4, false, ARITH, 2, false, true><<<grid, 512, 0, s>>>(p);   // + a 1 x 1 convolution of a second input
    else if (cot_eff == 256) conv_split_kernel<3, 256, 1, 3, 4, false, ARITH><<<grid, 512, 0, s>>>(p);
    else if (cot_eff == 128 && narrow) conv_split_kernel<3, 128, 1, 3, 2, false, ARITH><<<grid, 256, 0, s>>>(p);
    else if (cot_eff == 128) conv_split_kernel<3, 128, 3, 2, 4, false, ARITH><<<grid, 512, 0, s>>>(p);
    else if (narrow) conv_split_kernel<3, 64, 3, 2, 2, false, ARITH><<<grid, 256, 0, s>>>(p);
    else conv_split_kernel<3, 64, 3, 3, 4, false, ARITH><<<grid, 512, 0, s>>>(p);
  } else {
    if (cot_eff == 256) conv_split_kernel<1, 256, 1, 3, 4, false, ARITH><<<grid, 512, 0, s>>>(p);
    else if (cot_eff == 128) conv_split_kernel<1, 128, 1, 3, 4, false, ARITH><<<grid, 512, 0, s>>>(p);
    else if (narrow) conv_split_kernel<1, 64, 1, 3, 2, false, ARITH><<<grid, 256, 0, s>>>(p);
    else conv_split_kernel<1, 64, 1, 3, 4, false, ARITH><<<grid, 512, 0, s>>>(p);
  }
}

// in_fmt / out_fmt: FGVC_ACT_* (0 = (hi, lo) bf16, 1 = f16f8, 2 = (h, l) f16, 3 = f16f6); in_scale_log2 = log2(s_x s_w) of the operands (0 for bf16),
// out_scale_log2 = log2(s_out) of the split output (ignored for bf16); overflow: device word, required when out_fmt != 0
int conv_split_launch(const uint16_t* x, const uint16_t* w, const float* bias, const float* residual, uint16_t* y_split,
                      float* y_f32, int N, int H, int W, int Hp, int Wp, int Cin, int Cout, int KS, int relu, int in_fmt,
                      int in_scale_log2, int out_fmt, int out_scale_log2, int* overflow, hipStream_t s, unsigned char* y_bank,
                      int bank_normalize, const uint16_t* x2, const uint16_t* w2, int Cin2, int bank_row_bytes) {
  ConvSplitParams p;
  p.x = x; p.w = w; p.bias = bias; p.residual = residual; p.y_split = y_split; p.y_f32 = y_f32;
  p.N = N; p.H = H; p.W = W; p.Hp = Hp; p.Wp = Wp; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
  p.acc_scale = ldexpf(1.0f, -in_scale_log2); p.out_scale = ldexpf(1.0f, out_scale_log2); p.out_fmt = out_fmt; p.overflow = overflow;
  p.y_bank = y_bank; p.bank_normalize = bank_normalize; p.bank_row_bytes = bank_row_bytes;
  p.x2 = x2; p.w2 = w2; p.Cin2 = Cin2;
  const int cot = (Cout % 256 == 0) ? 256 : (Cout % 128 == 0) ? 128 : 64;
  const int cot_eff = (g_conv_cot_cap && cot > g_conv_cot_cap && !y_bank && !x2) ? g_conv_cot_cap : cot;     // (the bank epilogue needs a pixel's 256 channels in one workgroup)
  const bool narrow = (cot_eff == 64 && (g_conv_narrow & 1)) || (cot_eff == 128 && KS == 3 && (g_conv_narrow & 2));   // 4-row tiles, two workgroups per CU
  p.n_ty = cdiv(H, narrow ? 4 : 8); p.n_tx = cdiv(W, 32);
  p.debug = g_conv_debug;
  dim3 grid(p.n_ty * p.n_tx * N, Cout / cot_eff);
  // conv256p_kernel: the plain 256-channel-tile 3 x 3 form of the f16 + FP6 arithmetic (option conv_debug & 1024: conv_split_kernel)
  const bool fits32 = (unsigned long long)N * Hp * Wp * (Cin / 32) * 128ull < (1ull << 32) && (unsigned long long)9 * Cin * Cout * 4ull < (1ull << 32) &&
                      (!x2 || ((unsigned long long)N * Hp * Wp * (Cin2 / 32) * 128ull < (1ull << 32) && (unsigned long long)Cin2 * Cout * 4ull < (1ull << 32)));
  if (in_fmt == 3 && KS == 3 && (cot_eff == 256 || cot_eff == 128) && !narrow && (!y_bank || (cot_eff == 256 && !x2)) && (!x2 || (cot_eff == 256 && Cin2 >= 32)) &&
      Cin >= 32 && fits32 && !(g_conv_debug & 1024)) {
    if (y_bank) conv256p_kernel<256, -1, false, false, true, true><<<grid, 256, 0, s>>>(p);
    else if (cot_eff == 256) {
      if (y_split && out_fmt == 3 && !residual && !y_f32 && p.Cin == 128) conv256p_kernel<256, 3, false, false, false, false, 1><<<grid, 256, 0, s>>>(p);
      else if (y_split && out_fmt == 3 && !residual && !y_f32) conv256p_kernel<256, 3, false, false, false><<<grid, 256, 0, s>>>(p);
      else if (y_split && out_fmt == 3 && !residual && y_f32) conv256p_kernel<256, 3, false, true, false><<<grid, 256, 0, s>>>(p);
      else if (y_split && out_fmt == 3 && residual && y_f32) conv256p_kernel<256, 3, true, true, false><<<grid, 256, 0, s>>>(p);
      else if (y_split && out_fmt == 3 && residual && !y_f32) conv256p_kernel<256, 3, true, false, false><<<grid, 256, 0, s>>>(p);
      else conv256p_kernel<256, -1, false, false, true><<<grid, 256, 0, s>>>(p);
    } else {
      if (y_split && out_fmt == 3 && !residual && !y_f32) conv256p_kernel<128, 3, false, false, false><<<grid, 256, 0, s>>>(p);
      else if (y_split && out_fmt == 3 && residual && y_f32) conv256p_kernel<128, 3, true, true, false><<<grid, 256, 0, s>>>(p);
      else if (y_split && out_fmt == 3 && residual && !y_f32) conv256p_kernel<128, 3, true, false, false><<<grid, 256, 0, s>>>(p);
      else conv256p_kernel<128, -1, false, false, true><<<grid, 256, 0, s>>>(p);
    }
    FGVC_CHECK_LAUNCH("fgvc_conv_split_f32");
    return FGVC_OK;
  }
  if (in_fmt == 1) conv_split_dispatch<1>(p, grid, KS, cot_eff, narrow, s);
  else if (in_fmt == 3) conv_split_dispatch<3>(p, grid, KS, cot_eff, narrow, s);
  else if (in_fmt == 2) conv_split_dispatch<2>(p, grid, KS, cot_eff, narrow, s);
  else if (KS == 3 && cot_eff == 256 && !(g_conv_debug & 16) && !y_bank && !x2) conv_split_kernel<3, 256, 1, 3, 4, true><<<grid, 512, 0, s>>>(p);   // hand-placed operand reads
  else conv_split_dispatch<0>(p, grid, KS, cot_eff, narrow, s);                                                                  // (16: the compiler's operand schedule)
  FGVC_CHECK_LAUNCH("fgvc_conv_split_f32");
  return FGVC_OK;
}

int nchw_to_split_nhwc_launch(const float* in, uint16_t* out, float* out_f32, int N, int C, int H, int W, int Hp, int Wp,
                              hipStream_t s) {
  dim3 grid(cdiv(W, 256), H, N * (C / 32));
  nchw_to_split_nhwc_kernel<<<grid, 256, 0, s>>>(in, out, out_f32, C, H, W, Hp, Wp);
  FGVC_CHECK_LAUNCH("fgvc_nchw_to_split_nhwc_f32");
  return FGVC_OK;
}

// dense NHWC f32 [n][H][W][C] -> padded split NHWC, optionally through a ReLU that is also written back in place
__global__ __launch_bounds__(256) void nhwc_to_split_kernel(float* __restrict__ x, uint16_t* __restrict__ out, int C, int H,
                                                             int W, int Hp, int Wp, int relu, long long n_vec4) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= n_vec4) return;
  const long long e = g * 4;
  const long long pixel = e / C;
  const int c = (int)(e - pixel * C);
  const int nimg = (int)(pixel / ((long long)H * W));
  const int rem = (int)(pixel - (long long)nimg * H * W);
  const int y = rem / W, xx = rem - y * W;
  f32x4 v = *reinterpret_cast<const f32x4*>(x + e);
  if (relu) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    *reinterpret_cast<f32x4*>(x + e) = v;
  }
  ushort4 hv, lv;
  split_bf16_4(v, hv, lv);
  uint16_t* o = out + ((((size_t)nimg * Hp + (y + 1)) * Wp + (xx + 1)) * (C / 32) + (c >> 5)) * 64 + (c & 31);
  *reinterpret_cast<ushort4*>(o) = hv;
  *reinterpret_cast<ushort4*>(o + 32) = lv;
}

int nhwc_to_split_launch(float* x, uint16_t* out, int N, int C, int H, int W, int Hp, int Wp, int relu, hipStream_t s) {
  const long long n4 = (long long)N * H * W * C / 4;
  nhwc_to_split_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, s>>>(x, out, C, H, W, Hp, Wp, relu, n4);
  FGVC_CHECK_LAUNCH("fgvc_nhwc_to_split_f32");
  return FGVC_OK;
}

int normalize_nhwc_launch(const float* in, float* out, uint16_t* out_split, int N, int C, int H, int W, int normalize,
                          int split_fmt, hipStream_t s) {
  const long long npix = (long long)N * H * W;
  if (split_fmt == 2 || split_fmt == 3)      // 3: the 2 KiB rows of fgvc_split_f16f6x
    normalize_f16f6p_kernel<<<(unsigned)((npix + 3) / 4), 256, 0, s>>>(in, reinterpret_cast<unsigned char*>(out_split), normalize, npix, split_fmt == 3 ? 2048 : 1024);
  else if (split_fmt == 0) normalize_nhwc_kernel<0><<<(unsigned)((npix + 3) / 4), 256, 0, s>>>(in, out, out_split, C, normalize, npix);
  else normalize_nhwc_kernel<1><<<(unsigned)((npix + 3) / 4), 256, 0, s>>>(in, out, out_split, C, normalize, npix);
  FGVC_CHECK_LAUNCH("fgvc_normalize_nhwc_f32");
  return FGVC_OK;
}

}  // namespace fgvc
