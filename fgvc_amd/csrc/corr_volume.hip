// Dense (materialised) correlation volume  vol[j][i] = <k_j, q_i> / temperature  for one
// (query frame, key frame) pair -- what BASELINE.json's "ms/corr-volume" measures.
// Replaces affinity_utils.py:6-21, local_attention.py:231 and :321-323, correlation.py:51.
//
// Three arithmetic variants behind one output contract ([HWk][HWq] f32, row-major):
//   f32     v_mfma_f32_32x32x2_f32, exact f32 (compute-bound: 157 TF peak)
//   bf16x3  features pre-split x = hi + lo (both bf16); hi*hi + hi*lo + lo*hi on
//           v_mfma_f32_32x32x16_bf16 with f32 accumulation -> |err| ~ 2^-16 relative, i.e. within the
//           1e-3 score tolerance at 16x the MFMA rate (3 products -> ~5x the f32 kernel)
//   bf16    hi*hi only (reduced precision)
// The volume write (HWk*HWq*4 bytes, 2.6 GB at 480p stride 4) is the HBM-roofline term; stores
// are non-temporal so the L2-resident feature panels are not evicted by the output stream.
#include "common.hpp"

namespace fgvc {

// ------------------------------------------------------------------------------------------
// exact f32
// ------------------------------------------------------------------------------------------
constexpr int KCHUNK = 16;  // key blocks (of 32 pixels) per workgroup

template <int C>
__global__ __launch_bounds__(256, 2) void corr_volume_f32_kernel(const float* __restrict__ qfeat,
                                                                  const float* __restrict__ kfeat, int HWq,
                                                                  int HWk, float temperature,
                                                                  float* __restrict__ vol) {
  constexpr int LDK = C + 4;
  constexpr int BUF = 32 * LDK;
  constexpr int NLD = C / 32;
  __shared__ __attribute__((aligned(16))) float smem[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hi = lane >> 5;
  const int q = blockIdx.x * 128 + wave * 32 + n;
  float qreg[C / 2];
  {
    const float* qp = qfeat + (size_t)imin(q, HWq - 1) * C + 4 * hi;
#pragma unroll
    for (int j = 0; j < C / 8; ++j) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(qp + 8 * j);
      qreg[4 * j + 0] = t.x; qreg[4 * j + 1] = t.y; qreg[4 * j + 2] = t.z; qreg[4 * j + 3] = t.w;
    }
  }
  const int kb0 = blockIdx.y * KCHUNK;
  const int kb1 = imin(kb0 + KCHUNK, cdiv(HWk, 32));
  f32x4 stage[NLD];
  auto stage_load = [&](int kb) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = tid + 256 * i;
      const int row = f / (C / 4), c4 = f % (C / 4);
      const int pix = imin(kb * 32 + row, HWk - 1);
      stage[i] = *reinterpret_cast<const f32x4*>(kfeat + (size_t)pix * C + 4 * c4);
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
  stage_load(kb0);
  stage_store(0);
  __syncthreads();
  int buf = 0;
  for (int kb = kb0; kb < kb1; ++kb) {
    if (kb + 1 < kb1) stage_load(kb + 1);
    // two independent accumulator chains per wave x two waves per SIMD = four chains per matrix pipe:
    // a dependent f32 MFMA chain alone retires one MFMA per ~200 cycles (issue interval is 64)
    f32x16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc1 = acc0;
    const float* ka = &smem[buf * BUF + n * LDK + 4 * hi];
#pragma unroll
    for (int j = 0; j < C / 8; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ka + 8 * j);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[4 * j + 0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[4 * j + 1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[4 * j + 2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[4 * j + 3], acc1, 0, 0, 0);
    }
    const f32x16 acc = acc0 + acc1;
    if (q < HWq) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row < HWk) __builtin_nontemporal_store(acc[r] / temperature, &vol[(size_t)row * HWq + q]);
      }
    }
    if (kb + 1 < kb1) stage_store(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
}

// ------------------------------------------------------------------------------------------
// f32 -> (hi, lo) bf16 split.  out[pix][0][c] = bf16(x), out[pix][1][c] = bf16(x - hi)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t f2bf(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);  // RNE; inputs are finite
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ feat,
                                                          uint16_t* __restrict__ out, long long n_vec4, int C) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= n_vec4) return;
  const long long e = g * 4;
  const long long pix = e / C;
  const int c = (int)(e - pix * C);
  const f32x4 x = *reinterpret_cast<const f32x4*>(feat + e);
  ushort4 h, l;
  h.x = f2bf(x.x); h.y = f2bf(x.y); h.z = f2bf(x.z); h.w = f2bf(x.w);
  l.x = f2bf(x.x - bf2f(h.x)); l.y = f2bf(x.y - bf2f(h.y));
  l.z = f2bf(x.z - bf2f(h.z)); l.w = f2bf(x.w - bf2f(h.w));
  *reinterpret_cast<ushort4*>(out + pix * 2 * C + c) = h;
  *reinterpret_cast<ushort4*>(out + pix * 2 * C + C + c) = l;
}

// ------------------------------------------------------------------------------------------
// bf16 MFMA GEMM, 128 (keys) x 128 (queries) tile per workgroup, 2x2 waves of 64x64,
// K-step 64, LDS double buffered with register prefetch, rows padded to 144 B (conflict-free b128).
// NSEG = 3: K runs over [hi*hi | hi*lo | lo*hi];  NSEG = 1: hi*hi only.
// ------------------------------------------------------------------------------------------
template <int NSEG>
__global__ __launch_bounds__(256, 2) void corr_volume_bf16_kernel(const uint16_t* __restrict__ q_hl,
                                                                   const uint16_t* __restrict__ k_hl, int C,
                                                                   int HWq, int HWk, float temperature,
                                                                   float* __restrict__ vol, int n_qt) {
  constexpr int BK = 64;            // bf16 elements per K-step
  constexpr int ROWB = BK * 2 + 16; // padded row bytes
  constexpr int TILEB = 128 * ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILEB];  // [buf][A|B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int kt = tile / n_qt, qt = tile - kt * n_qt;   // consecutive tiles share the key panel
  const int k0 = kt * 128, q0 = qt * 128;

  const int ksteps_per_seg = C / BK;
  const int nsteps = NSEG * ksteps_per_seg;

  // staging: 128 rows x 128 B per operand = 1024 x 16 B -> 4 per thread per operand
  uint4 sa[4], sb[4];
  auto stage_load = [&](int step) {
    const int seg = step / ksteps_per_seg, kin = (step - seg * ksteps_per_seg) * BK;
    const int a_part = (seg == 2) ? 1 : 0;   // keys:    hi, hi, lo
    const int b_part = (seg == 1) ? 1 : 0;   // queries: hi, lo, hi
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      const int row = f >> 3, c16 = f & 7;
      const int kp = imin(k0 + row, HWk - 1), qp = imin(q0 + row, HWq - 1);
      sa[i] = *reinterpret_cast<const uint4*>(k_hl + ((size_t)kp * 2 + a_part) * C + kin + c16 * 8);
      sb[i] = *reinterpret_cast<const uint4*>(q_hl + ((size_t)qp * 2 + b_part) * C + kin + c16 * 8);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      const int row = f >> 3, c16 = f & 7;
      *reinterpret_cast<uint4*>(&smem[(buf * 2 + 0) * TILEB + row * ROWB + c16 * 16]) = sa[i];
      *reinterpret_cast<uint4*>(&smem[(buf * 2 + 1) * TILEB + row * ROWB + c16 * 16]) = sb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  stage_load(0);
  stage_store(0);
  __syncthreads();
  int buf = 0;
  for (int step = 0; step < nsteps; ++step) {
    if (step + 1 < nsteps) stage_load(step + 1);
    const unsigned char* A = &smem[(buf * 2 + 0) * TILEB + (wm * 64 + l31) * ROWB + hi * 16];
    const unsigned char* B = &smem[(buf * 2 + 1) * TILEB + (wn * 64 + l31) * ROWB + hi * 16];
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        af[t] = *reinterpret_cast<const bf16x8*>(A + t * 32 * ROWB + kk * 32);
        bfr[t] = *reinterpret_cast<const bf16x8*>(B + t * 32 * ROWB + kk * 32);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
    }
    if (step + 1 < nsteps) stage_store(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // epilogue: row (key) = (r&3) + 8*(r>>2) + 4*hi, column (query) = lane&31 -> 128 B contiguous per half-wave
  const float inv_t = 1.0f / temperature;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int qcol = q0 + wn * 64 + ni * 32 + l31;
      if (qcol < HWq) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int krow = k0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (krow < HWk) __builtin_nontemporal_store(acc[mi][ni][r] * inv_t, &vol[(size_t)krow * HWq + qcol]);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------
int corr_volume_f32_launch(const float* q, const float* k, int C, int HWq, int HWk, float temperature, float* vol,
                           hipStream_t s) {
  dim3 grid(cdiv(HWq, 128), cdiv(cdiv(HWk, 32), KCHUNK));
  switch (C) {
    case 32: corr_volume_f32_kernel<32><<<grid, 256, 0, s>>>(q, k, HWq, HWk, temperature, vol); break;
    case 64: corr_volume_f32_kernel<64><<<grid, 256, 0, s>>>(q, k, HWq, HWk, temperature, vol); break;
    case 128: corr_volume_f32_kernel<128><<<grid, 256, 0, s>>>(q, k, HWq, HWk, temperature, vol); break;
    case 256: corr_volume_f32_kernel<256><<<grid, 256, 0, s>>>(q, k, HWq, HWk, temperature, vol); break;
    default:
      set_error("fgvc_corr_volume_f32: C=%d unsupported (32, 64, 128 or 256)", C);
      return FGVC_ERR_UNSUPPORTED;
  }
  FGVC_CHECK_LAUNCH("fgvc_corr_volume_f32");
  return FGVC_OK;
}

int split_bf16_launch(const float* feat, uint16_t* out, long long n_pixels, int C, hipStream_t s) {
  const long long n4 = n_pixels * C / 4;
  split_bf16_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, s>>>(feat, out, n4, C);
  FGVC_CHECK_LAUNCH("fgvc_split_bf16");
  return FGVC_OK;
}

int corr_volume_bf16_launch(const uint16_t* q, const uint16_t* k, int C, int HWq, int HWk, float temperature,
                            float* vol, int nseg, hipStream_t s) {
  const int n_qt = cdiv(HWq, 128), n_kt = cdiv(HWk, 128);
  if (nseg == 3)
    corr_volume_bf16_kernel<3><<<n_qt * n_kt, 256, 0, s>>>(q, k, C, HWq, HWk, temperature, vol, n_qt);
  else
    corr_volume_bf16_kernel<1><<<n_qt * n_kt, 256, 0, s>>>(q, k, C, HWq, HWk, temperature, vol, n_qt);
  FGVC_CHECK_LAUNCH("fgvc_corr_volume_bf16");
  return FGVC_OK;
}

}  // namespace fgvc
