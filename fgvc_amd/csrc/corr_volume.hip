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
                                                                  float* __restrict__ vol, int kchunk) {
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
  // Force the query loads to be waited for HERE: left alone, hipcc waits for them lazily inside the loop
  // (s_waitcnt vmcnt(29..0) between the MFMAs), and on every later iteration those small counts also
  // wait for the freshly issued key loads and the previous tile's stores -- a memory round trip per tile.
#pragma unroll
  for (int j = 0; j < C / 2; ++j) asm volatile("" ::"v"(qreg[j]));
  const int kb0 = blockIdx.y * kchunk;
  const int kb1 = imin(kb0 + kchunk, cdiv(HWk, 32));
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
    lds_barrier();   // not __syncthreads(): must not wait for the volume stores above
    buf ^= 1;
  }
}

// ------------------------------------------------------------------------------------------
// f32 -> (hi, lo) bf16 split.  out[pix][0][c] = bf16(x), out[pix][1][c] = bf16(x - hi)
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ feat,
                                                          uint16_t* __restrict__ out, long long n_vec4, int C) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= n_vec4) return;
  const long long e = g * 4;
  const long long pix = e / C;
  const int c = (int)(e - pix * C);
  const f32x4 x = *reinterpret_cast<const f32x4*>(feat + e);
  ushort4 h, l;
  split_bf16_4(x, h, l);
  *reinterpret_cast<ushort4*>(out + pix * 2 * C + c) = h;
  *reinterpret_cast<ushort4*>(out + pix * 2 * C + C + c) = l;
}

// ------------------------------------------------------------------------------------------
// bf16 / split-bf16x3 volume.  Same decomposition as the f32 kernel above: a wave keeps its 32 query
// vectors resident in VGPRs as the MFMA B operand (hi and lo parts: 2 x C/4 registers), key blocks of 32
// pixels ([pixel][hi|lo][C] bf16 = one contiguous 4*C-byte row per pixel) are staged once per workgroup into
// padded LDS rows and read as ds_read_b128 A fragments; the 32x32 f32 tile goes straight from the
// accumulator to memory (query on the lane -> 128-B row segments, non-temporal).
//   NSEG == 3:  acc = k_hi*q_lo + k_lo*q_hi + k_hi*q_hi  (small terms first), 3*C/16 chained
//               v_mfma_f32_32x32x16_bf16 per tile -- the f32-accurate product at bf16 MFMA rate
//   NSEG == 1:  acc = k_hi*q_hi
// For a short K (= C = 256) this beats a classic LDS-tiled GEMM: there is no K loop to pipeline, the
// per-tile work is one straight MFMA chain, and the only recurring memory traffic besides the output
// stream is the key block (shared by the workgroup's four waves, L2-resident).
// ------------------------------------------------------------------------------------------
template <int C, int NSEG, int NW, int SUB>
__global__ __launch_bounds__(NW * 64, (NW == 8) ? 2 : 2) void corr_volume_bf16_kernel(const uint16_t* __restrict__ q_hl,
                                                                   const uint16_t* __restrict__ k_hl, int HWq,
                                                                   int HWk, float temperature,
                                                                   float* __restrict__ vol, int debug, int kchunk) {
  // C == 256: a key pixel's [hi|lo] row is exactly 1 KiB = one LDS-DMA wave instruction, so the whole row is
  // staged (also for NSEG == 1, which then simply ignores the lo half) with zero staging registers.
  constexpr bool DMA = (C == 256);
  constexpr int PARTS = (NSEG == 3 || DMA) ? 2 : 1;   // parts of a key row that are staged (hi [, lo])
  constexpr int ROWB = PARTS * C * 2;                 // bytes of one staged key row
  constexpr int LDB = ROWB + 16;                      // padded LDS row (bytes): stride 16*odd -> conflict-free b128
  constexpr int ROWS = 32 * SUB;                      // key rows per pipeline stage
  constexpr int BUFB = ROWS * LDB;
  constexpr int NT = NW * 64;
  constexpr int NLD = DMA ? 1 : (ROWS * ROWB / 16) / NT;   // 16-byte register-staged loads per thread (non-DMA)
  constexpr int KS = C / 16;                          // k16 steps per part
  static_assert(DMA || (ROWS * ROWB / 16) % NT == 0, "staging does not divide");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hi = lane >> 5;
  const int q = blockIdx.x * (NW * 32) + wave * 32 + n;

  // query fragments: lane (n, hi) holds k = 16j + 8hi .. +7 of query n, hi part [and lo part]
  bf16x8 qh[KS], ql[(NSEG == 3) ? KS : 1];
  {
    const uint16_t* qp = q_hl + (size_t)imin(q, HWq - 1) * 2 * C + 8 * hi;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      qh[j] = *reinterpret_cast<const bf16x8*>(qp + 16 * j);
      if constexpr (NSEG == 3) ql[j] = *reinterpret_cast<const bf16x8*>(qp + C + 16 * j);
    }
  }
  // wait for the query fragments before the loop, not lazily inside it (see the f32 kernel)
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    asm volatile("" ::"v"(qh[j]));
    if constexpr (NSEG == 3) asm volatile("" ::"v"(ql[j]));
  }
  const int kb0 = blockIdx.y * kchunk;               // in units of 32-key blocks; a stage covers SUB of them
  const int kb1 = imin(kb0 + kchunk, cdiv(HWk, 32));
  uint4 stage[NLD];
  auto stage_load = [&](int kb, int buf) {
    if constexpr (DMA) {
#pragma unroll
      for (int i = 0; i < ROWS / NW; ++i) {
        const int row = wave * (ROWS / NW) + i;
        const int pix = imin(kb * 32 + row, HWk - 1);
        const uint16_t* src = k_hl + (size_t)pix * 2 * C + 8 * lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)&smem[buf * BUFB + row * LDB],
                                         16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int f = tid + NT * i;
        const int row = f / (ROWB / 16), c16 = f % (ROWB / 16);
        const int pix = imin(kb * 32 + row, HWk - 1);
        stage[i] = *reinterpret_cast<const uint4*>(k_hl + (size_t)pix * 2 * C + 8 * c16);   // hi part first, lo follows
      }
    }
  };
  auto stage_store = [&](int buf) {
    if constexpr (!DMA) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int f = tid + NT * i;
        const int row = f / (ROWB / 16), c16 = f % (ROWB / 16);
        *reinterpret_cast<uint4*>(&smem[buf * BUFB + row * LDB + 16 * c16]) = stage[i];
      }
    }
  };
  stage_load(kb0, 0);
  stage_store(0);
  __syncthreads();
  int buf = 0;
  const float inv_t = 1.0f / temperature;
  // wave-uniform: all 32 queries of this wave exist -> the 16 stores of a full key block are unconditional,
  // so their count is known and the DMA can be waited for with a counted vmcnt that leaves them in flight
  const bool wave_full = (blockIdx.x * (NW * 32) + wave * 32 + 31) < HWq;
  for (int kb = kb0; kb < kb1; kb += SUB) {
    const bool more = kb + SUB < kb1;
    if (more) stage_load(kb + SUB, buf ^ 1);
    bool all_full = true;
#pragma unroll
    for (int sb = 0; sb < SUB; ++sb) {
      if (kb + sb >= kb1) break;                       // wave-uniform (ragged tail of the chunk)
      // three independent accumulator chains (hi*lo, lo*hi, hi*hi): a dependent v_mfma_f32_32x32x16_bf16 does not
      // issue back to back here (PMC: pipe 45 % busy, waves 60 % in SQ_WAIT_INST_ANY with two chains);
      // summing the small correction terms separately is also the better rounding order
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      f32x16 accx = acc, accy = acc;
      const unsigned char* ka = &smem[buf * BUFB + (sb * 32 + n) * LDB + 16 * hi];   // A operand: key row
      // A operands one group of G k16-steps ahead, by inline assembly with explicit waits: for loads it knows about hipcc
      // merges the waits of two groups into one `s_waitcnt lgkmcnt(0)` placed AFTER the next group's reads were issued, i.e.
      // every other group waits a full LDS round trip (25 % of the 12 MFMAs between them).  Here: wait for group g (the
      // only LDS operations outstanding), then issue group g + 1, then multiply group g.
      constexpr int G = 4;                               // k16 steps per group: 12 MFMAs (NSEG == 3) cover the LDS latency
      constexpr int NG = KS / G;
      bf16x8 ah[2][G], al[2][(NSEG == 3) ? G : 1];
      const uint32_t ka_lds = (uint32_t)(size_t)(const __attribute__((address_space(3))) unsigned char*)ka;
      auto load_group = [&](int g, int slot) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[slot][i]) : "v"(ka_lds), "i"(32 * (g * G + i)) : "memory");
          if constexpr (NSEG == 3)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[slot][i]) : "v"(ka_lds), "i"(C * 2 + 32 * (g * G + i)) : "memory");
        }
      };
      load_group(0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < NG) load_group(g + 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const int j = g * G + i;
          if constexpr (NSEG == 3) {
            accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g & 1][i], ql[j], accx, 0, 0, 0);
            accy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[g & 1][i], qh[j], accy, 0, 0, 0);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g & 1][i], qh[j], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (NSEG == 3) acc += (accx + accy);
      const int kbs = kb + sb;
      const bool full = wave_full && (kbs * 32 + 32 <= HWk);               // wave-uniform
      all_full = all_full && full;
      if (debug & 1) {   // profiling ablation: no volume stores (results are not written)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[r]));
        all_full = false;
      } else if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = kbs * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          __builtin_nontemporal_store(acc[r] * inv_t, &vol[(size_t)row * HWq + q]);
        }
      } else if (q < HWq) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = kbs * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (row < HWk) __builtin_nontemporal_store(acc[r] * inv_t, &vol[(size_t)row * HWq + q]);
        }
      }
    }
    if constexpr (DMA) {
      // a full stage issued exactly 16*SUB stores after this wave's DMA loads: the counted wait retires the
      // DMA (and everything older) but leaves those stores in flight across the barrier
      if (all_full && kb + SUB <= kb1) {
        if constexpr (SUB == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    if (more) stage_store(buf ^ 1);
    lds_barrier();   // not __syncthreads(): must not wait for the volume stores above
    buf ^= 1;
  }
}

// ------------------------------------------------------------------------------------------
int corr_volume_f32_launch(const float* q, const float* k, int C, int HWq, int HWk, float temperature, float* vol,
                           hipStream_t s) {
  // key blocks per workgroup: long workgroups amortise the query-fragment prologue (see corr_volume_bf16_launch); two
  // workgroups share a CU here, so the grid should cover 512 slots about 4 times
  const int n_q = cdiv(HWq, 128), n_kb = cdiv(HWk, 32);
  const int kchunk = imax(KCHUNK, cdiv(n_kb, imax(1, 2048 / n_q)));
  dim3 grid(n_q, cdiv(n_kb, kchunk));
  switch (C) {
    case 32: corr_volume_f32_kernel<32><<<grid, 256, 0, s>>>(q, k, HWq, HWk, temperature, vol, kchunk); break;
    case 64: corr_volume_f32_kernel<64><<<grid, 256, 0, s>>>(q, k, HWq, HWk, temperature, vol, kchunk); break;
    case 128: corr_volume_f32_kernel<128><<<grid, 256, 0, s>>>(q, k, HWq, HWk, temperature, vol, kchunk); break;
    case 256: corr_volume_f32_kernel<256><<<grid, 256, 0, s>>>(q, k, HWq, HWk, temperature, vol, kchunk); break;
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

static int g_corr_debug = 0;
void set_corr_debug(int v) { g_corr_debug = v; }

int corr_volume_bf16_launch(const uint16_t* q, const uint16_t* k, int C, int HWq, int HWk, float temperature,
                            float* vol, int nseg, hipStream_t s) {
  // C == 256: 8-wave workgroups (256 queries), 64-key stages through LDS-DMA; narrower features keep the
  // 4-wave / 32-key register-staged form
  if (C == 256) {
    // key blocks per workgroup: one workgroup per CU (248 VGPRs), and every workgroup pays ~4 us of prologue (128 VGPRs of
    // query fragments + the first stage) before its first MFMA -- as few, long workgroups as still fill the 256 CUs about
    // 4 times: at 480p 81 blocks = 10 x 101 workgroups (3.95 rounds) 1.04 ms against 1.14 ms for 16 blocks (20.1 rounds)
    const int n_q = cdiv(HWq, 256), n_kb = cdiv(HWk, 32);
    const int chunks = imax(1, 1024 / n_q);
    int kchunk = imax(16, cdiv(n_kb, chunks));
    kchunk += kchunk & 1;                                 // whole 64-key stages
    if (g_corr_debug >> 8) kchunk = g_corr_debug >> 8;    // tools/experiments/ablate_corr.py sweeps it
    dim3 grid(n_q, cdiv(n_kb, kchunk));
    if (nseg == 3)
      corr_volume_bf16_kernel<256, 3, 8, 2><<<grid, 512, 0, s>>>(q, k, HWq, HWk, temperature, vol, g_corr_debug & 255, kchunk);
    else
      corr_volume_bf16_kernel<256, 1, 8, 2><<<grid, 512, 0, s>>>(q, k, HWq, HWk, temperature, vol, g_corr_debug & 255, kchunk);
  } else {
    dim3 grid(cdiv(HWq, 128), cdiv(cdiv(HWk, 32), KCHUNK));
#define FGVC_BF(CC)                                                                                       \
  if (nseg == 3)                                                                                          \
    corr_volume_bf16_kernel<CC, 3, 4, 1><<<grid, 256, 0, s>>>(q, k, HWq, HWk, temperature, vol, g_corr_debug & 255, KCHUNK);         \
  else                                                                                                    \
    corr_volume_bf16_kernel<CC, 1, 4, 1><<<grid, 256, 0, s>>>(q, k, HWq, HWk, temperature, vol, g_corr_debug & 255, KCHUNK)
    switch (C) {
      case 64: FGVC_BF(64); break;
      case 128: FGVC_BF(128); break;
      default:
        set_error("fgvc_corr_volume_bf16: C=%d unsupported (64, 128 or 256)", C);
        return FGVC_ERR_UNSUPPORTED;
    }
#undef FGVC_BF
  }
  FGVC_CHECK_LAUNCH("fgvc_corr_volume_bf16");
  return FGVC_OK;
}

}  // namespace fgvc
