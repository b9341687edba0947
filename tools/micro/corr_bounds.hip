// Microbenchmark for the dense-volume kernel's design space (no correctness: operands are random bits).  One launch has the
// shape of the real problem (25680 x 25680 outputs, C = 256): every wave keeps QT tiles of 32 query fragments in registers and
// walks `kchunk` key blocks whose A fragments it reads from a (static) LDS image; what varies:
//   PROD  0 bf16 x1 (16 MFMA / tile)   1 bf16 x3 (48)   2 f16 + two block-scaled fp8 cross products (16 + 4 + 4, = 32 bf16-MFMA
//         times)   3 f16 x2 (32)
//   QT    query tiles per wave (A fragments shared between them)
//   NW    waves per workgroup (4 = one per SIMD, 8 = two)
//   ST    0 no stores, 1 dword nt stores of the accumulator as it lies (2 x 128 B per instruction), 2 plain dword stores,
//         3 (QT = 2) v_permlane32_swap -> one 256-B row piece per instruction, nt, 4 the same, plain
// Prints ms per volume and the implied TB/s of the 2.64 GB write.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 256, KS = C / 16, LDB = 1024 + 16;

template <int PROD, int QT, int NW, int ST, int BAR = 0, int STAG = 0, int DMA = 0>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void kern(const uint4* __restrict__ qsrc, float* __restrict__ vol, int HW,
                                                                 int pitch, int kchunk) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * LDB];
  __shared__ __attribute__((aligned(16))) unsigned char lds2[DMA ? 64 * LDB : 16];
  for (int i = threadIdx.x; i < 64 * LDB / 4; i += NW * 64) reinterpret_cast<uint32_t*>(lds)[i] = (i * 2654435761u) & 0x3bff3bffu;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, hi = lane >> 5;
  const int q0 = (blockIdx.x * NW + wave) * 32 * QT;
  // B operands: QT tiles x (hi 16 frags + lo 16 frags) = QT x 128 VGPRs
  uint4 bq[QT][2 * KS];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int j = 0; j < 2 * KS; ++j) {
      uint4 v = qsrc[((size_t)(q0 + 32 * t + n) % 4096) * 64 + 2 * j + hi];
      v.x &= 0x3bff3bffu; v.y &= 0x3bff3bffu; v.z &= 0x3bff3bffu; v.w &= 0x3bff3bffu;     // finite f16 / bf16 / fp8 patterns
      bq[t][j] = v;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t abase = (uint32_t)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds + n * LDB + 16 * hi;
  const int kb0 = blockIdx.y * kchunk;
  f32x16 acc[QT];
  bool pend = false;
  const bool defer = STAG && wave >= NW / 2;
  for (int kb = kb0; kb < kb0 + kchunk; ++kb) {
    if (defer && pend) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (kb - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float* p = &vol[(size_t)row * pitch + q0 + n];
        if (row < HW && q0 + n < HW) __builtin_nontemporal_store(acc[0][r], p);
      }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const uint32_t a0 = abase + (kb & 1) * 32 * LDB;
    constexpr int G = 4, NG = KS / G;
    uint4 ah[2][G], al[2][G];
    auto load_group = [&](int g, int slot) {
#pragma unroll
      for (int i = 0; i < G; ++i) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[slot][i]) : "v"(a0), "i"(32 * (g * G + i)) : "memory");
        if constexpr (PROD != 0)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[slot][i]) : "v"(a0), "i"(512 + 32 * (g * G + i)) : "memory");
      }
    };
    load_group(0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) {
        load_group(g + 1, (g + 1) & 1);
        if constexpr (PROD != 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int j = g * G + i;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          const bf16x8 kh = __builtin_bit_cast(bf16x8, ah[g & 1][i]);
          const bf16x8 kl = __builtin_bit_cast(bf16x8, al[g & 1][i]);
          const bf16x8 qh = __builtin_bit_cast(bf16x8, bq[t][j]);
          const bf16x8 ql = __builtin_bit_cast(bf16x8, bq[t][KS + j]);
          if constexpr (PROD == 4) {
            // 16x16 shapes: the 32x32 region = 2x2 tiles of 16x16; per k16 step pair (K = 32) 4 MFMAs 16x16x32; timing only
            f32x4* a4 = reinterpret_cast<f32x4*>(&acc[t]);
            if (i & 1) {
              const f16x8 k0 = __builtin_bit_cast(f16x8, ah[g & 1][i - 1]), k1 = __builtin_bit_cast(f16x8, ah[g & 1][i]);
              const f16x8 q0 = __builtin_bit_cast(f16x8, bq[t][j - 1]), q1 = __builtin_bit_cast(f16x8, bq[t][j]);
              a4[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, q0, a4[0], 0, 0, 0);
              a4[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, q1, a4[1], 0, 0, 0);
              a4[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, q0, a4[2], 0, 0, 0);
              a4[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, q1, a4[3], 0, 0, 0);
            }
            if ((i & 3) == 3) {
              i32x8 a8, b8, c8, d8;
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                a8[2 * u + 0] = (int)al[g & 1][u].x; a8[2 * u + 1] = (int)al[g & 1][u].y;
                c8[2 * u + 0] = (int)al[g & 1][u].z; c8[2 * u + 1] = (int)al[g & 1][u].w;
                const uint4 w = __builtin_bit_cast(uint4, bq[t][KS + g * G + u]);
                b8[2 * u + 0] = (int)w.x; b8[2 * u + 1] = (int)w.y; d8[2 * u + 0] = (int)w.z; d8[2 * u + 1] = (int)w.w;
              }
              // K = 64 of the region per group: 2x2 tiles x K 128 instruction = half an instruction each -> 2 per product per group
              a4[0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, a4[0], 0, 0, 0, 0x78787878, 0, 0x6e6e6e6e);
              a4[3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(c8, d8, a4[3], 0, 0, 0, 0x78787878, 0, 0x6e6e6e6e);
              a4[1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, d8, a4[1], 0, 0, 0, 0x6e6e6e6e, 0, 0x78787878);
              a4[2] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(c8, b8, a4[2], 0, 0, 0, 0x6e6e6e6e, 0, 0x78787878);
            }
          } else if constexpr (PROD == 0) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh, acc[t], 0, 0, 0);
          } else if constexpr (PROD == 1) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh, acc[t], 0, 0, 0);
          } else if constexpr (PROD == 3) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kl), __builtin_bit_cast(f16x8, qh), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kh), __builtin_bit_cast(f16x8, qh), acc[t], 0, 0, 0);
          } else {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kh), __builtin_bit_cast(f16x8, qh), acc[t], 0, 0, 0);
            if ((i & 3) == 3) {   // per 4 k16-steps (K = 64): two scaled fp8 MFMAs, operands = the 8 "lo" registers of the group
              i32x8 a8, b8;
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                a8[2 * (u & 3) + 0] = (int)al[g & 1][u].x; a8[2 * (u & 3) + 1] = (int)al[g & 1][u].y;
                b8[2 * (u & 3) + 0] = (int)bq[t][KS + g * G + u].x; b8[2 * (u & 3) + 1] = (int)bq[t][KS + g * G + u].y;
              }
              acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 0, 0, 0, 0x78787878, 0, 0x6e6e6e6e);
              i32x8 c8, d8;
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                c8[2 * (u & 3) + 0] = (int)al[g & 1][u].z; c8[2 * (u & 3) + 1] = (int)al[g & 1][u].w;
                d8[2 * (u & 3) + 0] = (int)bq[t][KS + g * G + u].z; d8[2 * (u & 3) + 1] = (int)bq[t][KS + g * G + u].w;
              }
              acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(c8, d8, acc[t], 0, 0, 0, 0x6e6e6e6e, 0, 0x78787878);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (ST == 0) {
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[t][r]));
    } else if (defer) {
      pend = true;
    } else if constexpr (ST == 1 || ST == 2) {
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          float* p = &vol[(size_t)row * pitch + q0 + 32 * t + n];
          if (row < HW && q0 + 32 * t + n < HW) {
            if (ST == 1) __builtin_nontemporal_store(acc[t][r], p);
            else *p = acc[t][r];
          }
        }
    } else {
      static_assert(ST < 3 || QT == 2, "row-piece stores need two query tiles");
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float x = acc[0][r], y = acc[QT - 1][r];
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));      // x: row R, 64 queries; y: row R + 4, 64 queries
        const int row = kb * 32 + (r & 3) + 8 * (r >> 2);
        float* p = &vol[(size_t)row * pitch + q0 + lane];
        if (q0 + lane < HW) {
          if (row < HW) { if (ST == 3) __builtin_nontemporal_store(x, p); else *p = x; }
          if (row + 4 < HW) { if (ST == 3) __builtin_nontemporal_store(y, p + 4 * (size_t)pitch); else p[4 * (size_t)pitch] = y; }
        }
      }
    }
    if (DMA && !((kb - kb0) & 1)) {   // stage start: LDS-DMA of the next 64 key rows (8 x 1 KiB per wave)
#pragma unroll
      for (int i = 0; i < 64 / NW; ++i) {
        const int row = wave * (64 / NW) + i;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(qsrc) + ((size_t)((kb * 32 + row) % 4096)) * 1024 + 16 * lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)&lds2[row * LDB], 16, 0, 0);
      }
    }
    if (BAR && ((kb - kb0) & 1)) {
      if (DMA == 1) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
}

template <int PROD, int QT, int NW, int ST, int BAR = 0, int STAG = 0, int DMA = 0>
float run(const char* name, const uint4* q, float* vol, int HW, int pitch) {
  const int n_q = (HW + 32 * QT * NW - 1) / (32 * QT * NW), n_kb = (HW + 31) / 32;
  const int chunks = std::max(1, 1024 / n_q);
  int kchunk = (n_kb + chunks - 1) / chunks;
  kchunk += kchunk & 1;
  dim3 grid(n_q, (n_kb + kchunk - 1) / kchunk);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) kern<PROD, QT, NW, ST, BAR, STAG, DMA><<<grid, NW * 64>>>(q, vol, HW, pitch, kchunk);
  (void)hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) kern<PROD, QT, NW, ST, BAR, STAG, DMA><<<grid, NW * 64>>>(q, vol, HW, pitch, kchunk);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int HW = 25680;
  const int pitchA = 25696;   // multiple of 32 floats: every row starts on a 128-byte line
  uint4* q;
  float* vol;
  (void)hipMalloc(&q, 4096 * 64 * 16);
  (void)hipMalloc(&vol, (size_t)(HW + 64) * pitchA * 4);
  {
    uint32_t* h = (uint32_t*)malloc(4096 * 64 * 16);
    for (int i = 0; i < 4096 * 64 * 4; ++i) h[i] = (uint32_t)rand() * 2654435761u;
    (void)hipMemcpy(q, h, 4096 * 64 * 16, hipMemcpyHostToDevice);
    free(h);
  }
  struct Cfg { const char* name; float (*fn)(const char*, const uint4*, float*, int, int); int pitch; float best; };
  Cfg cfgs[] = {
    {"16x16 nt aligned + barrier + DMA", run<4, 1, 8, 1, 1, 0, 1>, pitchA, 1e9f},
    {"16x16 nt aligned + barrier + DMA + stagger", run<4, 1, 8, 1, 1, 1, 1>, pitchA, 1e9f},
    {"16x16 PLAIN aligned + barrier + DMA", run<4, 1, 8, 2, 1, 0, 1>, pitchA, 1e9f},
    {"16x16 nt aligned + barrier + DMA not waited for", run<4, 1, 8, 1, 1, 0, 2>, pitchA, 1e9f},
    {"16x16 nt aligned + barrier + DMA not waited + stagger", run<4, 1, 8, 1, 1, 1, 2>, pitchA, 1e9f},
    {"16x16 nt aligned + barrier, no DMA", run<4, 1, 8, 1, 1, 0, 0>, pitchA, 1e9f},
  };
  const int n = sizeof(cfgs) / sizeof(cfgs[0]);
  float first[16], last[16];
  for (int round = 0; round < 4; ++round)
    for (int c = 0; c < n; ++c) {
      const float ms = cfgs[c].fn(cfgs[c].name, q, vol, HW, cfgs[c].pitch);
      if (round == 0) first[c] = ms;
      last[c] = ms;
      cfgs[c].best = std::min(cfgs[c].best, ms);
    }
  for (int c = 0; c < n; ++c)
    printf("%-46s min %.3f ms (%.2f TB/s)  first round %.3f  last round %.3f\n", cfgs[c].name, cfgs[c].best, (double)HW * HW * 4 / cfgs[c].best / 1e9, first[c], last[c]);
  return 0;
}
