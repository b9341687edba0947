// Microbenchmark: a dependent chain of 48 v_mfma_f32_32x32x16_f16 (volatile asm) whose A operands come from LDS two K-16 steps
// ahead (2 ds_read_b128 per 3 MFMAs, rows of 1040 bytes: conflict-free), as in pair_topk_v5.hip -- cycles per MFMA
//   mode 0: 4 waves per workgroup (one per SIMD)
//   mode 1: 8 waves, waves 4-7 poll an LDS word with s_sleep (the producers' wait loop)
//   mode 2: 8 waves, waves 4-7 stream global memory into the LDS by LDS-DMA (8 KiB per wave per round)
//   mode 3: 8 waves, waves 4-7 run a stream of vector min / max / xor (a selection network's instruction mix) until the chains end:
//           how many vector instructions does a PARTNER wave get per MFMA slot, and what does that cost the chain?
//   mode 4: 12 waves: chains, vector partners and LDS-DMA streamers together
// hipcc --offload-arch=gfx950 -O3 -o mfma_chain_lds tools/micro/mfma_chain_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(MODE == 4 ? 768 : 512, 1) void k(long long* out, const unsigned char* src, int reps) {
  constexpr int LDB = 1040;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 32 * LDB];
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4 * 32 * LDB / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
  if (tid == 0) flag = 0;
  __syncthreads();
  if ((MODE == 3 || MODE == 4) && wave >= 4 && wave < 8) {
    unsigned v[16];
    for (int i = 0; i < 16; ++i) v[i] = lane * 2654435761u + i * 40503u;
    long long iters = 0;
    for (int it = 0; it < (1 << 20); ++it) {
      if (__builtin_amdgcn_readfirstlane(*(volatile int*)&flag) >= 4) break;
#pragma unroll
      for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
        for (int d = 1; d < 16; d <<= 1)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if ((i & d) == 0) {
              const unsigned a = v[i], b = v[i | d];
              v[i] = a < b ? a : b;
              v[i | d] = a < b ? b : a;
            }
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] ^= (unsigned)(it + i);
      }
      ++iters;
    }
    unsigned x = 0;
    for (int i = 0; i < 16; ++i) x ^= v[i];
    if (x == 0x12345u) out[3] = 1;
    if (tid == 256 && blockIdx.x == 0) out[1] = iters * 4 * (32 * 2 + 16);     // vector instructions issued by this wave
    return;
  }
  if (wave >= 4) {
    if (MODE == 1) {
      for (int it = 0; it < (1 << 20); ++it) {
        if (__builtin_amdgcn_readfirstlane(*(volatile int*)&flag) >= 4) break;
        __builtin_amdgcn_s_sleep(2);
      }
    } else if (MODE == 2 || MODE == 4) {
      for (int it = 0; it < (1 << 20); ++it) {
        if (__builtin_amdgcn_readfirstlane(*(volatile int*)&flag) >= 4) break;
        for (int i = 0; i < 8; ++i) {
          const unsigned char* s = src + ((size_t)((blockIdx.x * 4 + (wave & 3)) * 8 + i + 64 * (it & 63)) * 1024) + 16 * lane;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                           (__attribute__((address_space(3))) void*)&smem[(3 * 32 + (wave & 3) * 8 + i) * LDB], 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    return;
  }
  const int n = lane & 31, hi = lane >> 5;
  f16x8 b;
  for (int i = 0; i < 8; ++i) b[i] = (_Float16)(i * 0.5f - lane * 0.002f);
  f32x16 acc;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    const unsigned char* ka = &smem[(r & 1) * 32 * LDB + n * LDB + 16 * hi];
#ifdef RING2
    // two-deep fragment rings: ah[j + 2] is read right after the second MFMA of step j (the last reader of ah[j]), al[j + 2] after the third
    f16x8 ah[2], al[2];
    for (int i = 0; i < 2; ++i) {
      ah[i] = *reinterpret_cast<const f16x8*>(ka + 32 * i);
      al[i] = *reinterpret_cast<const f16x8*>(ka + 512 + 32 * i);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(ah[0]), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah[j & 1]), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah[j & 1]), "v"(b));
      if (j + 2 < 16) ah[j & 1] = *reinterpret_cast<const f16x8*>(ka + 32 * (j + 2));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(al[j & 1]), "v"(b));
      if (j + 2 < 16) al[j & 1] = *reinterpret_cast<const f16x8*>(ka + 512 + 32 * (j + 2));
    }
#else
    f16x8 ah[3], al[3];
    for (int i = 0; i < 2; ++i) {
      ah[i] = *reinterpret_cast<const f16x8*>(ka + 32 * i);
      al[i] = *reinterpret_cast<const f16x8*>(ka + 512 + 32 * i);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j + 2 < 16) {
        ah[(j + 2) % 3] = *reinterpret_cast<const f16x8*>(ka + 32 * (j + 2));
        al[(j + 2) % 3] = *reinterpret_cast<const f16x8*>(ka + 512 + 32 * (j + 2));
      }
      if (j == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(ah[0]), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah[j % 3]), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah[j % 3]), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(al[j % 3]), "v"(b));
    }
#endif
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
    if (acc[0] == 12345.f) out[2] = 1;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) __hip_atomic_fetch_add(&flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE>
void run(long long* d, const unsigned char* src, const char* what) {
  const int reps = 100;
  hipMemset(d, 0, 64);
  k<MODE><<<256, MODE == 4 ? 768 : 512>>>(d, src, reps);
  k<MODE><<<256, MODE == 4 ? 768 : 512>>>(d, src, reps);
  long long h[2];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("%s: %.1f cycles per MFMA", what, (double)h[0] / (reps * 48));
  if (MODE >= 3) printf(", partner: %.1f vector instructions per MFMA slot (%.0f per 48-MFMA tile)", (double)h[1] / (reps * 48), (double)h[1] / reps);
  printf("\n");
}

int main() {
  long long* d;
  unsigned char* src;
  hipMalloc(&d, 64);
  hipMalloc(&src, (size_t)256 * 4 * 8 * 64 * 1024 + (1 << 20));
  run<0>(d, src, "one wave per SIMD, A from LDS");
  run<1>(d, src, "+ a partner wave polling an LDS word with s_sleep");
  run<2>(d, src, "+ a partner wave streaming by LDS-DMA");
  run<3>(d, src, "+ a partner wave issuing vector min/max/xor");
  run<4>(d, src, "+ a vector partner AND an LDS-DMA streamer (3 waves per SIMD)");
  return 0;
}
