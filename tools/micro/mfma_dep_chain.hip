// What does a DEPENDENT chain of v_mfma_f32_32x32x16_f16 (one accumulator) cost per instruction, in s_memtime ticks and in wall time --
// alone in registers, with an LDS read per MFMA, with a counted wait per MFMA, mixed with scaled FP6 MFMAs; 1 wave per SIMD, 1 workgroup
// per CU.  (Round 4: the pair kernel's consumer measured 72 ticks per MFMA step.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x6 __attribute__((ext_vector_type(6)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* ticks, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * 1040];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 1040 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i % 97);
  __syncthreads();
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (lane + i)); b[i] = (_Float16)(0.02f * (lane - i)); }
  i32x6 p6 = {0x11111111, 0x22222222, 0x01010101, 0x10101010, 0x12121212, 0x21212121};
  int sc = 0x7f7f7f7f;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem + (lane & 31) * 1040 + 16 * (lane >> 5);
  f16x8 ring[4] = {a, a, a, a};
  asm volatile("" : "+v"(a), "+v"(b), "+v"(p6), "+v"(sc));
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (MODE == 0) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
      } else if (MODE == 1) {      // + one ds_read_b128 per MFMA into a ring, no waits inside (wait at the end of the 16)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[j & 3]) : "v"(addr), "n"(32 * j) : "memory");
      } else if (MODE == 2) {      // the pair kernel's pattern: counted wait, MFMA on the ring register, read four ahead
        asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ring[j & 3]), "v"(b));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[j & 3]) : "v"(addr), "n"(32 * j) : "memory");
      } else if (MODE == 3) {      // two f16 + one scaled FP6, registers only
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        if (j & 1) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 cbsz:2 blgp:2" : "+v"(acc) : "v"(p6), "v"(p6), "v"(sc));
      } else if (MODE == 4) {      // independent accumulators? no: same accumulator, but s_nop 0 in front like the compiler's padding
        asm volatile("s_nop 0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
      }
    }
    if (MODE == 1 || MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int r = 0; r < 4; ++r) s += (float)ring[r][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[MODE] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, long long* ticks, int nblk, int mfma_per_16) {
  const int iters = 2000;
  k<MODE><<<nblk, 256>>>(out, ticks, 10);
  hipDeviceSynchronize();
  auto w0 = std::chrono::high_resolution_clock::now();
  k<MODE><<<nblk, 256>>>(out, ticks, iters);
  hipDeviceSynchronize();
  auto w1 = std::chrono::high_resolution_clock::now();
  long long t;
  hipMemcpy(&t, ticks + MODE, 8, hipMemcpyDeviceToHost);
  const double us = std::chrono::duration<double, std::micro>(w1 - w0).count();
  const double n = (double)iters * mfma_per_16;
  printf("%-44s %7.1f ticks per MFMA, %7.2f ns per MFMA (wall, incl. launch), ticks per us %.0f\n", name, t / n, us * 1e3 / n, t / us);
}

int main() {
  float* out; long long* ticks;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 64);
  for (int nblk : {1, 256}) {
    printf("---- %d workgroup(s) of 4 waves\n", nblk);
    run<0>("dependent f16 chain, registers", out, ticks, nblk, 16);
    run<4>("... with s_nop 0 in front of each", out, ticks, nblk, 16);
    run<1>("+ ds_read_b128 per MFMA, no waits", out, ticks, nblk, 16);
    run<2>("counted wait + MFMA + read four ahead", out, ticks, nblk, 16);
    run<3>("2 f16 + 1 scaled FP6 (K 64)", out, ticks, nblk, 24);
  }
  return 0;
}
