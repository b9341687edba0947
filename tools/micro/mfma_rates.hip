// Bare matrix-pipe rates (operands in registers, s_memtime around 256 instructions): cycles per instruction for the shapes the
// dense volume kernel uses, alone and mixed, one wave per SIMD (256 threads) and two (512).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(long long* out, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (lane + j)); b[j] = (_Float16)(0.02f * (lane - j)); }
  i32x8 a8, b8;
  for (int j = 0; j < 8; ++j) { a8[j] = 0x38404448 + lane + j; b8[j] = 0x30384044 + 3 * lane + j; }
  f32x16 c0 = {0}, c1 = {0};
  f32x4 d[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);                          // one chain
      if (MODE == 1) c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, 0, 0, 0, 0x7f, 0, 0x7f);  // one chain
      if (MODE == 2) {                                                                                         // the v1 mix: 4 f16 + 2 scaled, one chain
        if ((u % 6) < 4) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        else c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, 0, 0, 0, 0x77, 0, 0x7f);
      }
      if (MODE == 3) d[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d[u & 3], 0, 0, 0);                 // 4 chains
      if (MODE == 4) d[u & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, d[u & 3], 0, 0, 0, 0x7f, 0, 0x7f);
      if (MODE == 5) d[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d[0], 0, 0, 0);                         // 1 chain
      if (MODE == 6) d[0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, d[0], 0, 0, 0, 0x7f, 0, 0x7f);
      if (MODE == 7) { if (u & 1) c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c1, 0, 0, 0, 0x7f, 0, 0x7f);   // 2 chains scaled
                       else c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, 0, 0, 0, 0x7f, 0, 0x7f); }
      if (MODE == 8) d[(u >> 1) & 3] = (u & 1) ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, d[(u >> 1) & 3], 0, 0, 0, 0x7f, 0, 0x7f)
                                               : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8, a8, d[(u >> 1) & 3], 0, 0, 0, 0x77, 0, 0x7f);   // v2 P part: 2 back-to-back on the same acc
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = c0[0] + c0[15] + c1[3];
  for (int i = 0; i < 4; ++i) s += d[i][0] + d[i][3];
  if (lane == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int pipe_cycles) {
  long long* d; float* sink;
  (void)hipMalloc(&d, 8 * 8 * 1024); (void)hipMalloc(&sink, 1024 * 512 * 4);
  const int iters = 64;
  for (int threads : {256, 512})
    for (int blocks : {1, 256}) {
      k<MODE><<<blocks, threads>>>(d, sink, iters);
      (void)hipDeviceSynchronize();
      k<MODE><<<blocks, threads>>>(d, sink, iters);
      long long h[8];
      (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      const double n = (double)iters * 16, w = threads / 256.0;
      printf("%-44s %d thr x %3d blk: %.1f cycles / instr / wave -> %.1f per SIMD slot (nominal %d)\n", name, threads, blocks, h[0] / n, h[0] / n / w, pipe_cycles);
    }
  (void)hipFree(d); (void)hipFree(sink);
}
int main() {
  run<0>("f16 32x32x16, 1 chain", 32);
  run<1>("scaled fp8 32x32x64, 1 chain", 64);
  run<7>("scaled fp8 32x32x64, 2 chains", 64);
  run<2>("4 f16 + 2 scaled 32x32, 1 chain (v1)", 43);
  run<3>("f16 16x16x32, 4 chains", 16);
  run<5>("f16 16x16x32, 1 chain", 16);
  run<4>("scaled fp8 16x16x128, 4 chains", 32);
  run<6>("scaled fp8 16x16x128, 1 chain", 32);
  run<8>("scaled fp8 16x16x128, pairs on 4 accs (v2)", 32);
  return 0;
}
