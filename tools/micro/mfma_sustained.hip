// Sustained matrix-pipe rate of the WHOLE chip under its power cap: every SIMD runs two waves of independent MFMA chains out of
// registers (no memory traffic) for about a second per shape; TFLOP/s from HIP events.  The dense peaks the roofline fractions are
// priced against (2.5 PF for the 16-bit shapes, 5 PF for fp8) assume 2.4 GHz; what the board sustains at 1.4 kW is the number that
// bounds a kernel that has nothing but matrix work left.        hipcc --offload-arch=gfx950 -O3 mfma_sustained.hip -o mfma_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 a, b;
  bf16x8 ab, bb;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)(0.01f * (lane + j)); b[j] = (_Float16)(0.02f * (lane - j));
    ab[j] = (__bf16)(0.01f * (lane + j)); bb[j] = (__bf16)(0.02f * (lane - j));
  }
  i32x8 a8, b8;
  for (int j = 0; j < 8; ++j) { a8[j] = 0x38404448 + lane + j; b8[j] = 0x30384044 + 3 * lane + j; }
  f32x16 c[4] = {{0}, {0}, {0}, {0}};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[u & 3], 0, 0, 0);
      if (MODE == 1) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c[u & 3], 0, 0, 0);
      if (MODE == 2) c[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[u & 3], 0, 0, 0, 0x7f, 0, 0x7f);   // fp8 x fp8
      if (MODE == 3) c[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[u & 3], 2, 2, 0, 0x7f, 0, 0x7f);   // fp6 x fp6
      if (MODE == 4) {                                                                                                      // the f16f8 mix: 2 f16 + 1 fp8
        if (u % 3 < 2) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[u & 3], 0, 0, 0);
        else c[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[u & 3], 0, 0, 0, 0x7f, 0, 0x7f);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][15];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double flop_per_instr_avg, float* sink) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 4, threads = 512;                 // 2 waves per SIMD on every CU, 4 rounds of workgroups
  int iters = 2000;
  k<MODE><<<blocks, threads>>>(sink, iters);                  // warm-up + calibration
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    for (int l = 0; l < 8; ++l) k<MODE><<<blocks, threads>>>(sink, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = 8.0 * blocks * (threads / 64) * (double)iters * 16;
    printf("%-34s %8.1f ms  %7.1f TFLOP/s  (%.2f of the 2.4 GHz dense peak of the shape)\n", name, ms, instr * flop_per_instr_avg / ms / 1e9,
           instr * flop_per_instr_avg / ms / 1e9 / (MODE == 2 ? 5000.0 : MODE == 3 ? 10000.0 : MODE == 4 ? 3333.3 : 2500.0));
    fflush(stdout);
  }
}

int main() {
  float* sink;
  (void)hipMalloc(&sink, 1024 * 512 * 4);
  run<0>("f16 32x32x16", 2.0 * 32 * 32 * 16, sink);
  run<1>("bf16 32x32x16", 2.0 * 32 * 32 * 16, sink);
  run<2>("fp8 x fp8 32x32x64 (scaled)", 2.0 * 32 * 32 * 64, sink);
  run<3>("fp6 x fp6 32x32x64 (scaled)", 2.0 * 32 * 32 * 64, sink);
  run<4>("2 f16 + 1 fp8 (the f16f8 mix)", (2 * 2.0 * 32 * 32 * 16 + 2.0 * 32 * 32 * 64) / 3.0, sink);
  (void)hipFree(sink);
  return 0;
}
