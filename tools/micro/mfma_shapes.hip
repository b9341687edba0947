// Does the 16x16 MFMA shape sustain more than the 32x32 one under the board's power cap?  (MI355X_MICROARCH.md, DVFS give-back item 7:
// bare bf16 loops, 16x16x32 ~1.15x the FLOP/s of 32x32x16 at equal cycles per FLOP.)  Every SIMD runs two waves of independent chains
// out of registers on pseudo-random operands, ~1 s per shape: f16 and block-scaled FP6 alone, and the f16f6 convolution's mix
// (per 32-channel chunk and 32 x 32 output tile: 64 cycles of f16 + 32 of FP6) in both shapes.
//      hipcc --offload-arch=gfx950 -O3 mfma_shapes.hip -o mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__device__ unsigned rnd(unsigned x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

template <int MODE>
__global__ __launch_bounds__(512) void k(float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  unsigned s = 0x9e3779b9u * (threadIdx.x + 1) + blockIdx.x;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { s = rnd(s); a[j] = (_Float16)(((int)(s & 1023) - 512) * (1.0f / 256.f)); s = rnd(s); b[j] = (_Float16)(((int)(s & 1023) - 512) * (1.0f / 256.f)); }
  i32x8 a8, b8;
  for (int j = 0; j < 8; ++j) { s = rnd(s); a8[j] = (int)s; s = rnd(s); b8[j] = (int)s; }
  f32x16 c[4] = {{0}, {0}, {0}, {0}};
  f32x4 d[16];
  for (int i = 0; i < 16; ++i) d[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {                  // f16 32x32x16: 12 per iteration
#pragma unroll
      for (int u = 0; u < 12; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[u & 3], 0, 0, 0);
    } else if (MODE == 1) {           // f16 16x16x32: 24 per iteration (the same FLOPs)
#pragma unroll
      for (int u = 0; u < 24; ++u) d[u & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d[u & 15], 0, 0, 0);
    } else if (MODE == 2) {           // fp6 32x32x64: 12
#pragma unroll
      for (int u = 0; u < 12; ++u) c[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[u & 3], 2, 2, 0, 0x7f, 0, 0x7f);
    } else if (MODE == 3) {           // fp6 16x16x128: 12 (the same FLOPs: 16 * 16 * 128 * 2 = 32 * 32 * 64)... half of it: 24
#pragma unroll
      for (int u = 0; u < 24; ++u) d[u & 15] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, d[u & 15], 2, 2, 0, 0x7f, 0, 0x7f);
    } else if (MODE == 4) {           // the convolution's mix, 32x32: per chunk 2 f16 + 1 fp6; 4 chunks
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[u & 3], 0, 0, 0);
        c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[u & 3], 0, 0, 0);
        c[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c[u & 3], 2, 2, 0, 0x7f, 0, 0x7f);
      }
    } else {                          // the same work in the 16x16 shapes: per TWO chunks and four 16 x 16 sub-tiles: 8 f16 16x16x32 + 4 fp6 16x16x128; 2 x
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          d[4 * u + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d[4 * u + t], 0, 0, 0);
          d[4 * u + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d[4 * u + t], 0, 0, 0);
          d[4 * u + t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, d[4 * u + t], 2, 2, 0, 0x7f, 0, 0x7f);
        }
      }
    }
  }
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += c[i][0] + c[i][15];
  for (int i = 0; i < 16; ++i) r += d[i][0] + d[i][3];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, double flop_per_iter, float* sink) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 4, threads = 512;
  const int iters = 3000;
  k<MODE><<<blocks, threads>>>(sink, iters);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    for (int l = 0; l < 8; ++l) k<MODE><<<blocks, threads>>>(sink, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double waves = 8.0 * blocks * (threads / 64) * (double)iters;
    printf("%-44s %8.1f ms  %7.1f TFLOP/s\n", name, ms, waves * flop_per_iter / ms / 1e9);
    fflush(stdout);
  }
}

int main() {
  float* sink;
  (void)hipMalloc(&sink, 1024 * 512 * 4);
  const double F32 = 2.0 * 32 * 32 * 16, F16 = 2.0 * 16 * 16 * 32, P32 = 2.0 * 32 * 32 * 64, P16 = 2.0 * 16 * 16 * 128;
  run<0>("f16 32x32x16", 12 * F32, sink);
  run<1>("f16 16x16x32", 24 * F16, sink);
  run<2>("fp6 32x32x64 (scaled)", 12 * P32, sink);
  run<3>("fp6 16x16x128 (scaled)", 24 * P16, sink);
  run<4>("f16f6 mix, 32x32 (2 f16 + 1 fp6)", 4 * (2 * F32 + P32), sink);
  run<5>("f16f6 mix, 16x16 (8 f16 + 4 fp6 per 2 chunks)", 2 * 4 * (2 * F16 + P16), sink);
  (void)hipFree(sink);
  return 0;
}
