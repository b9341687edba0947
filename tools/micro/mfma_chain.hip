// Microbenchmark: cycles per v_mfma_f32_32x32x16_bf16 for 1..4 independent accumulation chains, one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NCH>
__global__ __launch_bounds__(512) void chain(long long* out, float* sink, int iters) {
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {3, 1, 4, 1, 5, 9, 2, (short)threadIdx.x};
  f32x16 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][15];
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NCH>
void run(int threads, int iters) {
  long long* d; float* sink;
  hipMalloc(&d, 8 * 8 * 8); hipMalloc(&sink, 8 * 512 * 4);
  chain<NCH><<<1, threads>>>(d, sink, iters);
  hipDeviceSynchronize();
  chain<NCH><<<1, threads>>>(d, sink, iters);
  long long h[8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * 16 * NCH;
  printf("chains=%d threads=%d: %.1f cycles per MFMA per wave (wave0), %.1f (last wave)\n", NCH, threads, h[0] / n, h[threads / 64 - 1] / n);
  hipFree(d); hipFree(sink);
}

int main() {
  for (int threads : {64, 256, 512}) {
    run<1>(threads, 200); run<2>(threads, 200); run<3>(threads, 200); run<4>(threads, 200);
  }
  return 0;
}
