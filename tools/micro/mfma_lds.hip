// Microbenchmark: cycles per v_mfma_f32_32x32x16_bf16 when the operands come from LDS (ds_read_b128, conflict-free rows),
// software-pipelined one block ahead.  R = fragment reads per block, M = MFMAs per block (each MFMA uses two of the block's
// fragments), 4 accumulator tiles.  Run with 256 threads (one wave per SIMD) and 512 (two).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int R, int M>
__global__ __launch_bounds__(512) void k(long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1040];
  for (int i = threadIdx.x; i < 64 * 1040 / 4; i += blockDim.x) reinterpret_cast<int*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned char* base = lds + (lane & 31) * 1040 + 16 * (lane >> 5);      // padded rows: conflict-free b128
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  bf16x8 f[2][R];
#pragma unroll
  for (int i = 0; i < R; ++i) f[0][i] = *reinterpret_cast<const bf16x8*>(base + 32 * i);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int i = 0; i < R; ++i) f[(u + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(base + 32 * ((i + u + it) & 31));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < M; ++m)
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[u & 1][m % R], f[u & 1][(m + 1) % R], acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][15];
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R, int M>
void run(int threads, int blocks) {
  long long* d; float* sink;
  (void)hipMalloc(&d, 8 * 8 * 1024); (void)hipMalloc(&sink, 1024 * 512 * 4);
  const int iters = 200;
  k<R, M><<<blocks, threads>>>(d, sink, iters);
  (void)hipDeviceSynchronize();
  k<R, M><<<blocks, threads>>>(d, sink, iters);
  long long h[8];
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * 2 * M;
  printf("R=%2d reads per %2d MFMAs, %d threads x %4d blocks: wave0 %.1f  last wave %.1f cycles per MFMA (per-SIMD rate = that / waves per SIMD)\n",
         R, M, threads, blocks, h[0] / n, h[threads / 64 - 1] / n);
  (void)hipFree(d); (void)hipFree(sink);
}

int main() {
  for (int blocks : {1, 256}) {
    for (int threads : {256, 512}) {
      run<2, 1>(threads, blocks);
      run<4, 4>(threads, blocks);
      run<12, 24>(threads, blocks);
      run<8, 24>(threads, blocks);
    }
  }
  return 0;
}
