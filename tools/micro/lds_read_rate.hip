// Microbenchmark: how many bytes per cycle does a CU's LDS deliver to ds_read_b128 in the convolution's access pattern (128-byte rows,
// 16-byte slot XOR-swizzled by the row: a wave reads 32 rows x 2 slots), with 4 / 8 / 16 waves per CU?  And to ds_read_b64 / b32?
// hipcc --offload-arch=gfx950 -O3 -o lds_read_rate tools/micro/lds_read_rate.hip && ./lds_read_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WIDTH>
__global__ __launch_bounds__(1024) void k(long long* out, int reps) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
  for (int i = threadIdx.x; i < 64 * 1024 / 4; i += blockDim.x) reinterpret_cast<int*>(lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int row = (wave * 32 + n) & 255;
  uint32_t base = (uint32_t)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds + row * 128 + (((h) ^ ((row >> 1) & 7)) << 4);
  int4 acc = {0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if constexpr (WIDTH == 16) {
        int4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"((u & 3) * 32 * 128) : "memory");
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc.x ^= v.x;
      } else if constexpr (WIDTH == 8) {
        int2 v;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"((u & 3) * 32 * 128) : "memory");
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc.x ^= v.x;
      } else {
        int v;
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"((u & 3) * 32 * 128) : "memory");
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc.x ^= v;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (acc.x == 0x12345) out[2] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int WIDTH>
void run(long long* d, int waves) {
  const int reps = 2000;
  k<WIDTH><<<256, waves * 64>>>(d, reps);
  k<WIDTH><<<256, waves * 64>>>(d, reps);
  long long h;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  const double bytes = (double)reps * 16 * waves * 64 * WIDTH;
  printf("ds_read_b%-3d %2d waves per CU: %.1f bytes per cycle per CU\n", WIDTH * 8, waves, bytes / (double)h);
}

int main() {
  long long* d;
  hipMalloc(&d, 64);
  for (int w : {4, 8, 16}) { run<16>(d, w); run<8>(d, w); run<4>(d, w); }
  return 0;
}
