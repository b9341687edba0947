// Microbenchmark: write a 25680 x 25680 f32 matrix (2.64 GB) with the store shapes of the dense-volume epilogue.
//   mode 0: one dword per lane, lanes 0-31 -> 128 B of row r, lanes 32-63 -> 128 B of row r+4 (accumulator register as it lies)
//   mode 1: 16 B per lane, 64 lanes -> 1 KiB of ONE row (what an LDS-transposed epilogue could do)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void w0(float* v, int HW) {           // block: 256 columns x 32 rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 256 + wave * 32 + (lane & 31);
  const int row0 = blockIdx.y * 32;
  if (col >= HW) return;
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < HW) __builtin_nontemporal_store((float)r, &v[(size_t)row * HW + col]);
  }
}
__global__ __launch_bounds__(512) void w1(float* v, int HW) {           // block: 256 columns x 32 rows, a wave writes 4 full 1-KiB rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 256 + lane * 4;
  const int row0 = blockIdx.y * 32 + wave * 4;
  if (col + 3 >= HW) return;
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + r;
    if (row < HW) {
      float4 x = {1.f, 2.f, 3.f, (float)r};
      __builtin_nontemporal_store(x.x, &v[(size_t)row * HW + col]);   // replaced below by a 16-byte store
      *reinterpret_cast<float4*>(&v[(size_t)row * HW + col]) = x;
    }
  }
}
int main() {
  const int HW = 25680;
  float* v;
  hipMalloc(&v, (size_t)HW * HW * 4);
  dim3 grid((HW + 255) / 256, (HW + 31) / 32);
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) { if (mode == 0) w0<<<grid, 512>>>(v, HW); else w1<<<grid, 512>>>(v, HW); }
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) { if (mode == 0) w0<<<grid, 512>>>(v, HW); else w1<<<grid, 512>>>(v, HW); }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.3f ms per 2.64 GB = %.2f TB/s\n", mode, ms / 5, (double)HW * HW * 4 / (ms / 5) / 1e9);
  }
  return 0;
}
