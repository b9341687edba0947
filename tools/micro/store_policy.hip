// Microbenchmark: the volume kernel's store shape (dword per lane: lanes 0-31 -> 128 B of row r, lanes 32-63 -> 128 B of row r + 8)
// under every cache-policy combination of global_store_dword (sc0 / sc1 / nt), pure stores of a 25680 x 25680 f32 matrix, padded
// pitch, the kernel's grid (256-column tiles x 5 row chunks, 8 waves).
// hipcc --offload-arch=gfx950 -O3 -o store_policy tools/micro/store_policy.hip && ./store_policy
#include <hip/hip_runtime.h>
#include <cstdio>

template <int POL>
__global__ __launch_bounds__(512) void wr(float* __restrict__ v, int HW, size_t pitch, int rows_per_chunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 256;
  const int r_begin = blockIdx.y * rows_per_chunk, r_end = min(r_begin + rows_per_chunk, HW);
  for (int rs = r_begin; rs < r_end; rs += 64)
    for (int t = 0; t < 2; ++t)
      for (int e = 0; e < 16; ++e) {
        const int row = rs + 32 * t + (e >> 3) * 16 + (e & 7) % 4 + 4 * ((e & 7) / 4) + 8 * (lane >> 5);
        const int col = c0 + wave * 32 + (lane & 31);
        if (row < r_end && col < HW) {
          float* p = v + (size_t)row * pitch + col;
          const float x = (float)e;
          if constexpr (POL == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(x) : "memory");
          else if constexpr (POL == 1) asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(x) : "memory");
          else if constexpr (POL == 2) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(x) : "memory");
          else if constexpr (POL == 3) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
          else if constexpr (POL == 4) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(x) : "memory");
          else if constexpr (POL == 5) asm volatile("global_store_dword %0, %1, off sc0 nt" ::"v"(p), "v"(x) : "memory");
          else if constexpr (POL == 6) asm volatile("global_store_dword %0, %1, off sc1 nt" ::"v"(p), "v"(x) : "memory");
          else asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(x) : "memory");
        }
      }
}

template <int POL>
float run(float* v, int HW, size_t pitch) {
  const int chunks = 5;
  const int rows_per_chunk = ((HW + chunks - 1) / chunks + 63) / 64 * 64;
  dim3 grid((HW + 255) / 256, (HW + rows_per_chunk - 1) / rows_per_chunk);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) wr<POL><<<grid, 512>>>(v, HW, pitch, rows_per_chunk);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) wr<POL><<<grid, 512>>>(v, HW, pitch, rows_per_chunk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  const int HW = 25680;
  const size_t pitch = 25696;
  float* v;
  if (hipMalloc(&v, (size_t)(HW + 64) * pitch * 4) != hipSuccess) return 1;
  const char* names[8] = {"(none)", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
  for (int rnd = 0; rnd < 3; ++rnd) {
    float t[8] = {run<0>(v, HW, pitch), run<1>(v, HW, pitch), run<2>(v, HW, pitch), run<3>(v, HW, pitch),
                  run<4>(v, HW, pitch), run<5>(v, HW, pitch), run<6>(v, HW, pitch), run<7>(v, HW, pitch)};
    if (rnd)
      for (int i = 0; i < 8; ++i) printf("%-12s %.3f ms  %.2f TB/s\n", names[i], t[i], 2.638 / t[i]);
    if (rnd) printf("\n");
  }
  return 0;
}
