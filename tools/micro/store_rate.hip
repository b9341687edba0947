// Microbenchmark: how fast can long-lived workgroups (the dense-volume kernel's grid: 256-query column tiles x key chunks, 8 waves)
// write an HW x HW f32 matrix, by store shape?  The matrix pitch is padded to a multiple of 32 floats so that every shape writes
// whole 128-byte lines (the volume kernel gets the same by its row classes).
//   shape 0: dword per lane, lanes 0-31 -> 128 B of row r, lanes 32-63 -> 128 B of row r + 8   (the accumulator-shaped store today)
//   shape 1: dwordx4 per lane, 64 lanes -> 1 KiB of ONE row                                   (LDS-transposed epilogue)
//   shape 2: dwordx4 per lane, 4 rows x 256 B (16 lanes per row)
//   shape 3: dwordx2 per lane, 2 rows x 256 B
//   shape 4: dword per lane, 64 lanes -> 256 B of ONE row
// nt = 1: non-temporal stores.  Each wave writes 32 rows x 32 columns per "tile" like the kernel (shapes 1-4 re-divide the
// workgroup's 64 rows x 256 columns per stage among the waves), spin = dummy ALU cycles between tiles (0 = pure stores).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int SHAPE, int NT>
__global__ __launch_bounds__(512) void wr(float* __restrict__ v, int HW, size_t pitch, int rows_per_chunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 256;
  const int r_begin = blockIdx.y * rows_per_chunk, r_end = min(r_begin + rows_per_chunk, HW);
  for (int rs = r_begin; rs < r_end; rs += 64) {        // a stage = 64 rows x 256 columns per workgroup = 64 KB
    if constexpr (SHAPE == 0) {
      for (int t = 0; t < 2; ++t)
        for (int e = 0; e < 16; ++e) {
          const int row = rs + 32 * t + (e >> 3) * 16 + (e & 7) % 4 + 4 * ((e & 7) / 4) + 8 * (lane >> 5);
          const int col = c0 + wave * 32 + (lane & 31);
          if (row < r_end && col < HW) {
            float* p = v + (size_t)row * pitch + col;
            if (NT) __builtin_nontemporal_store((float)e, p); else *p = (float)e;
          }
        }
    } else if constexpr (SHAPE == 1) {
      for (int e = 0; e < 8; ++e) {                      // wave w: rows 8 w .. 8 w + 7, each a whole KiB
        const int row = rs + 8 * wave + e, col = c0 + 4 * lane;
        if (row < r_end && col + 3 < HW) {
          f32x4* p = reinterpret_cast<f32x4*>(v + (size_t)row * pitch + col);
          const f32x4 x = {1.f, 2.f, 3.f, (float)e};
          if (NT) __builtin_nontemporal_store(x, p); else *p = x;
        }
      }
    } else if constexpr (SHAPE == 2) {
      for (int e = 0; e < 8; ++e) {                      // wave w: columns 64 w .. 64 w + 63 (256 B), 4 rows per instruction
        const int row = rs + 4 * e + (lane >> 4), col = c0 + 64 * (wave & 3) + 4 * (lane & 15);
        const int row2 = row + 32 * (wave >> 2);
        if (row2 < r_end && col + 3 < HW) {
          f32x4* p = reinterpret_cast<f32x4*>(v + (size_t)row2 * pitch + col);
          const f32x4 x = {1.f, 2.f, 3.f, (float)e};
          if (NT) __builtin_nontemporal_store(x, p); else *p = x;
        }
      }
    } else if constexpr (SHAPE == 3) {
      for (int e = 0; e < 16; ++e) {                     // wave w: columns 64 (w & 3) .., 2 rows x 256 B per instruction
        const int row = rs + 32 * (wave >> 2) + 2 * e + (lane >> 5), col = c0 + 64 * (wave & 3) + 2 * (lane & 31);
        if (row < r_end && col + 1 < HW) {
          f32x2* p = reinterpret_cast<f32x2*>(v + (size_t)row * pitch + col);
          const f32x2 x = {1.f, (float)e};
          if (NT) __builtin_nontemporal_store(x, p); else *p = x;
        }
      }
    } else {
      for (int e = 0; e < 32; ++e) {                     // wave w: columns 64 (w & 3) .., one row x 256 B per instruction
        const int row = rs + 32 * (wave >> 2) + e, col = c0 + 64 * (wave & 3) + lane;
        if (row < r_end && col < HW) {
          float* p = v + (size_t)row * pitch + col;
          if (NT) __builtin_nontemporal_store((float)e, p); else *p = (float)e;
        }
      }
    }
  }
}

template <int SHAPE, int NT>
float run(float* v, int HW, size_t pitch, int chunks) {
  const int rows_per_chunk = ((HW + chunks - 1) / chunks + 63) / 64 * 64;
  dim3 grid((HW + 255) / 256, (HW + rows_per_chunk - 1) / rows_per_chunk);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) wr<SHAPE, NT><<<grid, 512>>>(v, HW, pitch, rows_per_chunk);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) wr<SHAPE, NT><<<grid, 512>>>(v, HW, pitch, rows_per_chunk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main(int argc, char** argv) {
  const int HW = argc > 1 ? atoi(argv[1]) : 25680;
  const size_t pitch = (HW + 31) / 32 * 32;
  float* v;
  if (hipMalloc(&v, pitch * HW * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  const double gb = (double)HW * HW * 4 / 1e9;
  for (int round = 0; round < 2; ++round)
    for (int chunks : {5, 10}) {
      float t[10];
      t[0] = run<0, 1>(v, HW, pitch, chunks); t[1] = run<1, 1>(v, HW, pitch, chunks); t[2] = run<2, 1>(v, HW, pitch, chunks);
      t[3] = run<3, 1>(v, HW, pitch, chunks); t[4] = run<4, 1>(v, HW, pitch, chunks);
      t[5] = run<0, 0>(v, HW, pitch, chunks); t[6] = run<1, 0>(v, HW, pitch, chunks); t[7] = run<2, 0>(v, HW, pitch, chunks);
      t[8] = run<3, 0>(v, HW, pitch, chunks); t[9] = run<4, 0>(v, HW, pitch, chunks);
      printf("HW %d, %d chunks, round %d (ms | TB/s):\n", HW, chunks, round);
      const char* names[5] = {"2x128B dword", "1KiB row x4", "4x256B x4", "2x256B x2", "256B dword"};
      for (int i = 0; i < 10; ++i) printf("   %-14s %s  %.3f ms  %.2f TB/s\n", names[i % 5], i < 5 ? "nt   " : "plain", t[i], gb / t[i]);
    }
  return 0;
}
