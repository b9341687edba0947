// Microbenchmark: write a 25680 x 25680 f32 matrix (2.64 GB) as 32-row x 256-column slabs per 512-thread workgroup (the dense
// volume's decomposition), with different store instruction shapes.  `kchunk` slabs per workgroup, walked down the rows.
//   mode 0: dword per lane, lanes 0-31 -> 128 B of row r, lanes 32-63 -> 128 B of row r + 4 (accumulator as it lies): 16 instr / wave
//   mode 1: dwordx2 per lane, 64 lanes -> 512 B of one row: 8 instr / wave (2 per row pair)
//   mode 2: dwordx4 per lane, 64 lanes -> 1 KiB = one whole slab row: 4 instr / wave
// each with nt (non-temporal) and plain stores, row pitch 25680 (rows alternate 0 / 64 B line phase) and 25696 (aligned).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE, bool NT>
__global__ __launch_bounds__(512) void w(float* v, int HW, int pitch, int kchunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 256;
  for (int kb = blockIdx.y * kchunk; kb < (blockIdx.y + 1) * kchunk; ++kb) {
    const int row0 = kb * 32;
    if (row0 >= HW) break;
    if (MODE == 0) {
      const int col = c0 + wave * 32 + (lane & 31);
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < HW && col < HW) {
          float* p = &v[(size_t)row * pitch + col];
          if (NT) __builtin_nontemporal_store((float)r, p); else *p = (float)r;
        }
      }
    } else if (MODE == 1) {
      for (int r = 0; r < 8; ++r) {
        const int row = row0 + wave * 4 + (r >> 1);
        const int col = c0 + (r & 1) * 128 + lane * 2;
        if (row < HW && col + 1 < HW) {
          v2f* p = reinterpret_cast<v2f*>(&v[(size_t)row * pitch + col]);
          v2f x = {1.f, (float)r};
          if (NT) __builtin_nontemporal_store(x, p); else *p = x;
        }
      }
    } else {
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wave * 4 + r;
        const int col = c0 + lane * 4;
        if (row < HW && col + 3 < HW) {
          v4f* p = reinterpret_cast<v4f*>(&v[(size_t)row * pitch + col]);
          v4f x = {1.f, 2.f, 3.f, (float)r};
          if (NT) __builtin_nontemporal_store(x, p); else *p = x;
        }
      }
    }
  }
}
template <int MODE, bool NT>
void run(float* v, int HW, int pitch) {
  const int n_q = (HW + 255) / 256, n_kb = (HW + 31) / 32;
  const int chunks = 1024 / n_q > 0 ? 1024 / n_q : 1;
  const int kchunk = (n_kb + chunks - 1) / chunks;
  dim3 grid(n_q, (n_kb + kchunk - 1) / kchunk);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) w<MODE, NT><<<grid, 512>>>(v, HW, pitch, kchunk);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) w<MODE, NT><<<grid, 512>>>(v, HW, pitch, kchunk);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("mode %d %-5s pitch %d: %.3f ms = %.2f TB/s\n", MODE, NT ? "nt" : "plain", pitch, ms / 5, (double)HW * HW * 4 / (ms / 5) / 1e9);
}
int main() {
  const int HW = 25680;
  float* v;
  (void)hipMalloc(&v, (size_t)(HW + 64) * 25696 * 4);
  for (int pitch : {25680, 25696}) {
    run<0, true>(v, HW, pitch); run<0, false>(v, HW, pitch);
    run<1, true>(v, HW, pitch); run<1, false>(v, HW, pitch);
    run<2, true>(v, HW, pitch); run<2, false>(v, HW, pitch);
  }
  return 0;
}
