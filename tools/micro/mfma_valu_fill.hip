// Microbenchmark: how many vector operations hide behind each MFMA of a DEPENDENT chain (same accumulator)?
//   one wave per SIMD (256 threads per workgroup, 1 workgroup per CU), 48-MFMA chains of v_mfma_f32_32x32x16_f16, N x
//   (v_max_i32 | v_min_i32) between consecutive MFMAs, operands in registers (no LDS).  Prints cycles per MFMA by s_memtime.
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_fill tools/micro/mfma_valu_fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int N, int INDEP>
__global__ __launch_bounds__(256) void k(long long* out, int reps) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
  f32x16 acc, acc2;
  for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
  int v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * (i + 3);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int m = 0; m < 48; ++m) {
      if (INDEP && (m & 1)) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
      for (int i = 0; i < N; ++i) {
        if (i & 1) asm volatile("v_max_i32 %0, %1, %2" : "=v"(v[i & 7]) : "v"(v[(i + 1) & 7]), "v"(v[(i + 3) & 7]));
        else asm volatile("v_min_i32 %0, %1, %2" : "=v"(v[i & 7]) : "v"(v[(i + 2) & 7]), "v"(v[(i + 5) & 7]));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc), "+v"(acc2));
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
  int vs = 0;
  for (int i = 0; i < 8; ++i) vs += v[i];
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; }
  if (s == 12345.678f && vs == 77) out[1] = 1;
}

template <int N, int INDEP>
void run(long long* d, const char* what) {
  const int reps = 200;
  k<N, INDEP><<<256, 256>>>(d, reps);
  k<N, INDEP><<<256, 256>>>(d, reps);
  long long h[2];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("%s N=%d: %.1f cycles per MFMA\n", what, N, (double)h[0] / (reps * 48));
}

int main() {
  long long* d;
  hipMalloc(&d, 64);
  run<0, 0>(d, "dependent chain"); run<2, 0>(d, "dependent chain"); run<4, 0>(d, "dependent chain"); run<5, 0>(d, "dependent chain");
  run<6, 0>(d, "dependent chain"); run<7, 0>(d, "dependent chain"); run<8, 0>(d, "dependent chain"); run<10, 0>(d, "dependent chain");
  run<0, 1>(d, "two chains"); run<4, 1>(d, "two chains"); run<6, 1>(d, "two chains"); run<8, 1>(d, "two chains");
  return 0;
}
