// Cycles per 32x32 tile of the f16f8 inner loop (v2 form: 16x16 shapes, F / P register sets, compiler-counted LDS waits), static
// LDS image, no stores, no barriers.  Per wave of a 512-thread workgroup (2 waves per SIMD): matrix-pipe time per tile is 1024
// cycles, so 2048 per tile per wave = pipe saturated.  Variants: LD = 0 operands stay in registers (no LDS reads in the loop),
// 1 = natural LDS reads as in the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
constexpr int LDB = 1056;

template <int LD, int SHAPE>
__global__ __launch_bounds__(512, 2) void k(long long* out, float* sink, int tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * LDB];
  for (int i = threadIdx.x; i < 64 * LDB / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = (i * 2654435761u) & 0x3bff3bffu;
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  f16x8 bq16[2][8];
  i32x8 bq8h[2][2], bq8l[2][2];
  for (int qt = 0; qt < 2; ++qt) {
    const unsigned char* qp = smem + (16 * qt + r) * LDB + 16 * g;
    for (int t = 0; t < 8; ++t) bq16[qt][t] = *reinterpret_cast<const f16x8*>(qp + 64 * t);
    for (int u = 0; u < 2; ++u) {
      bq8h[qt][u] = *reinterpret_cast<const i32x8*>(qp + 512 + 128 * u + 16 * g);
      bq8l[qt][u] = *reinterpret_cast<const i32x8*>(qp + 768 + 128 * u + 16 * g);
    }
  }
  f32x4 acc[2][2];
  f16x8 F[2][4];
  i32x8 Ph[2], Pl[2];
  auto load_F = [&](const unsigned char* ka, int u) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) F[kt][tt] = *reinterpret_cast<const f16x8*>(ka + 16 * kt * LDB + 64 * (4 * u + tt));
  };
  auto load_P = [&](const unsigned char* ka, int u) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const unsigned char* p = ka + 16 * kt * LDB + 512 + 128 * u + 16 * g;
      Ph[kt] = *reinterpret_cast<const i32x8*>(p);
      Pl[kt] = *reinterpret_cast<const i32x8*>(p + 256);
    }
  };
  const unsigned char* ka0 = smem + r * LDB + 16 * g;
  load_F(ka0, 0);
  load_P(ka0, 0);
  float s = 0.f;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int tile = 0; tile < tiles; ++tile) {
    const unsigned char* ka = ka0 + (tile & 1) * 32 * LDB;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) acc[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (LD) load_P(ka, u);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int qt = 0; qt < 2; ++qt)
            acc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(F[kt][tt], bq16[qt][4 * u + tt], acc[kt][qt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (LD) load_F(u == 0 ? ka : ka0 + ((tile + 1) & 1) * 32 * LDB, u == 0 ? 1 : 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          acc[kt][qt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(Ph[kt], bq8l[qt][u], acc[kt][qt], 0, 0, 0, 0x77777777, 0, 0x7f7f7f7f);
          acc[kt][qt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(Pl[kt], bq8h[qt][u], acc[kt][qt], 0, 0, 0, 0x77777777, 0, 0x7f7f7f7f);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    s += acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3];
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  if (threadIdx.x == 0) out[4096 + blockIdx.x] = r1 - r0;
  sink[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int LD, int SHAPE>
void run(const char* name, int blocks) {
  long long* d; float* sink;
  (void)hipMalloc(&d, 8 * 8 * 1024 * 2); (void)hipMalloc(&sink, 1024 * 512 * 4);
  const int tiles = blocks > 1 ? 20000 : 200;
  for (int i = 0; i < (blocks > 1 ? 40 : 1); ++i) k<LD, SHAPE><<<blocks, 512>>>(d, sink, tiles);
  (void)hipDeviceSynchronize();
  k<LD, SHAPE><<<blocks, 512>>>(d, sink, tiles);
  long long h[8], rt;
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipMemcpy(&rt, d + 4096, 8, hipMemcpyDeviceToHost);
  printf("%-40s %3d blocks: cycles per tile per wave:", name, blocks);
  for (int w = 0; w < 8; ++w) printf(" %5.0f", (double)h[w] / tiles);
  printf("   (2048 = pipe saturated); clock %.2f GHz\n", (double)h[4] / (double)rt * 0.1);
  (void)hipFree(d); (void)hipFree(sink);
}
int main() {
  for (int blocks : {1, 256}) {
    run<0, 0>("registers only (no LDS reads in the loop)", blocks);
    run<1, 0>("LDS reads as in the kernel", blocks);
  }
  return 0;
}
