// The consumer stream of fgvc_pair_topk_f16f6 (fgvc_amd/csrc/pair_v7.inc, parts 1 and 3) on its own: static LDS contents, no ring, no
// hand-over, no other roles -- cycles per tile of the generated code itself; 4 waves per workgroup (one per SIMD), with and without
// 8 more idle waves (s_sleep) in the workgroup, 1 and 256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x6 __attribute__((ext_vector_type(6)));
__device__ __forceinline__ i32x6 v7_cat6(const i32x4v& a, const i32x2v& b) { return i32x6{a[0], a[1], a[2], a[3], b[0], b[1]}; }
#define V7_CAT6(A, B) v7_cat6(A, B)
#define V7_SUB8(Q, M) __builtin_shufflevector(Q, Q, 8 * (M), 8 * (M) + 1, 8 * (M) + 2, 8 * (M) + 3, 8 * (M) + 4, 8 * (M) + 5, 8 * (M) + 6, 8 * (M) + 7)

template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void k(float* out, long long* ticks, int tiles) {
  constexpr int LDB = 944, BUFB = 32 * LDB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * BUFB];
  for (int i = threadIdx.x; i < 4 * BUFB / 4; i += NW * 64) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u ^ (i * 2654435761u & 0x03ff03ffu);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, hi = lane >> 5;
  if (wave >= 4) {                                 // idle company
    for (int i = 0; i < tiles * 4; ++i) __builtin_amdgcn_s_sleep(8);
    return;
  }
  f16x32 qh4[4];
  i32x6 q6l[4];
  for (int v = 0; v < 4; ++v) {
    for (int i = 0; i < 32; ++i) qh4[v][i] = (_Float16)(0.01f * ((lane + i + v) % 13));
    q6l[v] = i32x6{0x11111111, 0x22222222, 0x01010101, 0x10101010, 0x12121212, 0x21212121};
    asm volatile("" : "+v"(qh4[v]), "+v"(q6l[v]));
  }
  int sqH = 0x7f7f7f7f, sqL = 0x7b7b7b7b;
  asm volatile("" : "+v"(sqH), "+v"(sqL));
  const uint32_t smem_l = (uint32_t)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;
  const uint32_t a_hand_free = smem_l, a_next_filled = smem_l + 64;
  const uint32_t a_q6h = smem_l + 16 * lane, a_q6t = smem_l + 4096 + 8 * lane;
  uint32_t ka_l = smem_l + wave * BUFB + n * LDB + 16 * hi, ka_n = ka_l;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f16x8 ah[4];
  i32x4v xm, ym, qm;
  i32x2v xt, yt, qt, ksc;
  int peek_free, peek_fill;
  float sum = 0.f;
#define FGVC_V7_PART 1
#include "../../fgvc_amd/csrc/pair_v7.inc"
#undef FGVC_V7_PART
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
#define V7_RELEASE() do { } while (0)
#define V7_LOOKAHEAD() do { ka_n = smem_l + ((wave + t + 1) & 3) * BUFB + n * LDB + 16 * hi; } while (0)
#define FGVC_V7_PART 3
#include "../../fgvc_amd/csrc/pair_v7.inc"
#undef FGVC_V7_PART
#undef V7_RELEASE
#undef V7_LOOKAHEAD
#define FGVC_V7_PART 5
#include "../../fgvc_amd/csrc/pair_v7.inc"
#undef FGVC_V7_PART
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc), "+v"(peek_free), "+v"(peek_fill));
    sum += acc[t & 15];
#define FGVC_V7_PART 4
#include "../../fgvc_amd/csrc/pair_v7.inc"
#undef FGVC_V7_PART
    ka_l = ka_n;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * 256 + (threadIdx.x & 255)] = sum + (float)peek_free + (float)peek_fill;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
  float* out; long long* ticks;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 64);
  const int tiles = 2000;
  for (int nblk : {1, 256}) {
    long long t;
    k<4><<<nblk, 256>>>(out, ticks, 10); hipDeviceSynchronize();
    k<4><<<nblk, 256>>>(out, ticks, tiles); hipDeviceSynchronize();
    hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("%3d workgroup(s),  4 waves: %7.1f cycles per tile (24 MFMAs = 768 of them)\n", nblk, (double)t / tiles);
    k<12><<<nblk, 768>>>(out, ticks, 10); hipDeviceSynchronize();
    k<12><<<nblk, 768>>>(out, ticks, tiles); hipDeviceSynchronize();
    hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("%3d workgroup(s), 12 waves (8 idle): %7.1f cycles per tile\n", nblk, (double)t / tiles);
  }
  return 0;
}
