// Probe (round 6): does global_load_lds_dwordx4 honour EXEC?  fgvc_pair_topk_f16f6's ring (pair_topk_v8.hpp) packs 1-KiB bank rows at a
// row stride of 944 bytes in the LDS -- the 928 bytes of a row that carry something + 16 -- so a row's DMA instruction must write 58
// lanes x 16 B and leave the next row's first 80 bytes alone.  Writes rows in DESCENDING order (a full-width write of row m would clobber
// the head of row m + 1, which was written before it) and compares every byte.
// Build: hipcc --offload-arch=gfx950 -O2 -o probe_dma_exec tools/micro/probe_dma_exec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr int ROWS = 32, LDB = 944, ROWB = 2048;

__global__ __launch_bounds__(64) void probe(const unsigned char* src, unsigned char* out, int lanes) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[ROWS * LDB + 1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < ROWS * LDB + 1024; i += 64) smem[i] = 0xEE;
  __syncthreads();
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;
  const unsigned off = 16u * lane;
  for (int m = ROWS - 1; m >= 0; --m) {
    const unsigned char* row = src + (size_t)m * ROWB;
    const unsigned dst = lds0 + (unsigned)(m * LDB);
    if (lane < lanes) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(row), "s"(dst) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  for (int i = lane; i < ROWS * LDB + 1024; i += 64) out[i] = smem[i];
}

int main() {
  std::vector<unsigned char> h(ROWS * ROWB), o(ROWS * LDB + 1024);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned char)((i * 131 + (i >> 11) * 17 + 1) % 251);
  unsigned char *d_src, *d_out;
  hipMalloc(&d_src, h.size());
  hipMalloc(&d_out, o.size());
  hipMemcpy(d_src, h.data(), h.size(), hipMemcpyHostToDevice);
  int bad_total = 0;
  for (int lanes : {58, 64}) {
    probe<<<1, 64>>>(d_src, d_out, lanes);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    hipMemcpy(o.data(), d_out, o.size(), hipMemcpyDeviceToHost);
    int bad = 0, tail_touched = 0;
    for (int m = 0; m < ROWS; ++m)
      for (int b = 0; b < 16 * lanes && m * LDB + b < (int)o.size(); ++b) {
        // with `lanes` = 64 rows overlap: only the bytes no later (lower) row overwrote are expected
        if (lanes == 64 && m > 0 && b < 1024 - LDB) continue;
        if (o[m * LDB + b] != h[(size_t)m * ROWB + b]) ++bad;
      }
    if (lanes == 58)
      for (int m = 0; m < ROWS; ++m)
        for (int b = 928; b < LDB; ++b) tail_touched += o[m * LDB + b] != 0xEE;
    printf("lanes %d: %d wrong bytes, %d pad bytes touched (LDS row stride %d)\n", lanes, bad, tail_touched, LDB);
    if (lanes == 58) bad_total += bad + tail_touched;
  }
  printf(bad_total == 0 ? "EXEC-masked LDS-DMA: OK (58 lanes write 928 bytes, nothing beyond)\n" : "EXEC-masked LDS-DMA: FAILED\n");
  return bad_total != 0;
}
