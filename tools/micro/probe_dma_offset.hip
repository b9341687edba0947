// Probe (round 6): global_load_lds_dwordx4 with an instruction offset -- is the offset added to the LDS address as well as to the
// global address?  (fgvc_pair_topk_f16f6's staging wants two bank rows per scalar base: the second through offset:rowb.)
// Build: hipcc --offload-arch=gfx950 -O2 -o probe_dma_offset tools/micro/probe_dma_offset.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void probe(const unsigned char* src, unsigned char* out) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[8192];
  const int lane = threadIdx.x;
  for (int i = lane; i < 8192; i += 64) smem[i] = 0xEE;
  __syncthreads();
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;
  const unsigned off = 16u * lane;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\ts_waitcnt vmcnt(0)" ::"v"(off), "s"(src), "s"(lds0 + 1024u) : "memory");
  __syncthreads();
  for (int i = lane; i < 8192; i += 64) out[i] = smem[i];
}
int main() {
  std::vector<unsigned char> h(8192), o(8192);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned char)((i * 131 + (i >> 10) * 17 + 1) % 251);
  unsigned char *d_src, *d_out;
  (void)hipMalloc(&d_src, h.size()); (void)hipMalloc(&d_out, o.size());
  (void)hipMemcpy(d_src, h.data(), h.size(), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d_src, d_out);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(o.data(), d_out, o.size(), hipMemcpyDeviceToHost);
  // where did the 1 KiB land, and which source bytes are they?
  int first = -1, n = 0;
  for (int i = 0; i < 8192; ++i) if (o[i] != 0xEE) { if (first < 0) first = i; ++n; }
  int src_off = -1;
  for (int s = 0; s + 1024 <= 8192 && first >= 0; s += 1024) { bool eq = true; for (int b = 0; b < 1024; ++b) eq = eq && o[first + b] == h[s + b]; if (eq) src_off = s; }
  printf("m0 = base + 1024, offset:2048 -> %d bytes written at LDS offset %d, holding source bytes from %d\n", n, first, src_off);
  printf("%s\n", first == 1024 ? "the instruction offset goes to the GLOBAL address only" : first == 3072 ? "the instruction offset goes to BOTH addresses" : "unexpected");
  return 0;
}
