#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const int* in, unsigned* out) {
  const int sqH = in[threadIdx.x];
  const unsigned c_exp4 = 4u << 23;
  for (int v = 0; v < 4; ++v) {
    unsigned a;
    if (v == 0) asm volatile("v_bfe_u32 %0, %1, 0, 8\n\tv_lshl_add_u32 %0, %0, 23, %2" : "=&v"(a) : "v"(sqH), "v"(c_exp4));
    if (v == 1) asm volatile("v_bfe_u32 %0, %1, 8, 8\n\tv_lshl_add_u32 %0, %0, 23, %2" : "=&v"(a) : "v"(sqH), "v"(c_exp4));
    if (v == 2) asm volatile("v_bfe_u32 %0, %1, 16, 8\n\tv_lshl_add_u32 %0, %0, 23, %2" : "=&v"(a) : "v"(sqH), "v"(c_exp4));
    if (v == 3) asm volatile("v_bfe_u32 %0, %1, 24, 8\n\tv_lshl_add_u32 %0, %0, 23, %2" : "=&v"(a) : "v"(sqH), "v"(c_exp4));
    const unsigned b = (((sqH >> (8 * v)) & 255) + 4) << 23;
    out[(threadIdx.x * 4 + v) * 2] = a;
    out[(threadIdx.x * 4 + v) * 2 + 1] = b;
  }
}
int main() {
  int h[64]; for (int i = 0; i < 64; ++i) h[i] = 0x7b7c7d7e + i * 0x01010101;
  int* d; unsigned* o; hipMalloc(&d, 256); hipMalloc(&o, 64 * 8 * 4);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o);
  unsigned r[512]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) bad += r[2 * i] != r[2 * i + 1];
  printf("bad %d ; sample %08x %08x | %08x %08x\n", bad, r[0], r[1], r[6], r[7]);
}
