// Does v_mfma_scale_f32_32x32x64_f8f6f4 take its A operand from the ACCUMULATION registers correctly (gfx950)?  Same operands once
// from vector registers, once "a"-constrained; prints the number of differing outputs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const i32x8* a, const i32x8* b, float* o, int sa, int sb) {
  const i32x8 A = a[threadIdx.x], B = b[threadIdx.x];
  f32x16 c0 = {0}, c1 = {0};
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c0) : "v"(A), "v"(B), "v"(sa), "v"(sb));
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c1) : "a"(A), "v"(B), "v"(sa), "v"(sb));
  asm volatile("s_nop 15\n\ts_nop 15");
  for (int i = 0; i < 16; ++i) { o[threadIdx.x * 32 + i] = c0[i]; o[threadIdx.x * 32 + 16 + i] = c1[i]; }
}
int main() {
  i32x8 ha[64], hb[64];
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) { ha[l][j] = 0x38404448 + 0x01010101 * ((l + j) & 7); hb[l][j] = 0x30384044 + 0x01000100 * ((3 * l + j) & 7); }
  i32x8 *da, *db; float* dout; float h[64 * 32];
  (void)hipMalloc(&da, sizeof(ha)); (void)hipMalloc(&db, sizeof(hb)); (void)hipMalloc(&dout, sizeof(h));
  (void)hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  k<<<1, 64>>>(da, db, dout, 0x7f7f7f7f, 0x7f7f7f7f);
  (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) bad += h[l * 32 + i] != h[l * 32 + 16 + i];
  printf("scaled MFMA, A from accumulation registers: %d of 1024 outputs differ from the vector-register form (sample %g vs %g)\n", bad, h[0], h[16]);
  return 0;
}
