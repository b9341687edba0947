// Probe (exact small integers): v_mfma_f32_16x16x32_f16 and v_mfma_scale_f32_16x16x128_f8f6f4 (fp8) operand / result lane maps,
// and v_permlane16_swap.  Assumed maps under test:
//   f16 16x16x32:   lane l = (r = l & 15, g = l >> 4): A[row r][k = 8 g + j], B[k = 8 g + j][col r], j = 0..7
//   fp8 16x16x128:  lane (r, g): A[row r][k = 32 g + b], B[k = 32 g + b][col r], b = 0..31
//   C/D 16x16:      lane (c = l & 15, g = l >> 4), register i: D[row 4 g + i][col c]
//   permlane16_swap(x, y): x' = [x.g0, y.g0, x.g2, y.g2], y' = [x.g1, y.g1, x.g3, y.g3]  (g = 16-lane group)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__host__ __device__ inline unsigned char e4m3(int v) {
  const unsigned char mag[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
  return (unsigned char)(mag[v < 0 ? -v : v] | (v < 0 ? 0x80 : 0));
}
__global__ void probe(const signed char* A /*[16][128]*/, const signed char* B /*[128][16]*/, float* D /*[3][16][16]*/, unsigned* P) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  f32x4 c = {0, 0, 0, 0};
  // f16: uses k = 0..31 of A/B
  f16x8 fa, fb;
  for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(float)A[r * 128 + 8 * g + j]; fb[j] = (_Float16)(float)B[(8 * g + j) * 16 + r]; }
  f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c, 0, 0, 0);
  i32x8 a, b;
  for (int w = 0; w < 8; ++w) {
    unsigned x = 0, y = 0;
    for (int j = 0; j < 4; ++j) {
      const int k = 32 * g + 4 * w + j;
      x |= (unsigned)e4m3(A[r * 128 + k]) << (8 * j);
      y |= (unsigned)e4m3(B[k * 16 + r]) << (8 * j);
    }
    a[w] = (int)x; b[w] = (int)y;
  }
  f32x4 d1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  f32x4 d2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, d1, 0, 0, 0, 0x77777777, 0, 0x7f7f7f7f);
  for (int i = 0; i < 4; ++i) {
    D[(4 * g + i) * 16 + r] = d0[i];
    D[256 + (4 * g + i) * 16 + r] = d1[i];
    D[512 + (4 * g + i) * 16 + r] = d2[i];
  }
  u32x2 s = __builtin_amdgcn_permlane16_swap(1000u + l, 2000u + l, false, false);
  P[l] = s[0]; P[64 + l] = s[1];
}
int main() {
  signed char hA[16 * 128], hB[128 * 16];
  srand(3);
  for (auto& v : hA) v = rand() % 7 - 3;
  for (auto& v : hB) v = rand() % 7 - 3;
  signed char *A, *B; float* D; unsigned* P;
  (void)hipMalloc(&A, sizeof hA); (void)hipMalloc(&B, sizeof hB); (void)hipMalloc(&D, 768 * 4); (void)hipMalloc(&P, 128 * 4);
  (void)hipMemcpy(A, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(B, hB, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(A, B, D, P);
  float hD[768]; unsigned hP[128];
  (void)hipMemcpy(hD, D, sizeof hD, hipMemcpyDeviceToHost); (void)hipMemcpy(hP, P, sizeof hP, hipMemcpyDeviceToHost);
  int bad0 = 0, bad1 = 0, bad2 = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      int s32 = 0, s128 = 0;
      for (int k = 0; k < 32; ++k) s32 += hA[i * 128 + k] * hB[k * 16 + j];
      for (int k = 0; k < 128; ++k) s128 += hA[i * 128 + k] * hB[k * 16 + j];
      bad0 += hD[i * 16 + j] != (float)s32;
      bad1 += hD[256 + i * 16 + j] != (float)s128;
      bad2 += hD[512 + i * 16 + j] != (float)s128 + (float)s128 / 256.f;
    }
  int badp = 0;
  for (int l = 0; l < 64; ++l) {
    const int g = l >> 4, c = l & 15;
    const unsigned ex = (g & 1) ? 2000u + 16 * (g - 1) + c : 1000u + l;          // x' = [x.g0, y.g0, x.g2, y.g2]
    const unsigned ey = (g & 1) ? 2000u + l : 1000u + 16 * (g + 1) + c;          // y' = [x.g1, y.g1, x.g3, y.g3]
    badp += hP[l] != ex;
    badp += hP[64 + l] != ey;
  }
  printf("f16 16x16x32: %d wrong; fp8 16x16x128: %d wrong; with C-in and scale 2^-8: %d wrong; permlane16_swap as assumed: %d wrong\n", bad0, bad1, bad2, badp);
  printf("permlane16_swap x': %u %u %u %u   y': %u %u %u %u (lanes 0,16,32,48)\n", hP[0], hP[16], hP[32], hP[48], hP[64], hP[80], hP[96], hP[112]);
  return 0;
}
