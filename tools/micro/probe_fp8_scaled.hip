// Probe: operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands, checked with exact small-integer data.
// Packing under test (P1): lane l = (r = l & 31, h = l >> 5) holds row r (A) / column r (B), k = 32 h + b in byte b of its
// 8 operand registers.  Also checks that a uniform E8M0 scale multiplies the product (2^(s-127)) and the f16 32x32x16 lane map.
// Round 3: WHERE a 32-element block's scale comes from.  Result: with per-lane-half scale registers, the products of bytes 0-15 of
// BOTH lane halves take the scale of lanes 0-31 and those of bytes 16-31 the scale of lanes 32-63 -- i.e. in the instruction's own K
// order a lane (r, h) holds k = 16 h + b (b < 16, block 0) and k = 32 + 16 h + (b - 16) (block 1); the P1 packing is a permutation of
// that order applied to both operands alike, which a dot product with UNIFORM scales cannot see.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline unsigned char e4m3(int v) {   // exact for |v| <= 4
  const unsigned char mag[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
  return (unsigned char)(mag[v < 0 ? -v : v] | (v < 0 ? 0x80 : 0));
}

__global__ void probe(const signed char* A /*[32][64]*/, const signed char* B /*[64][32]*/, float* D /*[32][32]*/, float* D2, float* D3,
                      const signed char* A16 /*[32][16]*/, const signed char* B16 /*[16][32]*/) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  i32x8 a, b;
  for (int w = 0; w < 8; ++w) {
    unsigned int x = 0, y = 0;
    for (int j = 0; j < 4; ++j) {
      const int k = 32 * h + 4 * w + j;
      x |= (unsigned int)e4m3(A[r * 64 + k]) << (8 * j);
      y |= (unsigned int)e4m3(B[k * 32 + r]) << (8 * j);
    }
    a[w] = (int)x; b[w] = (int)y;
  }
  f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  f32x16 d2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x77777777, 0, 0x7f7f7f7f);   // A scale 2^-8
  f32x16 d3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, d, 0, 0, 0, 0x7f7f7f7f, 0, 0x7b7b7b7b);   // B scale 2^-4, C = d
  // per-lane-half scales (round 3): lanes 32-63 carry 2^3 on A and 2^-1 on B: under "a lane's scale multiplies that lane's 32 elements
  // (k = 32 h ..)" the k >= 32 products come out x 4
  const int sa4 = h ? (int)0x82828282u : 0x7f7f7f7f, sb4 = h ? 0x7e7e7e7e : 0x7f7f7f7f;
  f32x16 d4 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa4, 0, sb4);
  for (int g = 0; g < 16; ++g) D[2048 + ((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = d4[g];
  for (int g = 0; g < 16; ++g) {
    const int row = (g & 3) + 8 * (g >> 2) + 4 * h;
    D[row * 32 + r] = d[g];
    D2[row * 32 + r] = d2[g];
    D3[row * 32 + r] = d3[g];
  }
  // f16 32x32x16: lane (r, h) holds k = 8 h + j
  f16x8 fa, fb;
  for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(float)A16[r * 16 + 8 * h + j]; fb[j] = (_Float16)(float)B16[(8 * h + j) * 32 + r]; }
  f32x16 e = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, c, 0, 0, 0);
  for (int g = 0; g < 16; ++g) D[1024 + ((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = e[g];
}

int main() {
  signed char hA[32 * 64], hB[64 * 32], hA16[32 * 16], hB16[16 * 32];
  srand(1);
  for (auto& v : hA) v = rand() % 7 - 3;
  for (auto& v : hB) v = rand() % 7 - 3;
  for (auto& v : hA16) v = rand() % 9 - 4;
  for (auto& v : hB16) v = rand() % 9 - 4;
  signed char *A, *B, *A16, *B16; float *D, *D2, *D3;
  hipMalloc(&A, sizeof hA); hipMalloc(&B, sizeof hB); hipMalloc(&A16, sizeof hA16); hipMalloc(&B16, sizeof hB16);
  hipMalloc(&D, 3072 * 4); hipMalloc(&D2, 1024 * 4); hipMalloc(&D3, 1024 * 4);
  hipMemcpy(A, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(B, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(A16, hA16, sizeof hA16, hipMemcpyHostToDevice); hipMemcpy(B16, hB16, sizeof hB16, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(A, B, D, D2, D3, A16, B16);
  float hD[3072], hD2[1024], hD3[1024];
  hipMemcpy(hD, D, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(hD2, D2, sizeof hD2, hipMemcpyDeviceToHost); hipMemcpy(hD3, D3, sizeof hD3, hipMemcpyDeviceToHost);
  int bad = 0, bad2 = 0, bad3 = 0, bad16 = 0, bad4 = 0, bad4b = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      int s = 0, s16 = 0;
      for (int k = 0; k < 64; ++k) s += hA[i * 64 + k] * hB[k * 32 + j];
      for (int k = 0; k < 16; ++k) s16 += hA16[i * 16 + k] * hB16[k * 32 + j];
      bad += hD[i * 32 + j] != (float)s;
      bad2 += hD2[i * 32 + j] != (float)s / 256.f;
      bad3 += hD3[i * 32 + j] != (float)s + (float)s / 16.f;
      bad16 += hD[1024 + i * 32 + j] != (float)s16;
      int s_lo = 0, s_hi = 0, s_a = 0, s_b = 0;
      for (int k = 0; k < 32; ++k) s_lo += hA[i * 64 + k] * hB[k * 32 + j];
      for (int k = 32; k < 64; ++k) s_hi += hA[i * 64 + k] * hB[k * 32 + j];
      for (int k = 0; k < 64; ++k) ((k & 16) ? s_b : s_a) += hA[i * 64 + k] * hB[k * 32 + j];    // alternative: 16-element interleave
      bad4 += hD[2048 + i * 32 + j] != (float)(s_lo + 4 * s_hi);
      bad4b += hD[2048 + i * 32 + j] != (float)(s_a + 4 * s_b);
    }
  printf("fp8 scaled 32x32x64, packing P1: %d wrong of 1024; scale_a 2^-8: %d wrong; C-in + scale_b 2^-4: %d wrong; f16 32x32x16: %d wrong\n", bad, bad2, bad3, bad16);
  printf("per-lane-half scales: %d wrong of 1024 under 'lane (r,h) scale x its k = 32h..32h+31' (alternative 16-interleave: %d wrong)\n", bad4, bad4b);
  printf("sample D[0][0..3] = %g %g %g %g ; D2 = %g %g\n", hD[0], hD[1], hD[2], hD[3], hD2[0], hD2[1]);
  return 0;
}
