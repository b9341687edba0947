// Probe: v_mfma_scale_f32_16x16x128_f8f6f4 with fp6 (e2m3) operands and PER-LANE block scales, checked with exact integers.
// Hypotheses under test:
//   H1  lane l = (r = l & 15, g = l >> 4) holds row r (A) / column r (B), k = 32 g + i, element i in bits [6 i, 6 i + 6) of the
//       192-bit little-endian string of its first 6 operand registers (registers 6, 7 ignored);
//   H2  e2m3 code = sign << 5 | exp << 3 | mant, value = (exp ? (1 + mant / 8) 2^(exp - 1) : mant / 8);
//   H3  the scale byte (E8M0) selected by opsel from lane l's scale register multiplies ALL 32 elements of lane l's operand,
//       i.e. scale block = (row r, k in [32 g, 32 g + 32)); opsel 0..3 picks byte 0..3;
//   H4  mixed formats: A fp8 (cbsz 0), B fp6 (blgp 2).
// Build: hipcc --offload-arch=gfx950 -O2 -o probe_fp6 tools/micro/probe_fp6_scaled.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline unsigned e2m3(int v) {   // exact for |v| <= 7
  const unsigned mag[8] = {0, 8, 16, 20, 24, 26, 28, 30};
  return mag[v < 0 ? -v : v] | (v < 0 ? 32u : 0u);
}
__host__ __device__ inline unsigned char e4m3(int v) {   // exact for |v| <= 4
  const unsigned char mag[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
  return (unsigned char)(mag[v < 0 ? -v : v] | (v < 0 ? 0x80 : 0));
}

__device__ inline i32x8 pack6(const signed char* v32) {   // 32 small integers -> 6 registers (H1)
  unsigned w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 32; ++i) {
    const unsigned c = e2m3(v32[i]);
    const int bit = 6 * i;
    w[bit >> 5] |= c << (bit & 31);
    if ((bit & 31) > 26) w[(bit >> 5) + 1] |= c >> (32 - (bit & 31));
  }
  i32x8 o;
  for (int i = 0; i < 8; ++i) o[i] = (int)w[i];
  o[6] = 0x5a5a5a5a; o[7] = 0x3c3c3c3c;    // must be ignored
  return o;
}

__global__ void probe(const signed char* A /*[16][128]*/, const signed char* B /*[128][16]*/, const signed char* sa /*[16][4]*/,
                      const signed char* sb /*[16][4]*/, float* D /*[4][16][16]*/) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  signed char av[32], bv[32];
  for (int i = 0; i < 32; ++i) { av[i] = A[r * 128 + 32 * g + i]; bv[i] = B[(32 * g + i) * 16 + r]; }
  const i32x8 a6 = pack6(av), b6 = pack6(bv);
  i32x8 a8;
  for (int w = 0; w < 8; ++w) {
    unsigned x = 0;
    for (int j = 0; j < 4; ++j) { int v = av[4 * w + j]; v = v > 4 ? 4 : (v < -4 ? -4 : v); x |= (unsigned)e4m3(v) << (8 * j); }
    a8[w] = (int)x;
  }
  const f32x4 c = {0, 0, 0, 0};
  // (0) unit scales
  f32x4 d0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  // (1) per-lane scales, byte 0 of each: A block (r, g) scaled by 2^sa[r][g], B block (g, r) by 2^sb[r][g]
  const int sA = (127 + sa[r * 4 + g]) & 0xff, sB = (127 + sb[r * 4 + g]) & 0xff;
  f32x4 d1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c, 2, 2, 0, sA | 0x11223300, 0, sB | 0x44556600);
  // (2) opsel: the same scales placed in byte 2 (A) and byte 3 (B)
  f32x4 d2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b6, c, 2, 2, 2, (sA << 16) | 0x11003344, 3, (sB << 24) | 0x00112233);
  // (3) mixed: A fp8 (values clamped to +-4), B fp6, unit scales
  f32x4 d3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b6, c, 0, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * g + i;
    D[0 * 256 + row * 16 + r] = d0[i];
    D[1 * 256 + row * 16 + r] = d1[i];
    D[2 * 256 + row * 16 + r] = d2[i];
    D[3 * 256 + row * 16 + r] = d3[i];
  }
}

int main() {
  signed char hA[16 * 128], hB[128 * 16], hsa[64], hsb[64];
  srand(3);
  for (auto& v : hA) v = rand() % 15 - 7;
  for (auto& v : hB) v = rand() % 15 - 7;
  for (auto& v : hsa) v = rand() % 7 - 3;
  for (auto& v : hsb) v = rand() % 7 - 3;
  signed char *A, *B, *sa, *sb; float* D;
  hipMalloc(&A, sizeof hA); hipMalloc(&B, sizeof hB); hipMalloc(&sa, 64); hipMalloc(&sb, 64); hipMalloc(&D, 4 * 256 * 4);
  hipMemcpy(A, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(B, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(sa, hsa, 64, hipMemcpyHostToDevice); hipMemcpy(sb, hsb, 64, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(A, B, sa, sb, D);
  float hD[4 * 256];
  if (hipMemcpy(hD, D, sizeof hD, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 1; }
  int bad[4] = {0, 0, 0, 0};
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s0 = 0, s1 = 0, s3 = 0;
      for (int k = 0; k < 128; ++k) {
        const int a = hA[i * 128 + k], b = hB[k * 16 + j], gk = k >> 5;
        s0 += a * b;
        s1 += a * b * std::ldexp(1.0, hsa[i * 4 + gk] + hsb[j * 4 + gk]);
        const int a4 = a > 4 ? 4 : (a < -4 ? -4 : a);
        s3 += a4 * b;
      }
      bad[0] += hD[0 * 256 + i * 16 + j] != (float)s0;
      bad[1] += hD[1 * 256 + i * 16 + j] != (float)s1;
      bad[2] += hD[2 * 256 + i * 16 + j] != (float)s1;
      bad[3] += hD[3 * 256 + i * 16 + j] != (float)s3;
    }
  printf("fp6 16x16x128: unit scales %d wrong of 256; per-lane scales (opsel 0) %d wrong; opsel 2/3 %d wrong; fp8 x fp6 %d wrong\n", bad[0], bad[1],
         bad[2], bad[3]);
  printf("sample D0[0][0..3] = %g %g %g %g\n", hD[0], hD[1], hD[2], hD[3]);
  return bad[0] + bad[1] + bad[2] + bad[3] != 0;
}
