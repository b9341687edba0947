// Probe: how precisely does v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 operands) add its 64 products to a LARGE accumulator C?
// A = all ones; B = all ones scaled by 2^-sb (E8M0): every product is 2^-sb, their exact sum is 64 * 2^-sb.
// C = 2^e.  D - C should be 64 * 2^-sb (rounded to f32 at C's magnitude).  A tree that aligns every product to C's exponent and drops
// the bits below some width returns less (or nothing).  Printed: (D - C) / exact for e = 0..30, three product sizes; and the same for
// v_mfma_f32_32x32x16_f16 with 16 products of 2^-sb.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void probe(float* D, int e, int av, int sb1, int sb2) {
  i32x8 a, b;
  for (int w = 0; w < 8; ++w) { a[w] = av; b[w] = av; }
  f16x8 fa, fb;
  for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)1.0f; fb[j] = (_Float16)0.0009765625f; }   // 2^-10
  f32x16 c;
  for (int g = 0; g < 16; ++g) c[g] = ldexpf(1.0f, e);
  const int one = 0x7f7f7f7f;
  const f32x16 d0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, one, 0, one);   // products 1
  const f32x16 d1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, one, 0, sb1);   // products 2^-10
  const f32x16 d2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sb1, 0, sb1);   // products 2^-20
  const f32x16 d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, c, 0, 0, 0);                       // 16 products 2^-10
  const f32x16 d4 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, one, 0, sb2);   // products 2^-17
  if (threadIdx.x == 37) {
    D[5 * e + 0] = d0[3] - c[3]; D[5 * e + 1] = d1[3] - c[3]; D[5 * e + 2] = d2[3] - c[3]; D[5 * e + 3] = d3[3] - c[3]; D[5 * e + 4] = d4[3] - c[3];
  }
}

int main() {
  float* D;
  hipMalloc(&D, 31 * 5 * 4 + 64);
  for (int e = 0; e <= 30; ++e) probe<<<1, 64>>>(D, e, 0x38383838, 0x75757575, 0x6e6e6e6e);
  float h[31 * 5];
  hipMemcpy(h, D, sizeof h, hipMemcpyDeviceToHost);
  printf("C = 2^e | fp8 x64: products 1 (sum 64) | 2^-10 (sum 2^-4) | 2^-20 (sum 2^-14) | 2^-17 (sum 2^-11) | f16 x16 products 2^-10 (sum 2^-6)   [D - C]\n");
  for (int e = 0; e <= 30; ++e)
    printf("e=%2d  %-12g [64]   %-12g [%g]   %-12g [%g]   %-12g [%g]   %-12g [%g]\n", e, h[5 * e], h[5 * e + 1], 0.0625, h[5 * e + 2], ldexp(1.0, -14),
           h[5 * e + 4], ldexp(1.0, -11), h[5 * e + 3], ldexp(1.0, -6));
  return 0;
}
