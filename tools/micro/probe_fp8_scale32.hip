// Probe: WHERE does v_mfma_scale_f32_32x32x64_f8f6f4 take the E8M0 scale of a (row, 32-element K block) from?
// A = B = all ones (e4m3 1.0), so D[r][c] = 32 * 2^(sa(r,0) + sb(c,0)) + 32 * 2^(sa(r,1) + sb(c,1)) in units of 2^-254.
// Experiments (B scale = 2^0 everywhere unless said):
//   E1  A-scale register: lanes 0-31 -> 2^0, lanes 32-63 -> 2^4 (all four bytes)            per-lane-half semantics: D = 32 + 512 = 544
//   E2  A-scale register: byte0 = 2^0, byte1 = 2^1, byte2 = 2^2, byte3 = 2^3, every lane    opsel 0: which byte is used?  D = 64 * 2^byte
//   E3  as E2 with opsel_a = 1, 2, 3
//   E4  A-scale register: lane L -> 2^(L & 3) for L < 32, 2^0 for L >= 32                    row dependence: D[r][.] = 32 * 2^(r & 3) + 32
//   E5  B-scale register: lanes 0-31 -> 2^0, lanes 32-63 -> 2^4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void probe(float* D) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  i32x8 a, b;
  for (int w = 0; w < 8; ++w) { a[w] = 0x38383838; b[w] = 0x38383838; }
  const f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int one = 0x7f7f7f7f;
  const int s1 = h ? (int)0x83838383u : one;
  const int s2 = (int)0x8281807fu;                        // byte0 = 127, byte1 = 128, byte2 = 129, byte3 = 130
  const int s4 = h ? one : (int)((127u + (r & 3)) * 0x01010101u);
  f32x16 d[8];
  d[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, s1, 0, one);
  d[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, s2, 0, one);
  d[2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, s2, 0, one);
  d[3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 2, s2, 0, one);
  d[4] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 3, s2, 0, one);
  d[5] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, s4, 0, one);
  d[6] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, one, 0, s1);
  d[7] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, s1, 0, s1);
  for (int e = 0; e < 8; ++e)
    for (int g = 0; g < 16; ++g) D[e * 1024 + ((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = d[e][g];
}

int main() {
  float* D;
  hipMalloc(&D, 8 * 1024 * 4);
  probe<<<1, 64>>>(D);
  static float h[8 * 1024];
  hipMemcpy(h, D, sizeof h, hipMemcpyDeviceToHost);
  const char* name[8] = {"E1 A-scale lanes>=32 = 2^4", "E2 bytes 0..3 = 2^0..2^3, opsel 0", "E3 opsel 1", "E3 opsel 2", "E3 opsel 3",
                         "E4 A-scale lane L<32 = 2^(L&3)", "E5 B-scale lanes>=32 = 2^4", "E6 both"};
  for (int e = 0; e < 8; ++e) {
    printf("%-36s D[0..3][0] = %g %g %g %g   D[0][0..3] = %g %g %g %g   D[5][7] = %g\n", name[e], h[e * 1024], h[e * 1024 + 32], h[e * 1024 + 64],
           h[e * 1024 + 96], h[e * 1024], h[e * 1024 + 1], h[e * 1024 + 2], h[e * 1024 + 3], h[e * 1024 + 5 * 32 + 7]);
  }
  return 0;
}
