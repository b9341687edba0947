// Probe: v_cvt_scalef32_pk_fp8_f16 (two f16 -> two e4m3 with a scale) -- does it divide or multiply by the scale, does it round to
// nearest even and saturate?  Prints decoded results for a few inputs and scales.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const float* in, int n, float scale, unsigned* out) {
  const int i = threadIdx.x;
  if (i >= n) return;
  f16x2 v = {(_Float16)in[2 * i], (_Float16)in[2 * i + 1]};
  i16x2 old = {0, 0};
  i16x2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, v, scale, false);
  out[i] = (unsigned)(unsigned short)r[0] | ((unsigned)(unsigned short)r[1] << 16);
}

static float e4m3(unsigned b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  if (e == 15 && m == 7) v = NAN;
  return s ? -v : v;
}

int main() {
  const float h[16] = {1.0f, 1.0625f, 1.1875f, 3.0f, 448.0f, 500.0f, -2.5f, 0.001953125f, 100.0f, 0.0f, 17.0f, 18.0f, 65504.0f, -65504.0f, 0.015625f, 0.0078125f};
  float* din; unsigned* dout;
  hipMalloc(&din, sizeof h); hipMalloc(&dout, 8 * 4);
  hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
  const float scales[4] = {1.0f, 2.0f, 128.0f, 0.25f};
  for (float sc : scales) {
    probe<<<1, 64>>>(din, 8, sc, dout);
    unsigned o[8];
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    printf("scale %g:", sc);
    for (int i = 0; i < 8; ++i) printf("  %g->%g %g->%g", h[2 * i], e4m3(o[i] & 0xff), h[2 * i + 1], e4m3((o[i] >> 8) & 0xff));
    printf("   [raw word0 %08x]\n", o[0]);
  }
  return 0;
}
