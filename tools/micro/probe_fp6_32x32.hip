// Probe (round 4): v_mfma_scale_f32_32x32x64_f8f6f4 with FP6 (e2m3) operands -- element packing, the k a lane element stands for, and
// where a 32-element block takes its scale from (fp8 on this shape: bytes 0-15 of BOTH lane halves = block 0, scale from lanes 0-31);
// and v_cvt_scalef32_pk32_fp6_f16 -- does it divide or multiply by its scale operand, how does it round, where do the 32 codes land.
// Build: hipcc --offload-arch=gfx950 -O2 -o probe_fp6_32x32 tools/micro/probe_fp6_32x32.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef unsigned int u32x6 __attribute__((ext_vector_type(6)));

__host__ __device__ inline unsigned e2m3(int v) {   // exact for |v| <= 7
  const unsigned mag[8] = {0, 8, 16, 20, 24, 26, 28, 30};
  return mag[v < 0 ? -v : v] | (v < 0 ? 32u : 0u);
}
__device__ inline i32x8 pack6(const signed char* v32) {
  unsigned w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 32; ++i) {
    const unsigned c = e2m3(v32[i]);
    const int bit = 6 * i;
    w[bit >> 5] |= c << (bit & 31);
    if ((bit & 31) > 26) w[(bit >> 5) + 1] |= c >> (32 - (bit & 31));
  }
  i32x8 o;
  for (int i = 0; i < 8; ++i) o[i] = (int)w[i];
  o[6] = 0x5a5a5a5a; o[7] = 0x3c3c3c3c;
  return o;
}

// packing P1: lane (r, h) element e <-> k = 32 h + e
__global__ void probe(const signed char* A /*[32][64]*/, const signed char* B /*[64][32]*/, float* D /*[3][32][32]*/, const _Float16* cin /*[64][32]*/,
                      unsigned* cout /*[3][64][6]*/) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  signed char av[32], bv[32];
  for (int e = 0; e < 32; ++e) { av[e] = A[r * 64 + 32 * h + e]; bv[e] = B[(32 * h + e) * 32 + r]; }
  const i32x8 a6 = pack6(av), b6 = pack6(bv);
  const f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 d0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a6, b6, c, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  const int sa = h ? (int)0x81818181u : 0x7f7f7f7f;            // lanes 32-63: 2^2 on A
  f32x16 d1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a6, b6, c, 2, 2, 0, sa, 0, 0x7f7f7f7f);
  const int sb = h ? 0x7e7e7e7e : (int)0x80808080u;             // B: lanes 0-31 2^1, lanes 32-63 2^-1; opsel 1 on A with byte 1 = 2^2 for h = 1
  f32x16 d2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a6, b6, c, 2, 2, 1, h ? 0x7f7f817f : 0x7f7f7f7f, 0, sb);
  for (int g = 0; g < 16; ++g) {
    const int row = (g & 3) + 8 * (g >> 2) + 4 * h;
    D[0 * 1024 + row * 32 + r] = d0[g];
    D[1 * 1024 + row * 32 + r] = d1[g];
    D[2 * 1024 + row * 32 + r] = d2[g];
  }
  // cvt: 32 f16 values of this lane, scales 1, 4, 0.25
  f16x32 x;
  for (int e = 0; e < 32; ++e) x[e] = cin[l * 32 + e];
  const float scales[3] = {1.f, 4.f, 0.25f};
  for (int s = 0; s < 3; ++s) {
    const u32x6 o = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, scales[s]);
    for (int w = 0; w < 6; ++w) cout[(s * 64 + l) * 6 + w] = o[w];
  }
}

static double e2m3_val(unsigned c) {
  const int sgn = (c >> 5) & 1, ex = (c >> 3) & 3, m = c & 7;
  const double v = ex ? (1.0 + m / 8.0) * std::ldexp(1.0, ex - 1) : m / 8.0;
  return sgn ? -v : v;
}
static unsigned e2m3_rne(double y) {   // nearest, ties to even, saturating at 7.5
  const double a = std::fabs(y);
  double best = 0; unsigned bc = 0; double bd = 1e9;
  for (unsigned c = 0; c < 32; ++c) {
    const double v = e2m3_val(c), d = std::fabs(v - a);
    if (d < bd || (d == bd && (c & 1) == 0)) { bd = d; best = v; bc = c; }
  }
  (void)best;
  return bc | (y < 0 ? 32u : 0u);
}

int main() {
  signed char hA[32 * 64], hB[64 * 32];
  srand(5);
  for (auto& v : hA) v = rand() % 15 - 7;
  for (auto& v : hB) v = rand() % 15 - 7;
  _Float16 hc[64 * 32];
  for (int i = 0; i < 64 * 32; ++i) {
    const int kind = i % 4;
    double v = (rand() % 2001 - 1000) / 1000.0;                // [-1, 1]
    if (kind == 0) v *= 7.5; else if (kind == 1) v *= 30.0; else if (kind == 2) v *= 1.9; else v = (rand() % 61 - 30) / 16.0;   // incl. exact ties
    hc[i] = (_Float16)v;
  }
  signed char *A, *B; float* D; _Float16* ci; unsigned* co;
  hipMalloc(&A, sizeof hA); hipMalloc(&B, sizeof hB); hipMalloc(&D, 3 * 1024 * 4); hipMalloc(&ci, sizeof hc); hipMalloc(&co, 3 * 64 * 6 * 4);
  hipMemcpy(A, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(B, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(ci, hc, sizeof hc, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(A, B, D, ci, co);
  float hD[3 * 1024]; unsigned ho[3 * 64 * 6];
  if (hipMemcpy(hD, D, sizeof hD, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 1; }
  hipMemcpy(ho, co, sizeof ho, hipMemcpyDeviceToHost);
  int bad0 = 0, bad1_lane = 0, bad1_inter = 0, bad2_inter = 0, bad2_lane = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double s = 0, lane_lo = 0, lane_hi = 0, in_a = 0, in_b = 0, m2i = 0, m2l = 0;
      for (int k = 0; k < 64; ++k) {
        const double p = hA[i * 64 + k] * hB[k * 32 + j];
        s += p;
        const int hh = k >> 5, e = k & 31;                      // P1: lane half hh, element e
        (hh ? lane_hi : lane_lo) += p;                          // "a lane's scale multiplies its own 32 elements"
        ((e & 16) ? in_b : in_a) += p;                          // "elements 0-15 of both halves = block 0 (scale of lanes 0-31)"
        // d2: A scale block1 = 2^2 (byte 1), B scale block0 = 2^1, block1 = 2^-1
        m2i += p * ((e & 16) ? 4.0 * 0.5 : 1.0 * 2.0);
        m2l += p * (hh ? 4.0 * 0.5 : 1.0 * 2.0);
      }
      bad0 += hD[i * 32 + j] != (float)s;
      bad1_lane += hD[1024 + i * 32 + j] != (float)(lane_lo + 4 * lane_hi);
      bad1_inter += hD[1024 + i * 32 + j] != (float)(in_a + 4 * in_b);
      bad2_inter += hD[2048 + i * 32 + j] != (float)m2i;
      bad2_lane += hD[2048 + i * 32 + j] != (float)m2l;
    }
  printf("fp6 32x32x64, packing P1 (k = 32 h + e, element e in bits [6e, 6e+6)): unit scales %d wrong of 1024\n", bad0);
  printf("  A scale 2^2 on lanes 32-63: 'own 32 elements' %d wrong, '16-interleave' %d wrong\n", bad1_lane, bad1_inter);
  printf("  opsel 1 on A + per-half B scales: '16-interleave' %d wrong, 'own' %d wrong\n", bad2_inter, bad2_lane);
  const double scales[3] = {1.0, 4.0, 0.25};
  for (int s = 0; s < 3; ++s) {
    int bad_div = 0, bad_mul = 0;
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 32; ++e) {
        const int bit = 6 * e;
        unsigned long long w = ho[(s * 64 + l) * 6 + (bit >> 5)];
        if ((bit >> 5) + 1 < 6) w |= (unsigned long long)ho[(s * 64 + l) * 6 + (bit >> 5) + 1] << 32;
        const unsigned c = (unsigned)(w >> (bit & 31)) & 63u;
        const double x = (double)hc[l * 32 + e];
        unsigned wd = e2m3_rne(x / scales[s]), wm = e2m3_rne(x * scales[s]);
        if ((wd & 31) == 0) wd &= 31 | (c & 32);                // the sign of a zero result is free
        if ((wm & 31) == 0) wm &= 31 | (c & 32);
        bad_div += c != wd;
        bad_mul += c != wm;
        if (s == 1 && l == 0 && e < 8) printf("    x = %9.5f -> code %2u = %6.3f (x/4 = %8.5f, x*4 = %8.4f)\n", x, c, e2m3_val(c), x / 4, x * 4);
      }
    printf("  v_cvt_scalef32_pk32_fp6_f16, scale %g: element e in bits [6e, 6e+6), RNE + saturation: 'x / scale' %d wrong of 2048, 'x * scale' %d wrong\n",
           scales[s], bad_div, bad_mul);
  }
  return 0;
}
