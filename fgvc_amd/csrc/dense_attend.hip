// topk=None branch of masked_attention_efficient (reference local_attention.py:376-383): weights over EVERY unmasked key
// (softmax over all T*HWk keys, or clamp(min=0)^2) instead of the k best, out[p][i] = sum_j w[j][i] * value[p][j].
//
// The affinity slab comes from the dense volume kernels (fgvc_corr_volume_*: vol[key j][query i], one key slot at a time, so
// only ONE HWk x HWq slab exists at any moment); this file streams it once:
//   dense_attend_kernel   one lane per query (64 consecutive queries per wave = one coalesced 256-byte segment of a volume
//                         row), the 4 waves of a workgroup and the `nsplit` workgroups of a column band take interleaved key
//                         rows; per lane an online softmax (running max m, denominator s, P weighted label sums) -- the
//                         rescale runs only when the max moves; label rows are wave-uniform (scalar loads);
//                         with a disc / box mask only the key rows the band can reach are visited;
//                         state[split][i] = {m, s, acc[0..P)} is carried from key slot to key slot in HBM;
//   dense_attend_finish   merges the splits' states and divides.
// HBM-bound by the slab read: HWk*HWq*4 bytes per key slot.
//
// The same streaming pass carries `propagate` (reference affinity_utils.py:33-50: new_img = img @ affinity for a GIVEN dense
// affinity): mode RAW takes the slab's entries as the weights, mode SHIFT takes max(a - thr[i], 0) / max(sum, 1e-12) with thr[i] the
// k-th largest entry of column i (the reference's `topk` branch, :36-44), found by dense_kth_kernel in one more pass over the slab.
#include "common.hpp"

namespace fgvc {

namespace {

constexpr int DA_THREADS = 256;
constexpr int DA_WAVES = DA_THREADS / WAVE;

// merge (m2, s2, a2[]) into (m, s, a[]) -- both relative to their own maxima
template <int PMAX>
__device__ __forceinline__ void merge_state(float& m, float& s, float (&a)[PMAX], float m2, float s2, const float (&a2)[PMAX]) {
  if (m2 == -INFINITY) return;
  if (m == -INFINITY) {
    m = m2;
    s = s2;
#pragma unroll
    for (int p = 0; p < PMAX; ++p) a[p] = a2[p];
    return;
  }
  const float mn = fmaxf(m, m2);
  const float r1 = __expf(m - mn), r2 = __expf(m2 - mn);
  s = s * r1 + s2 * r2;
#pragma unroll
  for (int p = 0; p < PMAX; ++p) a[p] = a[p] * r1 + a2[p] * r2;
  m = mn;
}

template <int PMAX>
__global__ __launch_bounds__(DA_THREADS) void dense_attend_kernel(const float* __restrict__ vol, const float* __restrict__ labels,
                                                                 int Hq, int Wq, int Hk, int Wk, int P, int masked, int r2max,
                                                                 int ry, int rx, int reach_y, int mode, int first,
                                                                 float* __restrict__ state, int nsplit,
                                                                 const float* __restrict__ thr) {
  const bool cosine = mode != 0;                          // every mode but the softmax keeps plain sums (no running maximum)
  __shared__ float sh[DA_WAVES - 1][WAVE][PMAX + 2];
  const int HWq = Hq * Wq, HWk = Hk * Wk;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const int i0 = blockIdx.x * WAVE;
  const int i = i0 + lane;
  const bool live = i < HWq;
  const int qy = live ? i / Wq : 0, qx = live ? i - (i / Wq) * Wq : 0;
  // key rows this band of queries can reach
  int jbeg = 0, jend = HWk;
  if (masked) {
    const int y0 = i0 / Wq, y1 = min(HWq - 1, i0 + WAVE - 1) / Wq;
    jbeg = max(0, y0 - reach_y) * Wk;
    jend = min(Hk, y1 + reach_y + 1) * Wk;
  }
  float m = -INFINITY, s = 0.f, acc[PMAX];
#pragma unroll
  for (int p = 0; p < PMAX; ++p) acc[p] = 0.f;
  const float th = (mode == 3 && live) ? thr[i] : 0.f;
  const int stride = DA_WAVES * nsplit;
  for (int j = jbeg + blockIdx.y * DA_WAVES + wave; j < jend; j += stride) {
    const int ky = j / Wk, kx = j - ky * Wk;          // wave-uniform
    float a = live ? __builtin_nontemporal_load(vol + (size_t)j * HWq + i) : -INFINITY;
    if (masked) {
      const int dy = ky - qy, dx = kx - qx;
      const bool keep = dy * dy + dx * dx <= r2max && abs(dy) <= ry && abs(dx) <= rx;
      a = keep ? a : -INFINITY;
    }
    const float* lab = labels + (size_t)j * P;        // uniform address: scalar loads
    if (cosine) {                                      // 1: clamp(min=0)^2, no normalisation (local_attention.py:379-380)
      float w = a > 0.f ? a * a : 0.f;
      if (mode == 2) w = a == -INFINITY ? 0.f : a;     // 2: the entry itself (propagate, affinity_utils.py:45-49); masked-out keys carry no weight
      if (mode == 3) {                                 // 3: max(a - k-th largest of the column, 0) (:36-44); s = their sum
        w = fmaxf(a - th, 0.f);
        s += w;
      }
#pragma unroll
      for (int p = 0; p < PMAX; ++p)
        if (p < P) acc[p] = fmaf(w, lab[p], acc[p]);
      continue;
    }
    if (a == -INFINITY) continue;
    if (a > m) {                                       // the running max moves: rescale what has been summed
      const float r = __expf(m - a);                   // m = -inf -> 0
      s *= r;
#pragma unroll
      for (int p = 0; p < PMAX; ++p) acc[p] *= r;
      m = a;
    }
    const float w = __expf(a - m);
    s += w;
#pragma unroll
    for (int p = 0; p < PMAX; ++p)
      if (p < P) acc[p] = fmaf(w, lab[p], acc[p]);
  }
  // waves 1..3 hand their state to wave 0
  if (wave > 0) {
    float* d = sh[wave - 1][lane];
    d[0] = m;
    d[1] = s;
#pragma unroll
    for (int p = 0; p < PMAX; ++p) d[2 + p] = acc[p];
  }
  __syncthreads();
  if (wave != 0 || !live) return;
  for (int w = 0; w < DA_WAVES - 1; ++w) {
    const float* d = sh[w][lane];
    float a2[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) a2[p] = d[2 + p];
    if (cosine) {
      s += d[1];
#pragma unroll
      for (int p = 0; p < PMAX; ++p) acc[p] += a2[p];
    } else {
      merge_state<PMAX>(m, s, acc, d[0], d[1], a2);
    }
  }
  float* st = state + ((size_t)blockIdx.y * HWq + i) * (P + 2);
  if (!first) {                                         // earlier key slots of this call sequence
    float a2[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) a2[p] = p < P ? st[2 + p] : 0.f;
    if (cosine) {
      s += st[1];
#pragma unroll
      for (int p = 0; p < PMAX; ++p) acc[p] += a2[p];
    } else {
      merge_state<PMAX>(m, s, acc, st[0], st[1], a2);
    }
  }
  st[0] = m;
  st[1] = s;
#pragma unroll
  for (int p = 0; p < PMAX; ++p)
    if (p < P) st[2 + p] = acc[p];
}

template <int PMAX>
__global__ __launch_bounds__(256) void dense_attend_finish_kernel(const float* __restrict__ state, int nsplit, int HWq, int P,
                                                                int mode, float* __restrict__ out) {
  const bool cosine = mode != 0;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HWq) return;
  float m = -INFINITY, s = 0.f, acc[PMAX];
#pragma unroll
  for (int p = 0; p < PMAX; ++p) acc[p] = 0.f;
  for (int sp = 0; sp < nsplit; ++sp) {
    const float* st = state + ((size_t)sp * HWq + i) * (P + 2);
    float a2[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; ++p) a2[p] = p < P ? st[2 + p] : 0.f;
    if (cosine) {
      s += st[1];
#pragma unroll
      for (int p = 0; p < PMAX; ++p) acc[p] += a2[p];
    } else {
      merge_state<PMAX>(m, s, acc, st[0], st[1], a2);
    }
  }
  // softmax over an all-masked column is NaN in the reference (0/0); same here
  const float inv = mode == 3 ? 1.f / fmaxf(s, 1e-12f) : cosine ? 1.f : 1.f / s;
#pragma unroll
  for (int p = 0; p < PMAX; ++p)
    if (p < P) out[(size_t)i * P + p] = acc[p] * inv;
}

// k-th largest entry of every column of a dense [HWk][HWq] slab (the threshold of propagate's `topk` branch, affinity_utils.py:39):
// one lane per column, a descending list of the K largest values seen (predicated insertion, no branches around the list), the
// waves of a workgroup and the `nsplit` workgroups of a column band take interleaved rows; partial lists [split][i][K] are merged
// by dense_kth_finish_kernel.
template <int K>
__global__ __launch_bounds__(DA_THREADS) void dense_kth_kernel(const float* __restrict__ vol, int HWk, int HWq, float* __restrict__ part,
                                                               int nsplit) {
  __shared__ float sh[DA_WAVES - 1][WAVE][K];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const int i = blockIdx.x * WAVE + lane;
  const bool live = i < HWq;
  float v[K];
#pragma unroll
  for (int r = 0; r < K; ++r) v[r] = -INFINITY;
  auto insert = [&](float a) {
#pragma unroll
    for (int r = 0; r < K; ++r) {
      const bool b = a > v[r];
      const float t = v[r];
      v[r] = b ? a : t;
      a = b ? t : a;
    }
  };
  for (int j = blockIdx.y * DA_WAVES + wave; j < HWk; j += DA_WAVES * nsplit) {
    const float a = live ? __builtin_nontemporal_load(vol + (size_t)j * HWq + i) : -INFINITY;
    if (__builtin_amdgcn_ballot_w64(a > v[K - 1]) != 0ull) insert(a);        // wave-uniform skip: most rows beat nobody's K-th
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < K; ++r) sh[wave - 1][lane][r] = v[r];
  }
  __syncthreads();
  if (wave != 0 || !live) return;
  for (int w = 0; w < DA_WAVES - 1; ++w)
#pragma unroll
    for (int r = 0; r < K; ++r) insert(sh[w][lane][r]);
  float* o = part + ((size_t)blockIdx.y * HWq + i) * K;
#pragma unroll
  for (int r = 0; r < K; ++r) o[r] = v[r];
}

template <int K>
__global__ __launch_bounds__(256) void dense_kth_finish_kernel(const float* __restrict__ part, int nsplit, int HWq, int k,
                                                               float* __restrict__ thr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HWq) return;
  float v[K];
#pragma unroll
  for (int r = 0; r < K; ++r) v[r] = -INFINITY;
  for (int sp = 0; sp < nsplit; ++sp) {
    const float* o = part + ((size_t)sp * HWq + i) * K;
#pragma unroll
    for (int r = 0; r < K; ++r) {
      float a = o[r];
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const bool b = a > v[q];
        const float t = v[q];
        v[q] = b ? a : t;
        a = b ? t : a;
      }
    }
  }
  float out = v[0];
#pragma unroll
  for (int r = 0; r < K; ++r)
    if (r == k - 1) out = v[r];
  thr[i] = out;
}

}  // namespace

int dense_kth_launch(const float* vol, int HWk, int HWq, int k, float* part, int nsplit, float* thr, hipStream_t stream) {
  const dim3 grid((HWq + WAVE - 1) / WAVE, nsplit), fgrid((HWq + 255) / 256);
  if (k <= 16) {
    hipLaunchKernelGGL(dense_kth_kernel<16>, grid, dim3(DA_THREADS), 0, stream, vol, HWk, HWq, part, nsplit);
    hipLaunchKernelGGL(dense_kth_finish_kernel<16>, fgrid, dim3(256), 0, stream, part, nsplit, HWq, k, thr);
  } else {
    hipLaunchKernelGGL(dense_kth_kernel<64>, grid, dim3(DA_THREADS), 0, stream, vol, HWk, HWq, part, nsplit);
    hipLaunchKernelGGL(dense_kth_finish_kernel<64>, fgrid, dim3(256), 0, stream, part, nsplit, HWq, k, thr);
  }
  FGVC_CHECK_LAUNCH("fgvc_dense_kth_f32");
  return FGVC_OK;
}

int dense_attend_splits(int HWq, int HWk) {
  // enough workgroups to cover the chip a few times, never more splits than key rows per wave
  const int bands = (HWq + WAVE - 1) / WAVE;
  int ns = (2048 + bands - 1) / bands;
  ns = max(1, min(ns, max(1, HWk / (DA_WAVES * 8))));
  return min(ns, 64);
}

int dense_attend_launch(const float* vol, const float* labels, int Hq, int Wq, int Hk, int Wk, int P, int masked, int r2max, int ry,
                        int rx, int cosine, int first, float* state, int nsplit, hipStream_t stream, const float* thr) {
  int reach = ry;
  if (r2max < FGVC_NO_LIMIT) reach = min(reach, (int)floor(sqrt((double)r2max)));
  reach = min(reach, Hk);
  const dim3 grid((Hq * Wq + WAVE - 1) / WAVE, nsplit);
#define FGVC_DA(PM)                                                                                                       \
  hipLaunchKernelGGL(dense_attend_kernel<PM>, grid, dim3(DA_THREADS), 0, stream, vol, labels, Hq, Wq, Hk, Wk, P, masked, \
                     r2max, ry, rx, reach, cosine, first, state, nsplit, thr)
  if (P <= 8) FGVC_DA(8);
  else if (P <= 16) FGVC_DA(16);
  else FGVC_DA(32);
#undef FGVC_DA
  FGVC_CHECK_LAUNCH("fgvc_dense_attend_f32");
  return FGVC_OK;
}

int dense_attend_finish_launch(const float* state, int nsplit, int HWq, int P, int cosine, float* out, hipStream_t stream) {
  const dim3 grid((HWq + 255) / 256);
  if (P <= 8) hipLaunchKernelGGL(dense_attend_finish_kernel<8>, grid, dim3(256), 0, stream, state, nsplit, HWq, P, cosine, out);
  else if (P <= 16) hipLaunchKernelGGL(dense_attend_finish_kernel<16>, grid, dim3(256), 0, stream, state, nsplit, HWq, P, cosine, out);
  else hipLaunchKernelGGL(dense_attend_finish_kernel<32>, grid, dim3(256), 0, stream, state, nsplit, HWq, P, cosine, out);
  FGVC_CHECK_LAUNCH("fgvc_dense_attend_finish_f32");
  return FGVC_OK;
}

}  // namespace fgvc
