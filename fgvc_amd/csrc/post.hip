// Bandwidth-bound companions of the correlation kernels: L2-normalise + layout change,
// per-slot list merge (+ temperature + softmax), label propagation, initial Gaussian labels and
// the fused bilinear-upsample / top-5 soft-argmax read-out.
#include "common.hpp"

namespace fgvc {

// ------------------------------------------------------------------------------------------
// F.normalize(dim=1) + NCHW -> channels-last.   in [n][C][HW] -> out [n][HW][C]
// One workgroup = 32 pixels x all channels through an LDS tile (row stride 33: conflict-free both ways).
// Reads are 128-B row segments per (channel, 32 pixels); writes are fully contiguous (32*C floats).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normalize_chw_to_hwc_kernel(const float* __restrict__ in,
                                                                     float* __restrict__ out, int C, int HW,
                                                                     int normalize, int Cout) {
  extern __shared__ float tile[];  // [C][33] + [8][32] partial sums + [32] inverse norms
  float* part = tile + (size_t)C * 33;
  float* inv = part + 8 * 32;
  const int tid = threadIdx.x;
  const int px = tid & 31, grp = tid >> 5;  // 8 channel groups
  const int p0 = blockIdx.x * 32;
  const size_t frame = blockIdx.y;
  const float* src = in + frame * (size_t)C * HW;
  float ss = 0.f;
  const int p = p0 + px;
  for (int c = grp; c < C; c += 8) {
    const float v = (p < HW) ? src[(size_t)c * HW + p] : 0.f;
    tile[c * 33 + px] = v;
    ss = fmaf(v, v, ss);
  }
  part[grp * 32 + px] = ss;
  __syncthreads();
  if (tid < 32) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += part[g * 32 + tid];
    // F.normalize: x / max(||x||, eps), eps = 1e-12
    inv[tid] = normalize ? fmaxf(sqrtf(s), 1e-12f) : 1.0f;
  }
  __syncthreads();
  // output rows are Cout >= C floats long; channels C..Cout-1 are written as zeros (they add
  // nothing to any dot product) so the MFMA kernels can run on their supported channel counts
  float* dst = out + (frame * HW + p0) * (size_t)Cout;
  const int npx = imin(32, HW - p0);
  for (int i = tid; i < npx * Cout; i += 256) {
    const int pp = i / Cout, c = i - pp * Cout;
    dst[i] = c < C ? tile[c * 33 + pp] / inv[pp] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------
// merge T per-pair top-k lists into the per-frame top-k, apply temperature, compute weights.
// One thread per (output frame, query pixel); lists are 4*k-byte contiguous rows.
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void merge_topk_kernel(const int32_t* __restrict__ pair_idx,
                                                          const float* __restrict__ pair_score,
                                                          const int32_t* __restrict__ slot_pair, int T, int HWq,
                                                          int HWk, int kout, float temperature, int weight_mode,
                                                          int32_t* __restrict__ idx_out,
                                                          float* __restrict__ logit_out,
                                                          float* __restrict__ weight_out) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  const int f = blockIdx.y;
  if (q >= HWq) return;
  TopK<K> top;
  top.init();
  for (int t = 0; t < T; ++t) {
    const int pid = slot_pair[f * T + t];
    if (pid < 0) continue;
    const size_t o = ((size_t)pid * HWq + q) * kout;
    if constexpr ((K & 1) == 0) {
      if (kout == K) {
        // full lists of an even length: 8-byte loads of the whole list (a lane's list is K * 4 contiguous bytes, 8-byte aligned; read
        // one dword at a time every load instruction of a wave touched 64 dwords K * 4 bytes apart), then the same early-out walk
        int2 ii[K / 2];
        float2 ss[K / 2];
#pragma unroll
        for (int j = 0; j < K / 2; ++j) {
          ii[j] = reinterpret_cast<const int2*>(pair_idx + o)[j];
          ss[j] = reinterpret_cast<const float2*>(pair_score + o)[j];
        }
        bool more = true;
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const int id = (j & 1) ? ii[j / 2].y : ii[j / 2].x;
          const float s = (j & 1) ? ss[j / 2].y : ss[j / 2].x;
          const int gid = t * HWk + id;
          more = more && id >= 0 && top.accepts(s, gid);
          if (more) top.insert(s, gid);
        }
        continue;
      }
    }
    for (int j = 0; j < kout; ++j) {
      const int id = pair_idx[o + j];
      if (id < 0) break;  // lists are sorted; -1 marks the empty tail
      const float s = pair_score[o + j];
      const int gid = t * HWk + id;  // the reference's flat key index (local_attention.py:312)
      if (!top.accepts(s, gid)) break;  // sorted input: nothing further down can enter either
      top.insert(s, gid);
    }
  }
  float lg[K];
#pragma unroll
  for (int j = 0; j < K; ++j) lg[j] = top.v[j] / temperature;  // einsum(...) / temperature, :321-323
  float w[K];
  if (weight_mode == FGVC_WEIGHT_SOFTMAX) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      w[j] = (j < kout) ? expf(lg[j] - lg[0]) : 0.f;
      sum += w[j];
    }
#pragma unroll
    for (int j = 0; j < K; ++j) w[j] = w[j] / sum;
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float c = fmaxf(lg[j], 0.f);
      w[j] = c * c;
    }
  }
  const size_t o = ((size_t)f * HWq + q) * kout;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    if (j < kout) {
      const bool e = top.ix[j] == IDX_EMPTY;
      idx_out[o + j] = e ? -1 : top.ix[j];
      logit_out[o + j] = lg[j];
      weight_out[o + j] = w[j];
    }
  }
}

template <int K>
static void launch_merge(const int32_t* pi, const float* ps, const int32_t* sp, int n_out, int T, int HWq, int HWk,
                         int kout, float temp, int mode, int32_t* io, float* lo, float* wo, hipStream_t s) {
  dim3 grid(cdiv(HWq, 256), n_out);
  merge_topk_kernel<K><<<grid, 256, 0, s>>>(pi, ps, sp, T, HWq, HWk, kout, temp, mode, io, lo, wo);
}

int merge_topk_launch(const int32_t* pi, const float* ps, const int32_t* sp, int n_out, int T, int HWq, int HWk,
                      int topk, float temp, int mode, int32_t* io, float* lo, float* wo, hipStream_t s) {
  if (topk <= 1) launch_merge<1>(pi, ps, sp, n_out, T, HWq, HWk, topk, temp, mode, io, lo, wo, s);
  else if (topk <= 5) launch_merge<5>(pi, ps, sp, n_out, T, HWq, HWk, topk, temp, mode, io, lo, wo, s);
  else if (topk <= 10) launch_merge<10>(pi, ps, sp, n_out, T, HWq, HWk, topk, temp, mode, io, lo, wo, s);
  else launch_merge<16>(pi, ps, sp, n_out, T, HWq, HWk, topk, temp, mode, io, lo, wo, s);
  FGVC_CHECK_LAUNCH("fgvc_merge_topk_f32");
  return FGVC_OK;
}

// ------------------------------------------------------------------------------------------
// label propagation: out[q][p] = sum_r w[q][r] * labels[slot_frame[slot]][pixel][p]
// thread = (query, label) with the label fastest: a query's P labels are one contiguous gather.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void propagate_kernel(const float* __restrict__ labels,
                                                         const int32_t* __restrict__ slot_frame, int T,
                                                         const int32_t* __restrict__ idx,
                                                         const float* __restrict__ weight, int Hq, int Wq, int Hk,
                                                         int Wk, int P, int topk, int window_L,
                                                         float* __restrict__ out) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const int HWq = Hq * Wq, HWk = Hk * Wk;
  if (gid >= (long long)HWq * P) return;
  const int q = (int)(gid / P), pl = (int)(gid - (long long)q * P);
  float acc = 0.f;
  const int span = window_L > 0 ? window_L * window_L : HWk;
  for (int r = 0; r < topk; ++r) {
    const int id = idx[(size_t)q * topk + r];
    if (id < 0) continue;
    const float w = weight[(size_t)q * topk + r];
    const int slot = id / span;
    int pix = id - slot * span;
    if (window_L > 0) {
      const int R = window_L >> 1;
      const int ky = q / Wq + pix / window_L - R, kx = q % Wq + pix % window_L - R;
      if (ky < 0 || ky >= Hk || kx < 0 || kx >= Wk) continue;  // zero padding (F.unfold)
      pix = ky * Wk + kx;
    }
    const float v = labels[((size_t)slot_frame[slot] * HWk + pix) * P + pl];
    acc = fmaf(w, v, acc);
  }
  out[gid] = acc;
}

// ------------------------------------------------------------------------------------------
// initial labels on the feature grid (vanilla_tracker.py:204-221)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gaussian_labels_kernel(const float* __restrict__ points, int P, int Hf,
                                                               int Wf, int stride, float two_sigma2,
                                                               float* __restrict__ out) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)Hf * Wf * P) return;
  const int pix = (int)(gid / P), pl = (int)(gid - (long long)pix * P);
  const float x = (float)((pix % Wf) * stride), y = (float)((pix / Wf) * stride);
  const float dx = x - points[2 * pl], dy = y - points[2 * pl + 1];
  out[gid] = expf(-(dx * dx + dy * dy) / two_sigma2);
}

// ------------------------------------------------------------------------------------------
// read-out: bilinear upsample (align_corners=False) + top-5 soft-argmax, one workgroup per (frame, label)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void src_index(int d, float scale, int in_size, int& i0, int& i1, float& l1) {
  // PyTorch area_pixel_compute_source_index(align_corners=False): max(scale*(d+0.5)-0.5, 0)
  float s = scale * ((float)d + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

// First-frame label at an image pixel (vanilla_tracker.py:204-221), same bits wherever it is evaluated.
__device__ __forceinline__ float gauss_value(int x, int y, float cx, float cy, float two_sigma2) {
  const float dx = (float)x - cx, dy = (float)y - cy;
  return expf(-fmaf(dx, dx, dy * dy) / two_sigma2);
}

// One pixel of the upsampled map.  The contraction order is spelled out so that every kernel evaluating a pixel gets the
// same bits (the pruned read-out and the full scan must agree on values AND ties).
__device__ __forceinline__ float fine_value(const float* __restrict__ lab, int Hf, int Wf, int P, float sy, float sx,
                                            int y, int x) {
  int y0, y1, x0, x1;
  float ly, lx;
  src_index(y, sy, Hf, y0, y1, ly);
  src_index(x, sx, Wf, x0, x1, lx);
  const float v00 = lab[((size_t)y0 * Wf + x0) * P], v01 = lab[((size_t)y0 * Wf + x1) * P];
  const float v10 = lab[((size_t)y1 * Wf + x0) * P], v11 = lab[((size_t)y1 * Wf + x1) * P];
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float top = fmaf(lx, v01, hx * v00), bot = fmaf(lx, v11, hx * v10);
  return fmaf(ly, bot, hy * top);
}

// Stage 1: one workgroup per (label, frame, row band): top-5 candidates + partial sum of its band.
// Stage 2: one wave per (label, frame) merges the bands and writes the coordinates.
// (A single workgroup per map leaves half of the 256 CUs idle at 8 frames x 16 labels and runs 0.9 ms.)
constexpr int RO_K = 5;

__global__ __launch_bounds__(256) void softargmax_band_kernel(const float* __restrict__ labels, int Hf, int Wf,
                                                               int P, int h, int w, int nbands,
                                                               const float* __restrict__ gauss_points,
                                                               float two_sigma2, float* __restrict__ part_v,
                                                               int* __restrict__ part_i, float* __restrict__ part_sum,
                                                               const int* __restrict__ need_scan) {
  constexpr int K = RO_K;
  __shared__ float sv[256 * K];
  __shared__ int si[256 * K];
  __shared__ float ssum[256];
  const int tid = threadIdx.x;
  const int pl = blockIdx.x, f = blockIdx.y, band = blockIdx.z;
  if (need_scan != nullptr && need_scan[(size_t)f * P + pl] == 0) return;   // the pruned read-out finished this map
  const bool analytic = (gauss_points != nullptr) && f == 0;
  const float* lab = labels + (size_t)f * Hf * Wf * P + pl;
  const float sy = (float)Hf / (float)h, sx = (float)Wf / (float)w;
  float cx = 0.f, cy = 0.f;
  if (analytic) {
    cx = gauss_points[2 * pl];
    cy = gauss_points[2 * pl + 1];
  }
  const int rows = cdiv(h, nbands);
  const int y_lo = band * rows, y_hi = imin(h, y_lo + rows);
  TopKHi<K> top;
  top.init();
  float sum = 0.f;
  for (int i = y_lo * w + tid; i < y_hi * w; i += 256) {
    const int y = i / w, x = i - y * w;
    float v;
    if (analytic) {
      v = gauss_value(x, y, cx, cy, two_sigma2);
    } else {
      v = fine_value(lab, Hf, Wf, P, sy, sx, y, x);
    }
    sum += v;
    if (top.accepts(v, i)) top.insert(v, i);
  }
#pragma unroll
  for (int j = 0; j < K; ++j) {
    sv[tid * K + j] = top.v[j];
    si[tid * K + j] = top.ix[j];
  }
  ssum[tid] = sum;
  __syncthreads();
  for (int stride = 128; stride >= 1; stride >>= 1) {
    if (tid < stride) {
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float v = sv[(tid + stride) * K + j];
        const int id = si[(tid + stride) * K + j];
        if (id >= 0 && top.accepts(v, id)) top.insert(v, id);
      }
#pragma unroll
      for (int j = 0; j < K; ++j) {
        sv[tid * K + j] = top.v[j];
        si[tid * K + j] = top.ix[j];
      }
      ssum[tid] += ssum[tid + stride];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const size_t o = (((size_t)f * P + pl) * nbands + band);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      part_v[o * K + j] = top.v[j];
      part_i[o * K + j] = top.ix[j];
    }
    part_sum[o] = ssum[0];
  }
}

// ------------------------------------------------------------------------------------------
// Pruned read-out (exact).  A bilinear sample is a convex combination of its four coarse corners, so no upsampled pixel of
// interval cell (i, j) exceeds B(i, j) = max of those corners.  One workgroup per (frame, label) map:
//   0. scan the coarse map: maximum M with its cell, minimum (negative labels -> no shortcut for the sum test -> full scan);
//   1. evaluate the upsampled pixels of the 3 x 3 cells around the maximum; tau = their 5th largest value (a lower bound on
//      the 5th largest of the whole map);
//   2. every cell with B >= tau (a few ulps of slack for the rounding of the interpolation) goes on a work list;
//   3. evaluate the pixels of the listed cells, keep those >= tau;
//   4. rank the survivors (value desc, higher index first among equals): ranks 0-4 are the map's top 5, identical to a full
//      scan because every pixel >= tau lies in a listed cell.
// For non-negative maps "sum == 0" <=> M == 0.  Whatever does not fit (negative labels, flat maps whose lists overflow,
// fewer than 5 evaluated pixels) sets need_scan[map] and is redone by softargmax_band_kernel: slower, never different.
// The first frame's analytic Gaussians decrease with distance from the centre: their top 5 are inside a 13 x 13 window.
// ------------------------------------------------------------------------------------------
constexpr int RO_CELLS = 1024, RO_CAND = 2048, RO_NEIGH = 1024, RO_BLOCK = 1024;

// first fine coordinate whose source interval index is >= cell (lower bound over d in [0, n])
__device__ __forceinline__ int first_fine_of_cell(int cell, float scale, int in_size, int n) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    int i0, i1;
    float l1;
    src_index(mid, scale, in_size, i0, i1, l1);
    if (i0 >= cell) hi = mid; else lo = mid + 1;
  }
  return lo;
}

__global__ __launch_bounds__(RO_BLOCK) void softargmax_pruned_kernel(const float* __restrict__ labels, int Hf, int Wf, int P,
                                                                 int h, int w, int nbands,
                                                                 const float* __restrict__ gauss_points, float two_sigma2,
                                                                 float* __restrict__ part_v, int* __restrict__ part_i,
                                                                 float* __restrict__ part_sum, int* __restrict__ need_scan) {
  constexpr int K = RO_K;
  __shared__ float red_v[RO_BLOCK];
  __shared__ int red_i[RO_BLOCK];
  __shared__ float red_m[RO_BLOCK];
  __shared__ float neigh[RO_NEIGH];
  __shared__ int cells[RO_CELLS];
  __shared__ float cand_v[RO_CAND];
  __shared__ int cand_i[RO_CAND];
  __shared__ int n_cells, n_cand;
  __shared__ float tau_s;
  const int tid = threadIdx.x;
  const int pl = blockIdx.x, f = blockIdx.y;
  const size_t map = (size_t)f * P + pl;
  const size_t o = map * nbands;
  const float* lab = labels + (size_t)f * Hf * Wf * P + pl;
  const float sy = (float)Hf / (float)h, sx = (float)Wf / (float)w;
  const bool analytic = (gauss_points != nullptr) && f == 0;
  if (tid == 0) { n_cells = 0; n_cand = 0; tau_s = -INFINITY; }
  for (int j = tid; j < nbands * K; j += RO_BLOCK) {     // bands other than 0 stay empty unless the full scan redoes the map
    part_v[o * K + j] = -INFINITY;
    part_i[o * K + j] = -1;
  }
  for (int j = tid; j < nbands; j += RO_BLOCK) part_sum[o + j] = 0.f;
  __syncthreads();
  float map_max;
  if (analytic) {
    const float cx = gauss_points[2 * pl], cy = gauss_points[2 * pl + 1];
    const int xc = (int)fminf(fmaxf(rintf(cx), 0.f), (float)(w - 1)), yc = (int)fminf(fmaxf(rintf(cy), 0.f), (float)(h - 1));
    float mx = -INFINITY;
    if (tid < 169) {
      const int y = yc + tid / 13 - 6, x = xc + tid % 13 - 6;
      if (y >= 0 && y < h && x >= 0 && x < w) {
        const float v = gauss_value(x, y, cx, cy, two_sigma2);
        const int slot = atomicAdd(&n_cand, 1);
        cand_v[slot] = v;
        cand_i[slot] = y * w + x;
        mx = v;
      }
    }
    red_v[tid] = mx;
    __syncthreads();
    for (int st = RO_BLOCK / 2; st >= 1; st >>= 1) {
      if (tid < st) red_v[tid] = fmaxf(red_v[tid], red_v[tid + st]);
      __syncthreads();
    }
    map_max = red_v[0];
    if (!(map_max >= 0.f) || n_cand < K) {            // NaN centre / degenerate window: let the full scan decide
      if (tid == 0) need_scan[map] = 1;
      return;
    }
  } else {
    // 0. coarse scan
    float mx = -INFINITY, mn = INFINITY;
    int am = 0;
    const int n_coarse = Hf * Wf;
    for (int c0 = tid; c0 < n_coarse; c0 += RO_BLOCK * 4) {       // four loads in flight per thread: the scan is latency-bound
      float v4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * RO_BLOCK;
        v4[u] = lab[(size_t)imin(c, n_coarse - 1) * P];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * RO_BLOCK;
        const float v = v4[u];
        if (c < n_coarse) {
          if (v > mx) { mx = v; am = c; }
          mn = (v < mn || v != v) ? v : mn;           // a NaN sticks (then the comparison below fails)
        }
      }
    }
    red_v[tid] = mx; red_i[tid] = am; red_m[tid] = mn;
    __syncthreads();
    for (int st = RO_BLOCK / 2; st >= 1; st >>= 1) {
      if (tid < st) {
        if (red_v[tid + st] > red_v[tid]) { red_v[tid] = red_v[tid + st]; red_i[tid] = red_i[tid + st]; }
        const float a = red_m[tid], b = red_m[tid + st];
        red_m[tid] = (a != a) ? a : ((b < a || b != b) ? b : a);
      }
      __syncthreads();
    }
    map_max = red_v[0];
    const int cell0 = red_i[0];
    if (!(red_m[0] >= 0.f)) {                          // negative or NaN labels: the sum test needs the full scan
      if (tid == 0) need_scan[map] = 1;
      return;
    }
    if (map_max > 0.f) {
      // 1. the neighbourhood of the maximum gives tau
      const int ci = cell0 / Wf, cj = cell0 - ci * Wf;
      int y_lo = first_fine_of_cell(imax(ci - 1, 0), sy, Hf, h), y_hi = first_fine_of_cell(imin(ci + 2, Hf), sy, Hf, h);
      int x_lo = first_fine_of_cell(imax(cj - 1, 0), sx, Wf, w), x_hi = first_fine_of_cell(imin(cj + 2, Wf), sx, Wf, w);
      if (ci + 2 >= Hf) y_hi = h;
      if (cj + 2 >= Wf) x_hi = w;
      y_hi = imin(y_hi, y_lo + 32);                     // any subset of pixels gives a valid (lower) tau
      x_hi = imin(x_hi, x_lo + 32);
      const int nx = x_hi - x_lo, nn = nx * (y_hi - y_lo);
      for (int t = tid; t < nn; t += RO_BLOCK) neigh[t] = fine_value(lab, Hf, Wf, P, sy, sx, y_lo + t / nx, x_lo + t % nx);
      __syncthreads();
      for (int t = tid; t < nn; t += RO_BLOCK) {
        const float v = neigh[t];
        int rank = 0;
        for (int u = 0; u < nn; ++u) {
          const float vu = neigh[u];
          rank += (vu > v || (vu == v && u < t)) ? 1 : 0;
        }
        if (rank == K - 1) tau_s = v;
      }
      __syncthreads();
      const float tau = tau_s;
      if (!(tau > -INFINITY)) {                         // fewer than 5 pixels evaluated
        if (tid == 0) need_scan[map] = 1;
        return;
      }
      // 2. cells that can hold a pixel >= tau
      for (int c0 = tid; c0 < n_coarse; c0 += RO_BLOCK * 4) {
        float b4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = imin(c0 + u * RO_BLOCK, n_coarse - 1);
          const int i = c / Wf, j = c - i * Wf;
          const int i1 = i + (i < Hf - 1 ? 1 : 0), j1 = j + (j < Wf - 1 ? 1 : 0);
          b4[u] = fmaxf(fmaxf(lab[((size_t)i * Wf + j) * P], lab[((size_t)i * Wf + j1) * P]),
                        fmaxf(lab[((size_t)i1 * Wf + j) * P], lab[((size_t)i1 * Wf + j1) * P]));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c0 + u * RO_BLOCK;
          if (c < n_coarse && b4[u] * (1.f + 4e-6f) >= tau) {
            const int slot = atomicAdd(&n_cells, 1);
            if (slot < RO_CELLS) cells[slot] = c;
          }
        }
      }
      __syncthreads();
      const int nc = n_cells;
      if (nc > RO_CELLS) {
        if (tid == 0) need_scan[map] = 1;
        return;
      }
      // 3. pixels of the listed cells: four threads per cell, rows dealt round-robin
      for (int k = tid >> 2; k < nc; k += RO_BLOCK / 4) {
        const int c = cells[k];
        const int i = c / Wf, j = c - i * Wf;
        const int ya = first_fine_of_cell(i, sy, Hf, h), yb = (i + 1 >= Hf) ? h : first_fine_of_cell(i + 1, sy, Hf, h);
        const int xa = first_fine_of_cell(j, sx, Wf, w), xb = (j + 1 >= Wf) ? w : first_fine_of_cell(j + 1, sx, Wf, w);
        for (int y = ya + (tid & 3); y < yb; y += 4)
          for (int x = xa; x < xb; ++x) {
            const float v = fine_value(lab, Hf, Wf, P, sy, sx, y, x);
            if (v >= tau) {
              const int slot = atomicAdd(&n_cand, 1);
              if (slot < RO_CAND) { cand_v[slot] = v; cand_i[slot] = y * w + x; }
            }
          }
      }
      __syncthreads();
      if (n_cand > RO_CAND || n_cand < K) {
        if (tid == 0) need_scan[map] = 1;
        return;
      }
    }
  }
  // 4. rank the candidates; sum flag: any positive number stands for "the map does not sum to zero"
  if (map_max > 0.f) {
    const int n = n_cand;
    for (int t = tid; t < n; t += RO_BLOCK) {
      const float v = cand_v[t];
      const int id = cand_i[t];
      int rank = 0;
      for (int u = 0; u < n; ++u) {
        const float vu = cand_v[u];
        rank += (vu > v || (vu == v && cand_i[u] > id)) ? 1 : 0;
      }
      if (rank < K) {
        part_v[o * K + rank] = v;
        part_i[o * K + rank] = id;
      }
    }
  }
  if (tid == 0) {
    part_sum[o] = map_max > 0.f ? 1.f : 0.f;
    need_scan[map] = 0;
  }
}

__global__ __launch_bounds__(64) void softargmax_merge_kernel(const float* __restrict__ part_v,
                                                               const int* __restrict__ part_i,
                                                               const float* __restrict__ part_sum, int nbands, int w,
                                                               int n_maps, double* __restrict__ coords) {
  constexpr int K = RO_K;
  const int m = blockIdx.x * 64 + threadIdx.x;      // one thread per (frame, label) map: nbands*5 candidates
  if (m >= n_maps) return;
  TopKHi<K> top;
  top.init();
  float sum = 0.f;
  for (int b = 0; b < nbands; ++b) {
    const size_t o = (size_t)m * nbands + b;
    sum += part_sum[o];
    for (int j = 0; j < K; ++j) {
      const int id = part_i[o * K + j];
      const float v = part_v[o * K + j];
      if (id >= 0 && top.accepts(v, id)) top.insert(v, id);
    }
  }
  double* o = coords + (size_t)m * 2;
  if (sum == 0.f) {  // np.sum(map) == 0  (vanilla_tracker.py:189)
    o[0] = -1.0;
    o[1] = -1.0;
    return;
  }
  float tot = 0.f;
#pragma unroll
  for (int j = K - 1; j >= 0; --j) tot += top.v[j];       // ascending order like np.sum over argsort[-5:]
  tot += 1e-9f;                                            // float32 + python float stays float32
  double ax = 0.0, ay = 0.0;
#pragma unroll
  for (int j = K - 1; j >= 0; --j) {
    const float wgt = top.v[j] / tot;                      // float32 weights (:183)
    ax += (double)(top.ix[j] % w) * (double)wgt;           // int64 * float32 -> float64 (:187)
    ay += (double)(top.ix[j] / w) * (double)wgt;
  }
  o[0] = ax;
  o[1] = ay;
}

// ------------------------------------------------------------------------------------------
// Encoder glue: inference BatchNorm (+ residual add) (+ ReLU) in ONE pass over an NCHW tensor.
// y = (x - mean[c]) * rsqrt(var[c] + eps) * gamma[c] + beta[c]  [+ residual]  [max(.,0)]
// Replaces the separate BN / add / ReLU kernels PyTorch launches after every MIOpen convolution of the
// ResNet (resnet.py:54-116: conv -> BN -> ReLU, conv -> BN, += identity, ReLU): 3 reads + 3 writes of the
// activation become 1-2 reads + 1 write.  float4 accesses (HW % 4 == 0) or scalar tail-safe path.
// ------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void bn_act_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                      const float* __restrict__ mean, const float* __restrict__ var,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, int relu, float* __restrict__ out, int C, int HW,
                                                      long long n) {
  const long long stride = (long long)gridDim.x * 256;
  if constexpr (VEC) {
    const long long n4 = n >> 2;
    const int HW4 = HW >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const int c = (int)((i / HW4) % C);
      const float inv = 1.0f / sqrtf(var[c] + eps);
      const float g = gamma[c], b = beta[c], m = mean[c];
      f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
      v.x = (v.x - m) * inv * g + b; v.y = (v.y - m) * inv * g + b;
      v.z = (v.z - m) * inv * g + b; v.w = (v.w - m) * inv * g + b;
      if (res != nullptr) {
        const f32x4 r = reinterpret_cast<const f32x4*>(res)[i];
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      if (relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      reinterpret_cast<f32x4*>(out)[i] = v;
    }
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      const int c = (int)((i / HW) % C);
      float v = (x[i] - mean[c]) * (1.0f / sqrtf(var[c] + eps)) * gamma[c] + beta[c];
      if (res != nullptr) v += res[i];
      if (relu) v = fmaxf(v, 0.f);
      out[i] = v;
    }
  }
}

int bn_act_launch(const float* x, const float* res, const float* mean, const float* var, const float* gamma,
                  const float* beta, float eps, int relu, float* out, int N, int C, int HW, hipStream_t s) {
  const long long n = (long long)N * C * HW;
  const bool vec = (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) |
                                      reinterpret_cast<uintptr_t>(res)) & 15u) == 0;
  const long long work = vec ? n / 4 : n;
  const int grid = (int)((work + 255) / 256 < 256 * 16 ? (work + 255) / 256 : 256 * 16);
  if (vec)
    bn_act_kernel<true><<<grid, 256, 0, s>>>(x, res, mean, var, gamma, beta, eps, relu, out, C, HW, n);
  else
    bn_act_kernel<false><<<grid, 256, 0, s>>>(x, res, mean, var, gamma, beta, eps, relu, out, C, HW, n);
  FGVC_CHECK_LAUNCH("fgvc_bn_act_f32");
  return FGVC_OK;
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
int normalize_launch(const float* in, float* out, int n, int C, int HW, int normalize, int Cout, hipStream_t s) {
  dim3 grid(cdiv(HW, 32), n);
  const size_t lds = ((size_t)C * 33 + 8 * 32 + 32) * sizeof(float);
  if (lds > 48 * 1024) {  // opt in to the large-LDS carve-out (attribute set, no sync, capture-safe)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(normalize_chw_to_hwc_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("fgvc_normalize_chw_to_hwc_f32: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
      return FGVC_ERR_LAUNCH;
    }
  }
  normalize_chw_to_hwc_kernel<<<grid, 256, lds, s>>>(in, out, C, HW, normalize, Cout);
  FGVC_CHECK_LAUNCH("fgvc_normalize_chw_to_hwc_f32");
  return FGVC_OK;
}

int propagate_launch(const float* labels, const int32_t* slot_frame, int T, const int32_t* idx, const float* weight,
                     int Hq, int Wq, int Hk, int Wk, int P, int topk, int window_L, float* out, hipStream_t s) {
  const long long n = (long long)Hq * Wq * P;
  propagate_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(labels, slot_frame, T, idx, weight, Hq, Wq, Hk, Wk, P,
                                                               topk, window_L, out);
  FGVC_CHECK_LAUNCH("fgvc_propagate_topk_f32");
  return FGVC_OK;
}

int gaussian_launch(const float* points, int P, int Hf, int Wf, int stride, float sigma, float* out, hipStream_t s) {
  const long long n = (long long)Hf * Wf * P;
  gaussian_labels_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(points, P, Hf, Wf, stride,
                                                                     2.f * sigma * sigma, out);
  FGVC_CHECK_LAUNCH("fgvc_gaussian_labels_f32");
  return FGVC_OK;
}

int softargmax_bands() { return 8; }
static int g_readout_prune = 1;
void set_readout_prune(int v) { g_readout_prune = v; }

int softargmax_launch(const float* labels, int n_frames, int Hf, int Wf, int P, int h, int w,
                      const float* gauss_points, float sigma, double* coords, float* ws, hipStream_t s) {
  // workspace layout: part_v [maps][bands][5] f32 | part_i [maps][bands][5] i32 | part_sum [maps][bands] f32 | need_scan [maps] i32
  const int nb = softargmax_bands();
  const size_t maps = (size_t)n_frames * P;
  float* part_v = ws;
  int* part_i = reinterpret_cast<int*>(ws + maps * nb * RO_K);
  float* part_sum = ws + 2 * maps * nb * RO_K;
  int* need_scan = reinterpret_cast<int*>(ws + maps * nb * (2 * RO_K + 1));
  if (g_readout_prune) {
    softargmax_pruned_kernel<<<dim3(P, n_frames), RO_BLOCK, 0, s>>>(labels, Hf, Wf, P, h, w, nb, gauss_points,
                                                                2.f * sigma * sigma, part_v, part_i, part_sum, need_scan);
  }
  dim3 grid(P, n_frames, nb);
  softargmax_band_kernel<<<grid, 256, 0, s>>>(labels, Hf, Wf, P, h, w, nb, gauss_points, 2.f * sigma * sigma, part_v,
                                              part_i, part_sum, g_readout_prune ? need_scan : nullptr);
  softargmax_merge_kernel<<<cdiv((int)maps, 64), 64, 0, s>>>(part_v, part_i, part_sum, nb, w, (int)maps, coords);
  FGVC_CHECK_LAUNCH("fgvc_softargmax_top5_f32");
  return FGVC_OK;
}

}  // namespace fgvc
