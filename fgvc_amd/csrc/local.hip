// Local-window variants of the propagation step.
//   * A7  (HRVanillaTracker, vanilla_tracker.py:547-566 / mmcv.ops.Correlation; torch twin
//          local_attention.py:1190-1240): square (2R+1)^2 window, zero padded.  The in-image taps
//          come from fgvc_pair_topk_f32 (square predicate); local_merge adds the zero-score padded
//          taps, re-indexes to window coordinates, divides by the temperature AFTER top-k (:563)
//          and applies the softmax.
//   * A6  (masked_attention_efficient_c2f, local_attention.py:721-880) fine stage: for every query
//          and key slot a (2Rf+1)^2 window of FINE key features centred on the coarse arg-max cell.
#include "common.hpp"

namespace fgvc {

template <int K>
__global__ __launch_bounds__(256) void local_merge_kernel(const int32_t* __restrict__ pair_idx,
                                                           const float* __restrict__ pair_score, int T, int H, int W,
                                                           int R, int kout, float temperature,
                                                           int32_t* __restrict__ idx_out,
                                                           float* __restrict__ logit_out,
                                                           float* __restrict__ weight_out) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  const int HW = H * W, L = 2 * R + 1;
  if (q >= HW) return;
  const int qy = q / W, qx = q - qy * W;
  const bool border = qy < R || qx < R || qy + R >= H || qx + R >= W;
  TopK<K> top;
  top.init();
  for (int t = 0; t < T; ++t) {
    const size_t o = ((size_t)t * HW + q) * kout;
    for (int j = 0; j < kout; ++j) {
      const int id = pair_idx[o + j];
      if (id < 0) break;
      const int ky = id / W, kx = id - ky * W;
      const int gid = t * L * L + (ky - qy + R) * L + (kx - qx + R);
      const float s = pair_score[o + j];
      if (!top.accepts(s, gid)) break;
      top.insert(s, gid);
    }
    if (border) {  // zero-padded taps: score exactly 0, ascending tap order = canonical tie order
      bool done = false;
      for (int a = 0; a < L && !done; ++a) {
        const int ky = qy + a - R;
        for (int b = 0; b < L; ++b) {
          const int kx = qx + b - R;
          if (ky >= 0 && ky < H && kx >= 0 && kx < W) continue;
          const int gid = t * L * L + a * L + b;
          if (!top.accepts(0.f, gid)) { done = true; break; }
          top.insert(0.f, gid);
        }
      }
    }
  }
  float lg[K], w[K];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) lg[j] = top.v[j] / temperature;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    w[j] = (j < kout) ? expf(lg[j] - lg[0]) : 0.f;
    sum += w[j];
  }
  const size_t o = (size_t)q * kout;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    if (j < kout) {
      idx_out[o + j] = top.ix[j] == IDX_EMPTY ? -1 : top.ix[j];
      logit_out[o + j] = lg[j];
      weight_out[o + j] = w[j] / sum;
    }
  }
}

// A7 get_coord (vanilla_tracker.py:445-488): expected coordinate of each query under its top-k window weights.
// The reference gathers from F.unfold(grid[:, :, ::scale, ::scale], 2R+1, padding=R): tap (dy,dx) of query (y,x)
// carries the image coordinate ((x+dx-R)*scale, (y+dy-R)*scale), and (0,0) where the tap is outside the grid
// (zero padding).  out [HW][2] = (x, y).
__global__ __launch_bounds__(256) void topk_coord_kernel(const int32_t* __restrict__ idx, const float* __restrict__ weight,
                                                          int H, int W, int R, int topk, int scale,
                                                          float* __restrict__ out) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= H * W) return;
  const int L = 2 * R + 1, qy = q / W, qx = q - qy * W;
  float ax = 0.f, ay = 0.f;
  for (int r = 0; r < topk; ++r) {
    const int id = idx[(size_t)q * topk + r];
    if (id < 0) continue;
    const int tap = id % (L * L);
    const int ky = qy + tap / L - R, kx = qx + tap % L - R;
    if (ky < 0 || ky >= H || kx < 0 || kx >= W) continue;
    const float wv = weight[(size_t)q * topk + r];
    ax = fmaf(wv, (float)(kx * scale), ax);
    ay = fmaf(wv, (float)(ky * scale), ay);
  }
  out[2 * q] = ax;
  out[2 * q + 1] = ay;
}

int topk_coord_launch(const int32_t* idx, const float* weight, int H, int W, int R, int topk, int scale, float* out,
                      hipStream_t s) {
  topk_coord_kernel<<<cdiv(H * W, 256), 256, 0, s>>>(idx, weight, H, W, R, topk, scale, out);
  FGVC_CHECK_LAUNCH("fgvc_topk_coord_f32");
  return FGVC_OK;
}

int local_merge_launch(const int32_t* pi, const float* ps, int T, int H, int W, int R, int topk, float temp,
                       int32_t* io, float* lo, float* wo, hipStream_t s) {
  const int grid = cdiv(H * W, 256);
  if (topk <= 1) local_merge_kernel<1><<<grid, 256, 0, s>>>(pi, ps, T, H, W, R, topk, temp, io, lo, wo);
  else if (topk <= 5) local_merge_kernel<5><<<grid, 256, 0, s>>>(pi, ps, T, H, W, R, topk, temp, io, lo, wo);
  else if (topk <= 10) local_merge_kernel<10><<<grid, 256, 0, s>>>(pi, ps, T, H, W, R, topk, temp, io, lo, wo);
  else local_merge_kernel<16><<<grid, 256, 0, s>>>(pi, ps, T, H, W, R, topk, temp, io, lo, wo);
  FGVC_CHECK_LAUNCH("fgvc_local_corr_topk_f32(merge)");
  return FGVC_OK;
}

// ------------------------------------------------------------------------------------------
// c2f fine stage: one wave per query.  Lanes stride over the T*(2Rf+1)^2 candidates (each a Cf-long
// dot product read as float4 rows of the channels-last fine maps), keep a private sorted list, then
// the wave extracts the global top-k with K rounds of a butterfly arg-max.
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void c2f_refine_kernel(const int32_t* __restrict__ coarse_arg,
                                                          const float* __restrict__ qfine,
                                                          const float* __restrict__ kfine,
                                                          const float* __restrict__ vfine, int T, int H, int W,
                                                          int scale, int Cf, int P, int Rf, int kout,
                                                          float temperature, int weight_mode, float* __restrict__ out,
                                                          int32_t* __restrict__ idx_out,
                                                          float* __restrict__ logit_out) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int HW = H * W;
  if (q >= HW) return;  // whole wave exits together
  const int sH = H * scale, sW = W * scale, L = 2 * Rf + 1, LL = L * L;
  const int qy = q / W, qx = q - qy * W;
  const float* qv = qfine + ((size_t)(qy * scale) * sW + qx * scale) * Cf;   // query_fine[:, ::scale, ::scale] (:785)
  TopK<K> top;
  top.init();
  const int lpr = Cf >> 2;                      // lanes that read one candidate's Cf-channel row (16 B each)
  if (lpr <= 64 && (lpr & (lpr - 1)) == 0) {
    // Row-cooperative mapping: 64/lpr candidates per wave instruction, every row a coalesced Cf*4-byte read (with one
    // lane per candidate every load instruction touched 64 different cache lines for 16 bytes each and the kernel was
    // bound by the texture-address path: 2.9 ms per frame at 480p).  The group's lanes reduce by shuffles; its first
    // lane keeps the group's running list.
    const int grp = lane / lpr, sub = lane - grp * lpr, G = 64 / lpr;
    const f32x4 qa = *reinterpret_cast<const f32x4*>(qv + 4 * sub);
    constexpr int U = 4;                          // candidates in flight per lane group (8: no faster)
    for (int c0 = grp; c0 < T * LL; c0 += G * U) {
      f32x4 a[U];
      bool in[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * G;
        in[u] = false;
        a[u] = {0.f, 0.f, 0.f, 0.f};
        if (c < T * LL) {
          const int t = c / LL, tap = c - t * LL;
          const int cell = coarse_arg[(size_t)t * HW + q];
          const int fy = (cell / W) * scale + tap / L - Rf, fx = (cell % W) * scale + tap % L - Rf;
          if (fy >= 0 && fy < sH && fx >= 0 && fx < sW) {
            in[u] = true;
            a[u] = *reinterpret_cast<const f32x4*>(kfine + (((size_t)t * sH + fy) * sW + fx) * Cf + 4 * sub);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * G;
        float sp = a[u].x * qa.x;
        sp = fmaf(a[u].y, qa.y, sp); sp = fmaf(a[u].z, qa.z, sp); sp = fmaf(a[u].w, qa.w, sp);
        for (int m = lpr >> 1; m >= 1; m >>= 1) sp += __shfl_xor(sp, m);
        const float sc = (in[u] ? sp : 0.f) / temperature;            // (:847) divided BEFORE the top-k here
        if (sub == 0 && c < T * LL && top.accepts(sc, c)) top.insert(sc, c);
      }
    }
  } else {
    for (int c = lane; c < T * LL; c += 64) {
      const int t = c / LL, tap = c - t * LL;
      const int cell = coarse_arg[(size_t)t * HW + q];
      const int fy = (cell / W) * scale + tap / L - Rf, fx = (cell % W) * scale + tap % L - Rf;
      float s = 0.f;
      if (fy >= 0 && fy < sH && fx >= 0 && fx < sW) {
        const float* kv = kfine + (((size_t)t * sH + fy) * sW + fx) * Cf;
        for (int ch = 0; ch < Cf; ch += 4) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(kv + ch);
          const f32x4 b = *reinterpret_cast<const f32x4*>(qv + ch);
          s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
        }
      }
      s = s / temperature;  // (:847) divided BEFORE the top-k here
      if (top.accepts(s, c)) top.insert(s, c);
    }
  }
  float win_s[K];
  int win_i[K];
#pragma unroll
  for (int r = 0; r < K; ++r) {
    float bs = top.v[0];
    int bi = top.ix[0];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const float os = __shfl_xor(bs, m);
      const int oi = __shfl_xor(bi, m);
      const bool take = os > bs || (os == bs && oi < bi);
      bs = take ? os : bs;
      bi = take ? oi : bi;
    }
    win_s[r] = bs;
    win_i[r] = bi;
    if (top.ix[0] == bi && bi != IDX_EMPTY) {  // the owner pops its head
#pragma unroll
      for (int j = 0; j + 1 < K; ++j) {
        top.v[j] = top.v[j + 1];
        top.ix[j] = top.ix[j + 1];
      }
      top.v[K - 1] = -INFINITY;
      top.ix[K - 1] = IDX_EMPTY;
    }
  }
  float w[K];
  float sum = 0.f;
  if (weight_mode == FGVC_WEIGHT_SOFTMAX) {
#pragma unroll
    for (int r = 0; r < K; ++r) {
      w[r] = (r < kout) ? expf(win_s[r] - win_s[0]) : 0.f;
      sum += w[r];
    }
  } else {                                  // 'cosine' (local_attention.py:860-861): clamp(affinity, 0)^2, not normalised
#pragma unroll
    for (int r = 0; r < K; ++r) {
      const float c = (r < kout) ? fmaxf(win_s[r], 0.f) : 0.f;
      w[r] = c * c;
    }
    sum = 1.f;
  }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < K; ++r)
      if (r < kout) {
        idx_out[(size_t)q * kout + r] = win_i[r] == IDX_EMPTY ? -1 : win_i[r];
        logit_out[(size_t)q * kout + r] = win_s[r];
      }
  }
  for (int pl = lane; pl < P; pl += 64) {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < K; ++r) {
      if (r < kout && win_i[r] != IDX_EMPTY) {
        const int t = win_i[r] / LL, tap = win_i[r] - t * LL;
        const int cell = coarse_arg[(size_t)t * HW + q];
        const int fy = (cell / W) * scale + tap / L - Rf, fx = (cell % W) * scale + tap % L - Rf;
        if (fy >= 0 && fy < sH && fx >= 0 && fx < sW)
          acc = fmaf(w[r] / sum, vfine[(((size_t)t * sH + fy) * sW + fx) * P + pl], acc);
      }
    }
    out[(size_t)q * P + pl] = acc;
  }
}

int c2f_refine_launch(const int32_t* coarse_arg, const float* qfine, const float* kfine, const float* vfine, int T,
                      int H, int W, int scale, int Cf, int P, int Rf, int topk, float temperature, int weight_mode, float* out,
                      int32_t* idx_out, float* logit_out, hipStream_t s) {
  const int grid = cdiv(H * W, 4);
#define FGVC_C2F(KK)                                                                                            \
  c2f_refine_kernel<KK><<<grid, 256, 0, s>>>(coarse_arg, qfine, kfine, vfine, T, H, W, scale, Cf, P, Rf, topk, \
                                             temperature, weight_mode, out, idx_out, logit_out)
  if (topk <= 1) FGVC_C2F(1);
  else if (topk <= 5) FGVC_C2F(5);
  else if (topk <= 10) FGVC_C2F(10);
  else FGVC_C2F(16);
#undef FGVC_C2F
  FGVC_CHECK_LAUNCH("fgvc_c2f_refine_f32");
  return FGVC_OK;
}

}  // namespace fgvc
