// fgvc_pair_topk_f16x3: the windowed correlation + top-k of fgvc_pair_topk_f32 on the f16 matrix pipe, f32-grade, with the
// matrix chain and the selection in ONE instruction stream and a ring of key blocks without workgroup barriers.
//
// What its predecessor (a bf16 hi/lo four-product kernel with a barrier per two key blocks, retired in round 3; docs/LAB_NOTES.md)
// left on the table, by its own ablations: the two waves of a SIMD alternated between a 64-MFMA chain (~2900 cycles for 2048 pipe
// cycles) and the selection of their previous tile (~2000), and every wave stepped through the UNION of the key blocks its 2 x 2
// query blocks reach (56 entries where a block reaches 41: 27 % of the multiply slots empty).  Here:
//   * Arithmetic: x -> h = f16(2^14 x), l = f16(2^14 x - h) (fgvc_split_f16x2): 22 significand bits per element (bf16 hi + lo:
//     16), products of f16 are exact in f32, and  2^28 <k, q> = sum h h + sum l h + sum h l  (sum l l = 2^-22, dropped) on
//     v_mfma_f32_32x32x16_f16: 48 MFMAs = 1536 pipe cycles per 32 x 32 tile (bf16 hi/lo with four products: 64 = 2048).  The scale makes the accumulator
//     the score in 2^-28 fixed point as it stands; rows must be L2-normalised (|x| <= 1: 2^14 x fits f16).
//   * Roles: waves 0-3 = consumers, one 4 x 8 query block each (B operands resident); waves 4-7 = producers, one pixel row of
//     every key block each, by LDS-DMA.  A consumer multiplies key block e and, IN THE SAME STREAM, selects the candidates of the
//     tile it computed before: the VALU work (~290 operations: mask predicate, integer keys, 60-comparator selection network,
//     bitonic merge into the running list) is cut into 48 slices, one behind each MFMA, every slice one generated inline-asm
//     statement (pair_v5_chain.inc, tools/gen_pair_v5_chain.py: an MFMA holds the SIMD's vector issue for 8 of its 32 cycles).
//   * Ring: 4 slots, two counters per slot in LDS.  filled[s] counts producer arrivals (4 per key block), done[s] consumer
//     releases (4 per key block, also by the consumers that do not reach the block).  A producer refills slot s with block e + 4
//     once done[s] shows block e released by all four consumers; a consumer reads block e once filled[s] shows its four rows
//     landed.  No workgroup barrier after the prologue: a consumer whose query block does not reach a key block moves on (once the
//     block has landed: a release must never be sent for a block that is not staged yet, the slot's counter would take it for the
//     block before), up to the ring's depth ahead of the others.  With the list in alternating order (first, last, second, ...) every window of 4
//     entries is reached about evenly, and the loop takes ~42 tile times instead of 56 (tools/sim_pair_ring.py).
//     Every spin is bounded: after 2^16 polls a wave raises g_pair_v5_timeout and the workgroup's LDS flag and stops waiting; the
//     launch still ends, and the lists that workgroup writes from then on are POISON (index 0, score +inf -> NaN weights, labels and
//     trajectories downstream): fail closed.  fgvc_pair_topk_f16x3_timed_out() reports (and clears) the flag.
#include "pair_common.hpp"

namespace fgvc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float F16X2_SCALE = 16384.f;

// f32 rows -> [h C | l C] f16 per pixel
__global__ __launch_bounds__(256) void split_f16x2_kernel(const float* __restrict__ feat, uint16_t* __restrict__ out,
                                                           long long n_vec4, int C) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= n_vec4) return;
  const long long e = g * 4;
  const long long pix = e / C;
  const int c = (int)(e - pix * C);
  const f32x4 x = *reinterpret_cast<const f32x4*>(feat + e);
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  f16x4 h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float xs = x[i] * F16X2_SCALE;
    h[i] = (_Float16)xs;
    l[i] = (_Float16)(xs - (float)h[i]);
  }
  *reinterpret_cast<f16x4*>(out + pix * 2 * C + c) = h;
  *reinterpret_cast<f16x4*>(out + pix * 2 * C + C + c) = l;
}

int split_f16x2_launch(const float* feat, uint16_t* out, long long n_pixels, int C, hipStream_t s) {
  const long long n4 = n_pixels * C / 4;
  split_f16x2_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, s>>>(feat, out, n4, C);
  FGVC_CHECK_LAUNCH("fgvc_split_f16x2");
  return FGVC_OK;
}

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
  return (uint32_t)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}

__device__ int g_pair_v5_timeout = 0;
__device__ long long g_pair_v5_probe[32];      // debug & 256: s_memtime stamps of one workgroup (tools/experiments/time_pair_v5.py)

// bounded spin on an LDS word (wave-uniform): true = the word reached `target`.  A wave that has given up once (`dead`) never
// waits again: a broken protocol costs milliseconds, not a hung GPU.
// Fail closed: the wave that gives up also raises the workgroup's LDS flag `wg_dead`; the role that owns the output lists reads it
// before every store and writes POISON lists (index 0, score +inf: the merge's softmax turns that into NaN weights, the sweep into NaN
// labels and trajectories) for every pair that may have seen an unsynchronised key block -- a consumer of the lists cannot mistake them
// for results.  (A producer raises the flag BEFORE it stages anything unsynchronised, and the LDS executes a wave's operations in order:
// whoever sees the block's `filled` count also sees the flag.)
constexpr float PAIR_POISON_SCORE = INFINITY;
template <int SLEEP = 2, bool REPORT = true>   // REPORT = false: the caller raises the global flag itself when it ends (two registers less across its loop)
__device__ __forceinline__ bool spin_ge(volatile int* w, int target, bool& dead, volatile int* wg_dead, long long* waited = nullptr) {
  if (dead) return false;
  const long long t0 = waited ? __builtin_amdgcn_s_memtime() : 0;
  struct Stamp {
    long long* w; long long t0;
    __device__ ~Stamp() { if (w) *w += __builtin_amdgcn_s_memtime() - t0; }
  } stamp{waited, t0};
  // the word lives in the LDS: a ds_read (one address register), not the flat load a generic pointer compiles to
  const volatile __attribute__((address_space(3))) int* wl = (const volatile __attribute__((address_space(3))) int*)w;
  for (int it = 0; it < (1 << 16); ++it) {
    const int v = __builtin_amdgcn_readfirstlane(*wl);
    if (v >= target) return true;
    __builtin_amdgcn_s_sleep(SLEEP);
  }
  if (REPORT) g_pair_v5_timeout = 1;
  *wg_dead = 1;
  dead = true;
  return false;
}

template <int K, int DBG>   // DBG (compile time, so that the pieces stay branch-free; results wrong): 1 = no selection, 2 = no MFMA
__global__ __launch_bounds__(512, 1) void pair_topk_kernel_v5(PairParamsB p) {
  constexpr int C = 256;
  constexpr int LDB = 2 * C * 2 + 16;          // padded LDS row of one pixel: [h | l] + 16 B -> conflict-free b128
  constexpr int BUFB = 32 * LDB;
  constexpr int NSLOT = 4;
  constexpr int KS = C / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * BUFB];
  __shared__ uint32_t blist[PAIR_LIST_CAP];      // by | bx << 12 | (query blocks that reach it) << 24
  __shared__ int blist_n;
  __shared__ int filled[NSLOT], done[NSLOT];
  __shared__ int wg_dead;                        // a wave of this workgroup gave up waiting: the lists written from here on are poison

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = wave & 3, par = wave >> 2;      // par 0 = consumer of query block qb, par 1 = producer of pixel row qb
  const int n = lane & 31, hi = lane >> 5;

  // a workgroup takes a RUN of pairs that share the query frame and the mask flag (blockIdx.y = run): the query rows, the block list
  // and the B operands are set up once, and the producers run from the last key block of one pair into the first of the next
  int g_start = blockIdx.y, g_count = 1;
  if (p.groups) {
    const int2 gr = p.groups[blockIdx.y];
    g_start = gr.x;
    g_count = gr.y;
  }
  const int4 pr = p.pairs[g_start];
  const int qf = pr.x;
  const bool masked = (pr.z & FGVC_PAIR_MASKED) != 0;
  const int reach_y = masked ? p.reach_y : FGVC_NO_LIMIT;
  const int reach_x = masked ? p.reach_x : FGVC_NO_LIMIT;

  const int tile = xcd_remap(blockIdx.x, p.n_ty * p.n_tx);
  const int ty = tile / p.n_tx, tx = tile - ty * p.n_tx;
  ReachTest reach;
  reach.r2max = masked ? p.r2max : FGVC_NO_LIMIT;
  reach.ry = masked ? p.ry : FGVC_NO_LIMIT;
  reach.rx = masked ? p.rx : FGVC_NO_LIMIT;
  const int TY0 = ty * (2 * QBH), TX0 = tx * (2 * QBW);
  const int QY0 = TY0 + (qb & 1) * QBH, QX0 = TX0 + (qb >> 1) * QBW;
  const int qy = QY0 + (n >> 3), qx = QX0 + (n & 7);

  // ---- prologue 1: the query rows of the four blocks through the ring (coalesced 1 KiB rows by LDS-DMA)
  {
    const uint16_t* qbase = p.q_hl + (size_t)qf * p.Hq * p.Wq * (2 * C) + 8 * lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = par * 16 + i;                                  // row of query block qb
      const int y = imin(QY0 + (r >> 3), p.Hq - 1), x = imin(QX0 + (r & 7), p.Wq - 1);
      lds_dma_16(qbase + ((size_t)y * p.Wq + x) * (2 * C), &smem[qb * BUFB + r * LDB]);
    }
  }
  // ---- prologue 2 (overlaps the DMA): the list of key blocks this super-tile visits, row-major, then re-ordered
  //      first, last, second, second to last, ...: the entries only the upper (lower) query blocks reach sit at the head (tail) of
  //      the row-major list; alternating them keeps every consumer busy within the ring's depth
  if (wave == 0) {
    const int by_lo = imax(0, TY0 - imin(reach_y, TY0)) / QBH;
    const int by_hi = imin(p.Hk - 1, TY0 + 2 * QBH - 1 + imin(reach_y, p.Hk)) / QBH;
    const int bxl = imax(0, TX0 - imin(reach_x, TX0)) / QBW;
    const int bxh = imin(p.Wk - 1, TX0 + 2 * QBW - 1 + imin(reach_x, p.Wk)) / QBW;
    const int nbx = bxh - bxl + 1;
    const int nall = (by_hi - by_lo + 1) * nbx;
    // the host has checked that a MASKED pair's reach fits the list; an unmasked pair on a larger key grid than the list holds
    // (the caller promised there was none: `all_masked`) gets EMPTY lists (-1 / -inf), never truncated ones
    const int ncand = nall > PAIR_LIST_CAP ? 0 : nall;
    auto reach_bits = [&](int c) -> uint32_t {
      const int by = by_lo + c / nbx, bx = bxl + c % nbx;
      uint32_t m = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b)
        m |= (uint32_t)reach(TY0 + (b & 1) * QBH, TX0 + (b >> 1) * QBW, by * QBH, bx * QBW) << b;
      return m ? ((uint32_t)by | ((uint32_t)bx << 12) | (m << 24)) : 0u;
    };
    int total = 0;                                                   // pass 1: how many blocks are reached at all
    for (int base = 0; base < ncand; base += 64) {
      const int c = base + lane;
      total += __popcll(__ballot(c < ncand && reach_bits(c) != 0u));
    }
    int count = 0;                                                   // pass 2: row-major rank r -> position 2 r | 2 (total - 1 - r) + 1
    const int head = (total + 1) >> 1;
    for (int base = 0; base < ncand; base += 64) {
      const int c = base + lane;
      const uint32_t ent = c < ncand ? reach_bits(c) : 0u;
      const unsigned long long bal = __ballot(ent != 0u);
      if (ent) {
        const int r = count + __popcll(bal & ((1ull << lane) - 1));
        blist[(p.debug & 8) ? r : (r < head ? 2 * r : 2 * (total - 1 - r) + 1)] = ent;      // debug & 8: plain row-major order
      }
      count += __popcll(bal);
    }
    if (lane == 0) blist_n = count;
  }
  if (tid < NSLOT) {
    filled[tid] = 0;
    done[tid] = 0;
  }
  if (tid == 0) wg_dead = (p.debug & 4096) ? 1 : 0;      // 4096: fault injection for the fail-closed test
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int n_steps = blist_n;
  bool dead = false;

  if (par == 1) {
    // =========================================== producer: pixel row qb of every key block ===========================================
    __syncthreads();                                                 // the consumers have read their query fragments: the ring is free
    if (p.debug & (16 | 64)) return;
    const uint32_t lane16 = 16u * lane;
    const int n_total = g_count * n_steps;                             // key blocks of the whole run, numbered G = pair * n_steps + e
    int cur_pair = -1;
    const unsigned char* kbase = nullptr;
    auto stage = [&](int G) {
      const int pi = G / n_steps, e = G - pi * n_steps;
      if (pi != cur_pair) {                                            // wave-uniform
        cur_pair = pi;
        const int kf = p.pairs[g_start + pi].y;
        kbase = reinterpret_cast<const unsigned char*>(p.k_hl) + (size_t)kf * p.Hk * p.Wk * (4 * C);
      }
      const uint32_t ent = __builtin_amdgcn_readfirstlane(blist[e]);
      int sby = ent & 0xfff, sbx = (ent >> 12) & 0xfff;
      if (p.debug & 32768) { sby = blockIdx.x & 3; sbx = 0; }         // experiment: every workgroup reads the same few key blocks (L2-hot)
      const int ky = imin(sby * QBH + qb, p.Hk - 1), kx0 = sbx * QBW;
      const unsigned char* src = kbase + ((size_t)ky * p.Wk + kx0) * (4 * C) + lane16;
      const int xmax = p.Wk - 1 - kx0;                                 // >= 0: the block starts inside the frame
      unsigned char* dst = &smem[(G & (NSLOT - 1)) * BUFB + (qb * 8) * LDB];
      if ((p.debug & 16384) && lane >= 32) return;                     // experiment: half of every row (bytes halved, instructions unchanged)
#pragma unroll
      for (int i = 0; i < 8; ++i) lds_dma_16(src + (size_t)imin(i, xmax) * (4 * C), dst + i * LDB);
    };
    const bool probe = (p.debug & 256) && blockIdx.x == 100 && blockIdx.y == 5;
    long long pw = 0, pt0 = probe ? __builtin_amdgcn_s_memtime() : 0;
    if ((p.debug & 4) == 0 && n_total > 0) {
      stage(0);
      for (int G = 0; G < n_total; ++G) {
        if (G + 1 < n_total) {
          const int s1 = (G + 1) & (NSLOT - 1), gen1 = (G + 1) / NSLOT;
          if (gen1 > 0) spin_ge<6>(&done[s1], 4 * gen1, dead, &wg_dead, probe ? &pw : nullptr);   // block G + 1 - NSLOT released by all four consumers
          asm volatile("" ::: "memory");                                                // (slots free up a tile time apart: poll rarely)
          stage(G + 1);
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            // block G landed, block G + 1 in flight
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (lane == 0) __hip_atomic_fetch_add(&filled[G & (NSLOT - 1)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else if (n_total > 0) {                                          // ablation: no staging, every block "arrives" at once
      for (int G = 0; G < n_total; ++G)
        if (lane == 0) __hip_atomic_fetch_add(&filled[G & (NSLOT - 1)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (probe && lane == 0) {
      g_pair_v5_probe[16 + 4 * qb] = __builtin_amdgcn_s_memtime() - pt0;     // producer: loop cycles, cycles waiting for a free slot
      g_pair_v5_probe[16 + 4 * qb + 1] = pw;
      g_pair_v5_probe[16 + 4 * qb + 2] = n_steps;
    }
    return;
  }

  // =============================================== consumer: query block qb ===============================================
  // query fragments (MFMA B operand): lane (n, hi) holds channels 16 j + 8 hi .. + 7 of query n, h and l parts
  f16x8 qh[KS], ql[KS];
  {
    const unsigned char* qp = &smem[qb * BUFB + n * LDB + 16 * hi];
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      qh[j] = *reinterpret_cast<const f16x8*>(qp + 32 * j);
      ql[j] = *reinterpret_cast<const f16x8*>(qp + 2 * C + 32 * j);
    }
  }
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    asm volatile("" ::"v"(qh[j]));
    asm volatile("" ::"v"(ql[j]));
  }
  __syncthreads();                               // the ring is free for key blocks
  if (p.debug & 16) return;                      // ablation: prologue only

  int lk[K], lb[K];                              // running list, ASCENDING: lk[0] = K-th best ... lk[K-1] = best
  f32x16 acc;
  int ck[16];                                    // the pending tile: fixed-point scores, turned into keys in place by pieces 0 and 1
#pragma unroll
  for (int r = 0; r < 16; ++r) ck[r] = 0;

  // Inputs of the generated selection slices (pair_v5_chain.inc) for the pending tile: the candidate in accumulator register r of
  // lane (n, hi) is key pixel (ky0 + (r >> 2), kx0 + 4 hi + (r & 3)); it is kept iff (dx0 + (r & 3))^2 <= r2lim - (dy0 + (r >> 2))^2.
  // Tiles that need more than the disc (a key block that leaves the frame, a rectangular window) are resolved right after their
  // chain by the general predicate: rejected candidates become KEY_EMPTY scores and the slices see an always-true disc.
  int v_dy0 = 0, v_dx0 = 0, v_base = 0;
  int s_r2lim = -1;                              // no pending tile: every candidate is rejected
  int v_empty = KEY_EMPTY;
  asm volatile("" : "+v"(v_empty));              // keep it in a register: the slices read it as an operand
  constexpr bool do_sel = (DBG & 1) == 0, do_mfma = (DBG & 2) == 0;
  auto flush_pending = [&]() {                    // the same slices without a chain around them
#define FGVC_V5_PART 2
    if constexpr (K == 10) {
#define FGVC_V5_K 10
#include "pair_v5_chain.inc"
#undef FGVC_V5_K
    } else {
#define FGVC_V5_K 5
#include "pair_v5_chain.inc"
#undef FGVC_V5_K
    }
#undef FGVC_V5_PART
  };
  const int n_loop = (p.debug & 64) ? 0 : n_steps;
  const bool probe = (p.debug & 256) && blockIdx.x == 100 && blockIdx.y == 5;
  long long cw = 0, cchain = 0, ct0 = probe ? __builtin_amdgcn_s_memtime() : 0;
  int n_comp = 0;
  for (int pi = 0; pi < g_count; ++pi) {           // ---- the pairs of the run, one after the other through the same ring
#pragma unroll
  for (int j = 0; j < K; ++j) {
    lk[j] = KEY_EMPTY;
    lb[j] = -1;
  }
  s_r2lim = -1;                                    // no pending tile
  uint32_t ent_next = n_loop > 0 ? blist[0] : 0u;
  for (int e = 0; e < n_loop; ++e) {
    const uint32_t ent = __builtin_amdgcn_readfirstlane(ent_next);
    ent_next = e + 1 < n_loop ? blist[e + 1] : 0u;          // lands during this entry's work
    const int G = pi * n_steps + e;                          // the ring counts key blocks over the whole run
    const int slot = G & (NSLOT - 1), gen = G / NSLOT;
    const bool comp = ((ent >> (24 + qb)) & 1) != 0;
    // Block G has landed (its four pixel rows).  A consumer that does not reach the block waits for this too before it releases the
    // slot: the counters are per SLOT, so a release sent before block G was staged would be counted for block G - NSLOT, which a
    // slower consumer may still be reading -- the producers would then refill the slot under it (seen as rare wrong scores with one
    // pair per workgroup and often with runs of pairs, where half of the consumers skip the first blocks of a pair).
    spin_ge(&filled[slot], 4 * (gen + 1), dead, &wg_dead, probe ? &cw : nullptr);
    asm volatile("" ::: "memory");
    if (!comp) {                                  // not within this query block's reach: release and move on
      if (lane == 0) __hip_atomic_fetch_add(&done[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      continue;
    }
    const long long cc0 = probe ? __builtin_amdgcn_s_memtime() : 0;
    if constexpr (do_mfma) {
      const unsigned char* ka = &smem[slot * BUFB + n * LDB + 16 * hi];
      // A fragments: a ring of three K-16 steps, read two steps (6 MFMAs, ~190 cycles) ahead: one step ahead the chain waited
      // ~30 cycles per step for the LDS (s_memtime probe: 42 cycles per MFMA without any selection work)
      f16x8 ah[3], al[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(ka + 32 * i);
        al[i] = *reinterpret_cast<const f16x8*>(ka + 2 * C + 32 * i);
      }
      // The chain: 48 MFMAs as volatile inline assembly, one generated slice of the pending tile's selection (~6 vector
      // instructions, one asm statement) behind each, fragment reads one K-16 step ahead (tools/gen_pair_v5_chain.py has the why).
#define FGVC_V5_RELEASE()                                                                                                        \
  do {                                                                                                                           \
    /* every LDS read of this block has been issued (the LDS executes a wave's operations in order): release the slot */         \
    asm volatile("" ::: "memory");                                                                                               \
    if (lane == 0) __hip_atomic_fetch_add(&done[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                      \
    asm volatile("" ::: "memory");                                                                                               \
  } while (0)
#define FGVC_V5_PART 1
      if constexpr (K == 10) {
#define FGVC_V5_K 10
#include "pair_v5_chain.inc"
#undef FGVC_V5_K
      } else {
#define FGVC_V5_K 5
#include "pair_v5_chain.inc"
#undef FGVC_V5_K
      }
#undef FGVC_V5_PART
#undef FGVC_V5_RELEASE
      // MFMA result -> vector read: the last MFMA's 8 passes must have written back (18 wait states for a 16-pass-equivalent op)
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
    } else {
      if (lane == 0) __hip_atomic_fetch_add(&done[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if constexpr (do_sel) flush_pending();
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
    // this tile becomes the pending one
#pragma unroll
    for (int r = 0; r < 16; ++r) ck[r] = (int)acc[r];
    if constexpr (!do_sel) {                      // ablation: keep the chain alive
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(ck[r]));
    }
    {
      const int ky0 = (int)(ent & 0xfff) * QBH, kx0 = (int)((ent >> 12) & 0xfff) * QBW;
      v_base = ky0 * p.Wk + kx0;
      const bool interior = ky0 + QBH <= p.Hk && kx0 + QBW <= p.Wk;
      const bool circle_only = reach.ry >= FGVC_NO_LIMIT && reach.rx >= FGVC_NO_LIMIT;
      if (interior && circle_only) {              // wave-uniform
        v_dy0 = ky0 - qy;
        v_dx0 = kx0 + 4 * hi - qx;
        s_r2lim = reach.r2max;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dy = ky0 - qy + (r >> 2), dx = kx0 + 4 * hi - qx + (r & 3);
          const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
          const int cy = imin(ady, 32767), cx = imin(adx, 32767);       // squares stay below 2^30 on any supported grid
          const bool ok = (ky0 + (r >> 2) < p.Hk) & (kx0 + 4 * hi + (r & 3) < p.Wk) &
                          (cy * cy + cx * cx <= reach.r2max) & (ady <= reach.ry) & (adx <= reach.rx);
          ck[r] = ok ? ck[r] : KEY_EMPTY;
        }
        v_dy0 = 0;
        v_dx0 = 0;
        s_r2lim = FGVC_NO_LIMIT;                  // (r & 3)^2 <= FGVC_NO_LIMIT - (r >> 2)^2: always kept
      }
    }
    if (probe) { cchain += __builtin_amdgcn_s_memtime() - cc0; ++n_comp; }
  }
  if constexpr (do_sel) flush_pending();          // the last pending tile
  if (p.debug & 32) continue;                    // ablation: no epilogue

  // ---- epilogue: two partial lists per query (the two lane halves) -> canonical top-K.  Entries become 64-bit words
  //      (score_fx : ~pixel), larger = better (higher score, then LOWER pixel index)
  long long L[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int key = lk[i];
    const int r = 15 - (key & 15);
    const int pix = lb[i] + 4 * hi + (r >> 2) * p.Wk + (r & 3);
    const bool em = key < (int)0xC0000000;      // KEY_EMPTY, or KEY_EMPTY | register tag (a candidate the general predicate rejected)
    const uint32_t lo = em ? 0u : ~(uint32_t)pix;
    L[i] = (long long)(((unsigned long long)(uint32_t)(key & ~15) << 32) | lo);
  }
  // a lane's list is ascending in (score, tag); entries of different tiles with the same score may be out of pixel order
#define X(I, J)                                   \
  {                                               \
    const bool b_ = L[I] < L[J];                  \
    const long long lo_ = b_ ? L[I] : L[J];       \
    const long long hi_ = b_ ? L[J] : L[I];       \
    L[J] = lo_; L[I] = hi_;                       \
  }
  if constexpr (K == 10) { FGVC_SORTNET_10(X) }    // descending in I < J ...
  else { FGVC_SORTNET_5(X) }
#undef X
#pragma unroll
  for (int i = 0; i < K / 2; ++i) {                // ... so reverse: ascending like the running list
    const long long tmp = L[i];
    L[i] = L[K - 1 - i];
    L[K - 1 - i] = tmp;
  }
  {
    long long B[K];
#pragma unroll
    for (int i = 0; i < K; ++i) B[i] = __shfl_xor(L[i], 32);
#pragma unroll
    for (int i = 0; i < K; ++i) L[i] = L[i] > B[K - 1 - i] ? L[i] : B[K - 1 - i];
#define X(I, J)                                   \
    {                                             \
      const bool b_ = L[I] > L[J];                \
      const long long lo_ = b_ ? L[J] : L[I];     \
      const long long hi_ = b_ ? L[I] : L[J];     \
      L[I] = lo_; L[J] = hi_;                     \
    }
    if constexpr (K == 10) { FGVC_VMERGE_ASC_10(X) }
    else { FGVC_VMERGE_ASC_5(X) }
#undef X
  }
  const bool poison = dead || __builtin_amdgcn_readfirstlane(*(volatile int*)&wg_dead) != 0;
  if (poison) g_pair_v5_timeout = 1;
  if (hi == 0 && qy < p.Hq && qx < p.Wq) {
    const size_t oo = ((size_t)(g_start + pi) * p.Hq * p.Wq + (size_t)qy * p.Wq + qx) * p.kout;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (j < p.kout) {
        const long long v = L[K - 1 - j];
        const int sk = (int)(v >> 32);
        const bool em = sk < (int)0xC0000000;
        p.idx_out[oo + j] = poison ? 0 : em ? -1 : (int)~(uint32_t)v;
        p.score_out[oo + j] = poison ? PAIR_POISON_SCORE : em ? -INFINITY : (float)sk * 0x1p-28f;
      }
    }
  }
  }   // pairs of the run
  if (probe && lane == 0) {
    g_pair_v5_probe[4 * qb] = __builtin_amdgcn_s_memtime() - ct0;     // consumer: loop cycles, cycles waiting for data, chain cycles, tiles
    g_pair_v5_probe[4 * qb + 1] = cw;
    g_pair_v5_probe[4 * qb + 2] = cchain;
    g_pair_v5_probe[4 * qb + 3] = n_comp;
  }
}

// ---- Three roles (fgvc_pair_topk_f16x3, default form): the consumer of a query block only multiplies; the selection of its tiles
// runs on a SELECTOR wave of the same SIMD, whose vector instructions issue in the shadows of the consumer's MFMAs
// (tools/micro/mfma_chain_lds.hip: a partner wave gets 6.3-6.7 vector instructions per MFMA slot while the chain stays at 38-39
// cycles per MFMA; merged into one stream the same work cost 53).  Waves 0-3 consumers, 4-7 selectors, 8-11 producers: three
// waves per SIMD = 168 registers each, which a consumer (128 of query operands + 16 accumulators + key fragments) just fits
// without the selection state.  A tile's 16 fixed-point scores per lane go from consumer to selector through 4 KiB of LDS per
// query block (four conflict-free ds_write_b128), guarded by two counters like the ring's.
constexpr int V6_LIST_CAP = 2048;                // the hand-over buffers take half of the two-role form's list space

template <int K>
__global__ __launch_bounds__(768, 1) void pair_topk_kernel_v6(PairParamsB p) {
  constexpr int DBG = 0;
  constexpr int C = 256;
  constexpr int LDB = 2 * C * 2 + 16;          // padded LDS row of one pixel: [h | l] + 16 B -> conflict-free b128
  constexpr int BUFB = 32 * LDB;
  constexpr int NSLOT = 4;
  constexpr int KS = C / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * BUFB];
  __shared__ uint32_t blist[V6_LIST_CAP];      // by | bx << 12 | (query blocks that reach it) << 24
  __shared__ __attribute__((aligned(16))) int hand[4][16 * 64];   // consumer -> selector: a tile's fixed-point scores, register-major per 16-byte piece
  __shared__ int hand_full[4], hand_free[4];
  __shared__ int blist_n;
  __shared__ int filled[NSLOT], done[NSLOT];
  __shared__ int wg_dead;                        // a wave of this workgroup gave up waiting: the lists written from here on are poison

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = wave & 3, role = wave >> 2;     // role 0 = consumer of query block qb, 1 = its selector, 2 = producer of pixel row qb
  const int par = role;
  const int n = lane & 31, hi = lane >> 5;

  // a workgroup takes a RUN of pairs that share the query frame and the mask flag (blockIdx.y = run): the query rows, the block list
  // and the B operands are set up once, and the producers run from the last key block of one pair into the first of the next
  int g_start = blockIdx.y, g_count = 1;
  if (p.groups) {
    const int2 gr = p.groups[blockIdx.y];
    g_start = gr.x;
    g_count = gr.y;
  }
  const int4 pr = p.pairs[g_start];
  const int qf = pr.x;
  const bool masked = (pr.z & FGVC_PAIR_MASKED) != 0;
  const int reach_y = masked ? p.reach_y : FGVC_NO_LIMIT;
  const int reach_x = masked ? p.reach_x : FGVC_NO_LIMIT;

  const int tile = xcd_remap(blockIdx.x, p.n_ty * p.n_tx);
  const int ty = tile / p.n_tx, tx = tile - ty * p.n_tx;
  ReachTest reach;
  reach.r2max = masked ? p.r2max : FGVC_NO_LIMIT;
  reach.ry = masked ? p.ry : FGVC_NO_LIMIT;
  reach.rx = masked ? p.rx : FGVC_NO_LIMIT;
  const int TY0 = ty * (2 * QBH), TX0 = tx * (2 * QBW);
  const int QY0 = TY0 + (qb & 1) * QBH, QX0 = TX0 + (qb >> 1) * QBW;
  const int qy = QY0 + (n >> 3), qx = QX0 + (n & 7);

  // ---- prologue 1: the query rows of the four blocks through the ring (coalesced 1 KiB rows by LDS-DMA)
  if (role < 2) {
    const uint16_t* qbase = p.q_hl + (size_t)qf * p.Hq * p.Wq * (2 * C) + 8 * lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = par * 16 + i;                                  // row of query block qb
      const int y = imin(QY0 + (r >> 3), p.Hq - 1), x = imin(QX0 + (r & 7), p.Wq - 1);
      lds_dma_16(qbase + ((size_t)y * p.Wq + x) * (2 * C), &smem[qb * BUFB + r * LDB]);
    }
  }
  // ---- prologue 2 (overlaps the DMA): the list of key blocks this super-tile visits, row-major, then re-ordered
  //      first, last, second, second to last, ...: the entries only the upper (lower) query blocks reach sit at the head (tail) of
  //      the row-major list; alternating them keeps every consumer busy within the ring's depth
  if (wave == 0) {
    const int by_lo = imax(0, TY0 - imin(reach_y, TY0)) / QBH;
    const int by_hi = imin(p.Hk - 1, TY0 + 2 * QBH - 1 + imin(reach_y, p.Hk)) / QBH;
    const int bxl = imax(0, TX0 - imin(reach_x, TX0)) / QBW;
    const int bxh = imin(p.Wk - 1, TX0 + 2 * QBW - 1 + imin(reach_x, p.Wk)) / QBW;
    const int nbx = bxh - bxl + 1;
    const int nall = (by_hi - by_lo + 1) * nbx;
    // the host has checked that a MASKED pair's reach fits the list; an unmasked pair on a larger key grid than the list holds
    // (the caller promised there was none: `all_masked`) gets EMPTY lists (-1 / -inf), never truncated ones
    const int ncand = nall > V6_LIST_CAP ? 0 : nall;
    auto reach_bits = [&](int c) -> uint32_t {
      const int by = by_lo + c / nbx, bx = bxl + c % nbx;
      uint32_t m = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b)
        m |= (uint32_t)reach(TY0 + (b & 1) * QBH, TX0 + (b >> 1) * QBW, by * QBH, bx * QBW) << b;
      return m ? ((uint32_t)by | ((uint32_t)bx << 12) | (m << 24)) : 0u;
    };
    int total = 0;                                                   // pass 1: how many blocks are reached at all
    for (int base = 0; base < ncand; base += 64) {
      const int c = base + lane;
      total += __popcll(__ballot(c < ncand && reach_bits(c) != 0u));
    }
    int count = 0;                                                   // pass 2: row-major rank r -> position 2 r | 2 (total - 1 - r) + 1
    const int head = (total + 1) >> 1;
    for (int base = 0; base < ncand; base += 64) {
      const int c = base + lane;
      const uint32_t ent = c < ncand ? reach_bits(c) : 0u;
      const unsigned long long bal = __ballot(ent != 0u);
      if (ent) {
        const int r = count + __popcll(bal & ((1ull << lane) - 1));
        blist[(p.debug & 8192) ? r : (r < head ? 2 * r : 2 * (total - 1 - r) + 1)] = ent;      // debug & 8192: plain row-major order (A/B)
      }
      count += __popcll(bal);
    }
    if (lane == 0) blist_n = count;
  }
  if (tid < NSLOT) {
    filled[tid] = 0;
    done[tid] = 0;
    hand_full[tid] = 0;
    hand_free[tid] = 0;
  }
  if (tid == 0) wg_dead = (p.debug & 4096) ? 1 : 0;      // 4096: fault injection for the fail-closed test
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int n_steps = blist_n;
  bool dead = false;

  if (role == 2) {
    // =========================================== producer: pixel row qb of every key block ===========================================
    __syncthreads();                                                 // the consumers have read their query fragments: the ring is free
    const uint32_t lane16 = 16u * lane;
    const int n_total = g_count * n_steps;                             // key blocks of the whole run, numbered G = pair * n_steps + e
    int cur_pair = -1;
    const unsigned char* kbase = nullptr;
    auto stage = [&](int G) {
      const int pi = G / n_steps, e = G - pi * n_steps;
      if (pi != cur_pair) {                                            // wave-uniform
        cur_pair = pi;
        const int kf = p.pairs[g_start + pi].y;
        kbase = reinterpret_cast<const unsigned char*>(p.k_hl) + (size_t)kf * p.Hk * p.Wk * (4 * C);
      }
      const uint32_t ent = __builtin_amdgcn_readfirstlane(blist[e]);
      const int sby = ent & 0xfff, sbx = (ent >> 12) & 0xfff;
      const int ky = imin(sby * QBH + qb, p.Hk - 1), kx0 = sbx * QBW;
      const unsigned char* src = kbase + ((size_t)ky * p.Wk + kx0) * (4 * C) + lane16;
      const int xmax = p.Wk - 1 - kx0;                                 // >= 0: the block starts inside the frame
      unsigned char* dst = &smem[(G & (NSLOT - 1)) * BUFB + (qb * 8) * LDB];
#pragma unroll
      for (int i = 0; i < 8; ++i) lds_dma_16(src + (size_t)imin(i, xmax) * (4 * C), dst + i * LDB);
    };
    if (n_total > 0) {
      stage(0);
      for (int G = 0; G < n_total; ++G) {
        if (G + 1 < n_total) {
          const int s1 = (G + 1) & (NSLOT - 1), gen1 = (G + 1) / NSLOT;
          if (gen1 > 0) spin_ge<6, false>(&done[s1], 4 * gen1, dead, &wg_dead);   // block G + 1 - NSLOT released by all four consumers
          asm volatile("" ::: "memory");                                                // (slots free up a tile time apart: poll rarely)
          stage(G + 1);
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            // block G landed, block G + 1 in flight
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (lane == 0) __hip_atomic_fetch_add(&filled[G & (NSLOT - 1)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (dead) g_pair_v5_timeout = 1;
    return;
  }

  const int n_loop = n_steps;
  if (role == 0) {
    // =============================================== consumer: query block qb ===============================================
    // query fragments (MFMA B operand): lane (n, hi) holds channels 16 j + 8 hi .. + 7 of query n, h and l parts
    f16x8 qh[KS], ql[KS];
    {
      const unsigned char* qp = &smem[qb * BUFB + n * LDB + 16 * hi];
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        qh[j] = *reinterpret_cast<const f16x8*>(qp + 32 * j);
        ql[j] = *reinterpret_cast<const f16x8*>(qp + 2 * C + 32 * j);
      }
    }
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      asm volatile("" ::"v"(qh[j]));
      asm volatile("" ::"v"(ql[j]));
    }
    __syncthreads();                               // the ring is free for key blocks
    f32x16 acc;
    int t_con = 0;                                 // tiles handed over so far (over the whole run)
    int fnext = -1;                                // filled[] of the NEXT key block's slot, asked for between a chain and its hand-over
    int* const hw = &hand[qb][4 * lane];
    for (int pi = 0; pi < g_count; ++pi) {
      uint32_t ent_next = n_loop > 0 ? blist[0] : 0u;
      for (int e = 0; e < n_loop; ++e) {
        const uint32_t ent = __builtin_amdgcn_readfirstlane(ent_next);
        ent_next = e + 1 < n_loop ? blist[e + 1] : 0u;          // lands during this entry's work
        const int G = pi * n_steps + e;                          // the ring counts key blocks over the whole run
        const int slot = G & (NSLOT - 1), gen = G / NSLOT;
        const bool comp = ((ent >> (24 + qb)) & 1) != 0;
        // block G has landed; a consumer that does not reach it waits for this too before it releases the slot (see the two-role form)
        // (after a chain the counter has been asked for before the hand-over: normally the block is there and nothing is waited for)
        if (fnext < 4 * (gen + 1)) spin_ge<2, false>(&filled[slot], 4 * (gen + 1), dead, &wg_dead);
        fnext = -1;
        asm volatile("" ::: "memory");
        if (!comp) {
          if (lane == 0) __hip_atomic_fetch_add(&done[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          continue;
        }
        const unsigned char* ka = &smem[slot * BUFB + n * LDB + 16 * hi];
        // The chain: 48 MFMAs as volatile inline assembly (the compiler neither reorders nor pads them).  Key fragments in two-deep
        // rings: ah[j + 2] is read right after the second MFMA of step j (the last reader of ah[j]), al[j + 2] after the third --
        // four to five MFMAs ahead of their first use, which the micro-benchmark says is enough (38 cycles per MFMA beside a
        // vector partner and an LDS-DMA streamer), and eight registers fewer than the three-deep ring of the two-role form.
        // The fragment reads are inline assembly as well: as C++ loads the compiler hoisted one above the last reader of its
        // register, took a fifth fragment register and spilled a query fragment for it -- reloaded from scratch memory between the
        // first two MFMAs of every tile.  The LDS returns a wave's reads in order, so the waits are counted by hand: when a fragment
        // is needed, the three reads issued after it may still be in flight (the last step waits for everything).
        f16x8 ah[2], al[2];
        int peek = 0;
        const uint32_t ka_l = lds_addr_of(ka);
#define V6_READ(DST, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ka_l), "n"(OFF) : "memory")
        V6_READ(ah[0], 0); V6_READ(al[0], 2 * C); V6_READ(ah[1], 32); V6_READ(al[1], 2 * C + 32);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          if (j == KS - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
          if (j == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(ah[0]), "v"(qh[0]));
          else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah[j & 1]), "v"(qh[j]));
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah[j & 1]), "v"(ql[j]));
          if (j + 2 < KS) V6_READ(ah[j & 1], 32 * (j + 2));
          if (j == KS - 1)         // has the selector read the previous tile?  Asked before the last MFMA, needed ~60 cycles later
            asm volatile("ds_read_b32 %0, %1" : "=v"(peek) : "v"(lds_addr_of(&hand_free[qb])) : "memory");
          if (j < KS - 1) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(al[j & 1]), "v"(qh[j]));
          if (j + 2 < KS) V6_READ(al[j & 1], 2 * C + 32 * (j + 2));
          if (j == KS - 3) {
            // every LDS read of this block has been issued (the LDS executes a wave's operations in order): release the slot
            if (lane == 0) __hip_atomic_fetch_add(&done[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
          }
        }
#undef V6_READ
        // MFMA result -> vector read: the last MFMA's passes must have written back
        asm volatile("s_nop 15\n\ts_nop 7\n\ts_waitcnt lgkmcnt(0)" : "+v"(acc), "+v"(peek));     // (and the peek has landed)
        int peek2;                                     // is the next key block there?  The answer lands under the hand-over
        asm volatile("ds_read_b32 %0, %1" : "=v"(peek2) : "v"(lds_addr_of(&filled[(G + 1) & (NSLOT - 1)])) : "memory");
        // hand the tile over once the selector has read the one before (it normally has: the answer came with the last fragments)
        if (__builtin_amdgcn_readfirstlane(peek) < t_con) spin_ge<2, false>(&hand_free[qb], t_con, dead, &wg_dead);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          int4 v;
          v.x = (int)acc[4 * g4 + 0]; v.y = (int)acc[4 * g4 + 1]; v.z = (int)acc[4 * g4 + 2]; v.w = (int)acc[4 * g4 + 3];
          *reinterpret_cast<int4*>(hw + g4 * 256) = v;
        }
        asm volatile("" ::: "memory");                 // the LDS executes a wave's operations in order: the count follows the data
        if (lane == 0) __hip_atomic_fetch_add(&hand_full[qb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ++t_con;
        asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(peek2) : : "memory");   // the read is older than the four stores and the count
        fnext = __builtin_amdgcn_readfirstlane(peek2);
      }
    }
    if (dead) g_pair_v5_timeout = 1;
    return;
  }

  // =============================================== selector of query block qb ===============================================
  __syncthreads();                                 // (the consumers read their query fragments)
  int lk[K], lb[K];                                // running list, ASCENDING: lk[0] = K-th best ... lk[K-1] = best
  int ck[16];                                      // the tile at hand: fixed-point scores, turned into keys in place
  // inputs of the generated selection slices (pair_v5_chain.inc, part 2: the slices without a chain around them)
  int v_dy0 = 0, v_dx0 = 0, v_base = 0;
  int s_r2lim = -1;
  int v_empty = KEY_EMPTY;
  asm volatile("" : "+v"(v_empty));                // keep it in a register: the slices read it as an operand
  constexpr bool do_sel = true;
  (void)do_sel;
  auto select_tile = [&]() {
#define FGVC_V5_PART 2
    if constexpr (K == 10) {
#define FGVC_V5_K 10
#include "pair_v5_chain.inc"
#undef FGVC_V5_K
    } else {
#define FGVC_V5_K 5
#include "pair_v5_chain.inc"
#undef FGVC_V5_K
    }
#undef FGVC_V5_PART
  };
  int t_sel = 0;
  const int* const hr = &hand[qb][4 * lane];
  for (int pi = 0; pi < g_count; ++pi) {           // ---- the pairs of the run
#pragma unroll
  for (int j = 0; j < K; ++j) {
    lk[j] = KEY_EMPTY;
    lb[j] = -1;
  }
  for (int e = 0; e < n_loop; ++e) {
    const uint32_t ent = __builtin_amdgcn_readfirstlane(blist[e]);
    if (((ent >> (24 + qb)) & 1) == 0) continue;   // not a tile of this query block
    spin_ge<2, false>(&hand_full[qb], t_sel + 1, dead, &wg_dead);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int4 v = *reinterpret_cast<const int4*>(hr + g4 * 256);
      ck[4 * g4 + 0] = v.x; ck[4 * g4 + 1] = v.y; ck[4 * g4 + 2] = v.z; ck[4 * g4 + 3] = v.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(ck[r]));
    if (lane == 0) __hip_atomic_fetch_add(&hand_free[qb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ++t_sel;
    {
      const int ky0 = (int)(ent & 0xfff) * QBH, kx0 = (int)((ent >> 12) & 0xfff) * QBW;
      v_base = ky0 * p.Wk + kx0;
      const bool interior = ky0 + QBH <= p.Hk && kx0 + QBW <= p.Wk;
      const bool circle_only = reach.ry >= FGVC_NO_LIMIT && reach.rx >= FGVC_NO_LIMIT;
      if (interior && circle_only) {              // wave-uniform
        v_dy0 = ky0 - qy;
        v_dx0 = kx0 + 4 * hi - qx;
        s_r2lim = reach.r2max;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dy = ky0 - qy + (r >> 2), dx = kx0 + 4 * hi - qx + (r & 3);
          const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
          const int cy = imin(ady, 32767), cx = imin(adx, 32767);       // squares stay below 2^30 on any supported grid
          const bool ok = (ky0 + (r >> 2) < p.Hk) & (kx0 + 4 * hi + (r & 3) < p.Wk) &
                          (cy * cy + cx * cx <= reach.r2max) & (ady <= reach.ry) & (adx <= reach.rx);
          ck[r] = ok ? ck[r] : KEY_EMPTY;
        }
        v_dy0 = 0;
        v_dx0 = 0;
        s_r2lim = FGVC_NO_LIMIT;                  // (r & 3)^2 <= FGVC_NO_LIMIT - (r >> 2)^2: always kept
      }
    }
    if ((p.debug & 2048) == 0) select_tile();     // 2048: ablation (results wrong): hand-over only
  }

  // ---- epilogue: two partial lists per query (the two lane halves) -> canonical top-K.  Entries become 64-bit words
  //      (score_fx : ~pixel), larger = better (higher score, then LOWER pixel index)
  long long L[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int key = lk[i];
    const int r = 15 - (key & 15);
    const int pix = lb[i] + 4 * hi + (r >> 2) * p.Wk + (r & 3);
    const bool em = key < (int)0xC0000000;      // KEY_EMPTY, or KEY_EMPTY | register tag (a candidate the general predicate rejected)
    const uint32_t lo = em ? 0u : ~(uint32_t)pix;
    L[i] = (long long)(((unsigned long long)(uint32_t)(key & ~15) << 32) | lo);
  }
  // a lane's list is ascending in (score, tag); entries of different tiles with the same score may be out of pixel order
#define X(I, J)                                   \
  {                                               \
    const bool b_ = L[I] < L[J];                  \
    const long long lo_ = b_ ? L[I] : L[J];       \
    const long long hi_ = b_ ? L[J] : L[I];       \
    L[J] = lo_; L[I] = hi_;                       \
  }
  if constexpr (K == 10) { FGVC_SORTNET_10(X) }    // descending in I < J ...
  else { FGVC_SORTNET_5(X) }
#undef X
#pragma unroll
  for (int i = 0; i < K / 2; ++i) {                // ... so reverse: ascending like the running list
    const long long tmp = L[i];
    L[i] = L[K - 1 - i];
    L[K - 1 - i] = tmp;
  }
  {
    long long B[K];
#pragma unroll
    for (int i = 0; i < K; ++i) B[i] = __shfl_xor(L[i], 32);
#pragma unroll
    for (int i = 0; i < K; ++i) L[i] = L[i] > B[K - 1 - i] ? L[i] : B[K - 1 - i];
#define X(I, J)                                   \
    {                                             \
      const bool b_ = L[I] > L[J];                \
      const long long lo_ = b_ ? L[J] : L[I];     \
      const long long hi_ = b_ ? L[I] : L[J];     \
      L[I] = lo_; L[J] = hi_;                     \
    }
    if constexpr (K == 10) { FGVC_VMERGE_ASC_10(X) }
    else { FGVC_VMERGE_ASC_5(X) }
#undef X
  }
  const bool poison = dead || __builtin_amdgcn_readfirstlane(*(volatile int*)&wg_dead) != 0;
  if (poison) g_pair_v5_timeout = 1;
  if (hi == 0 && qy < p.Hq && qx < p.Wq) {
    const size_t oo = ((size_t)(g_start + pi) * p.Hq * p.Wq + (size_t)qy * p.Wq + qx) * p.kout;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (j < p.kout) {
        const long long v = L[K - 1 - j];
        const int sk = (int)(v >> 32);
        const bool em = sk < (int)0xC0000000;
        p.idx_out[oo + j] = poison ? 0 : em ? -1 : (int)~(uint32_t)v;
        p.score_out[oo + j] = poison ? PAIR_POISON_SCORE : em ? -INFINITY : (float)sk * 0x1p-28f;
      }
    }
  }
  }   // pairs of the run
  if (dead) g_pair_v5_timeout = 1;
}

static int g_pair_v5_debug = 0;
void set_pair_v5_debug(int v) { g_pair_v5_debug = v; }

int pair_v5_probe_read(long long* out32) {
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_pair_v5_probe), 32 * sizeof(long long)) == hipSuccess ? 0 : -1;
}

// read and clear: one event fails one check, not every later one of the process
int pair_v5_timeout_flag() {
  int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_pair_v5_timeout), sizeof(int)) != hipSuccess) return -1;
  if (v != 0) {
    const int zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_pair_v5_timeout), &zero, sizeof(int)) != hipSuccess) return -1;
  }
  return v;
}

int pair_topk_v5_launch(const uint16_t* q_hl, const uint16_t* k_hl, const int32_t* pairs, int n_pairs, int Hq, int Wq,
                        int Hk, int Wk, int r2max, int ry, int rx, int topk, int all_masked, const int32_t* groups, int n_groups,
                        int32_t* idx_out, float* score_out, hipStream_t s) {
  PairParamsB p;
  p.q_hl = q_hl; p.k_hl = k_hl; p.pairs = reinterpret_cast<const int4*>(pairs);
  p.Hq = Hq; p.Wq = Wq; p.Hk = Hk; p.Wk = Wk;
  p.r2max = r2max; p.ry = ry; p.rx = rx;
  int rr = 0;  // floor(sqrt(r2max)) in integers
  while (rr < 46340 && (long long)(rr + 1) * (rr + 1) <= (long long)r2max) ++rr;
  p.reach_y = imin(ry, rr); p.reach_x = imin(rx, rr);
  p.kout = topk;
  p.n_ty = cdiv(Hq, 2 * QBH); p.n_tx = cdiv(Wq, 2 * QBW);
  p.idx_out = idx_out; p.score_out = score_out;
  p.groups = reinterpret_cast<const int2*>(groups);
  p.debug = g_pair_v5_debug;
  long long need_blocks = 0;
  {  // the per-workgroup block list must hold every key block a super-tile can reach: the mask's reach for a masked pair, the whole key grid for a pair without FGVC_PAIR_MASKED
    const long long nby = imin(cdiv(Hk, QBH), (2 * QBH - 1 + 2 * (long long)imin(p.reach_y, Hk)) / QBH + 2);
    const long long nbx = imin(cdiv(Wk, QBW), (2 * QBW - 1 + 2 * (long long)imin(p.reach_x, Wk)) / QBW + 2);
    const long long whole = (long long)cdiv(Hk, QBH) * cdiv(Wk, QBW);
    const long long need = all_masked ? nby * nbx : whole;
    need_blocks = need;
    if (need > PAIR_LIST_CAP || Hk >= 4096 * QBH || Wk >= 4096 * QBW) {
      set_error("fgvc_pair_topk_f16x3: key grid %dx%d needs %lld > %d key blocks per query tile (%s); use fgvc_pair_topk_f32",
                Hk, Wk, need, PAIR_LIST_CAP, all_masked ? "mask reach" : "a pair without FGVC_PAIR_MASKED scans the frame");
      return FGVC_ERR_UNSUPPORTED;
    }
  }
  dim3 grid(p.n_ty * p.n_tx, groups ? n_groups : n_pairs);
  // default: the three-role form; pair_f16_debug & 1024, any ablation bit, or a block list beyond its (smaller) capacity: the two-role form
  const bool three_roles = (g_pair_v5_debug & (1024 | 511)) == 0 && need_blocks <= V6_LIST_CAP;   // (2048: three-role ablation)
  if (three_roles) {
    if (topk <= 5) pair_topk_kernel_v6<5><<<grid, 768, 0, s>>>(p);
    else pair_topk_kernel_v6<10><<<grid, 768, 0, s>>>(p);
  } else if (topk <= 5) pair_topk_kernel_v5<5, 0><<<grid, 512, 0, s>>>(p);
  else switch (g_pair_v5_debug & 3) {
    case 0: pair_topk_kernel_v5<10, 0><<<grid, 512, 0, s>>>(p); break;
    case 1: pair_topk_kernel_v5<10, 1><<<grid, 512, 0, s>>>(p); break;
    case 2: pair_topk_kernel_v5<10, 2><<<grid, 512, 0, s>>>(p); break;
    default: pair_topk_kernel_v5<10, 3><<<grid, 512, 0, s>>>(p); break;
  }
  FGVC_CHECK_LAUNCH("fgvc_pair_topk_f16x3");
  return FGVC_OK;
}

}  // namespace fgvc

#include "pair_topk_v7.hpp"
#include "pair_topk_v8.hpp"      // round 6: the same arithmetic as ONE kind of wave, a key block staged once for eight query blocks      // fgvc_pair_topk_f16f6: the same protocol at 1.5 pipe units (shares the helpers above)
