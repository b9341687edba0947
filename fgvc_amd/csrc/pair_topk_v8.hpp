// pair_topk_kernel_v8 (round 6, OPT-IN: pair_f16_debug & 4194304): fgvc_pair_topk_f16f6's windowed correlation + top-k as ONE kind of
// wave.  Included by pair_topk_v5.hip behind pair_topk_v7.hpp (same translation unit: row format, bounded spins, timeout flag, poison lists).
//
// What round 5's profile said about pair_topk_kernel_v7 (27 pairs of a 480p clip, 1.17 ms, matrix pipe 24.5 % busy, 3.59 GB fetched): a
// 32 x 32 tile is a chain of LDS round trips between three wave roles -- a consumer multiplies and hands its 4 KiB of sums to a selector
// through the LDS, producers stage a 30 KiB key block through registers for FOUR query blocks (56 block loads per 8 x 16 query tile).
// This kernel is the review's proposal built and measured:
//   * 8 waves, two per SIMD, all alike.  A wave owns one 4 x 8 query block of a 16 x 16 query tile with ALL its operands resident (h 64 +
//     l6 24 + h6 24 registers), multiplies it with a key block as the LDS holds it (33 operand reads per tile; v7: 43 + the query's h6
//     from the LDS), keeps the sums and selects from them itself: no hand-over, no mailbox.
//   * A tile = ONE assembly statement (pair_v8.inc, tools/gen_pair_v8.py): the 24 matrix instructions of v7's chain -- same operands,
//     same order: the scores are v7's bit for bit -- with the selection of the tile BEFORE it (212 vector operations) dealt behind them,
//     the chain's buffers physical registers named in the text, the selection's values renamed in place by the generator's allocator
//     (statement per instruction, the allocator rotated the 6-register operands through v_mov chains and spilled: 270 registers wanted).
//   * A key block is staged ONCE for eight query blocks (68 block loads per 16 x 16 tile: 8.5 per query block, v7: 14) by LDS-DMA
//     (global_load_lds_dwordx4 under an exec mask of 58 lanes = the 928 bytes of a row that carry something: tools/micro/
//     probe_dma_exec.hip; four instructions per row, m0 and vcc the only registers), all 32 rows of block G by wave G mod 8, three
//     blocks ahead; posted two steps later.
//   * The ring has no workgroup barrier: per slot `filled` counts posted blocks, `done` the waves that have finished with one (also those
//     whose query block does not reach it -- once it is complete: a release for a block that is not staged yet would be counted for the
//     block before it).  The block list pairs a key row only the upper query blocks reach with one only the lower reach, step by step
//     (row-major: 1.38 ms instead of 1.20).  Every spin is bounded; a wave that gives up raises the workgroup's flag and the lists
//     written from then on are POISON, as in v7.
//   * Selection keys are canonical: (22-bit score << 10) | tag, tag = the key pixel's rank in (row, column) order inside the window the
//     query block can reach, so equal scores resolve to the LOWER pixel index whatever the order the blocks were visited in.
// Measured (profiles/r06_pair_v8_cfg2.log, r06_pmc.json; same box as v7): bit-identical scores on 13 M lists, identical lists on every one
// of them -- and 1-7 % SLOWER than v7 at every BASELINE shape (1.20 against 1.18 ms at 480p; cfg4 10.5 / 9.8, cfg5 13.2 / 12.6), matrix
// pipe 21 % busy, 3.9 GB of L2 fills (fewer bytes reach each CU, but the tile order of this kernel hits its L2 only 42 % of the time:
// v7's was tuned to 65 %).  Why the proposal does not pay here, from the ablations: with key blocks, chain and selection all switched
// off the launch still takes 0.52 of its 1.20 ms -- a SIMD issues ONE instruction per ~4 cycles whatever number of its waves is ready
// (the protocol of a step is ~150 instructions per wave), the selection is 212 + 32 vector instructions per tile that NO arrangement
// makes fewer, so a tile costs >= 1 100 issue cycles per SIMD against 768 of matrix pipe: both kernels sit at 2 400-2 500 cycles per
// tile and SIMD, bound by instruction issue, not by the pipe, the LDS or the bytes.  What would move them is fewer instructions per
// candidate, not another arrangement of the same ones.  v7 stays the default; this kernel is kept, tested against
// v7 on every run of the GPU suite, as the measured answer to the proposal.
// Limits: those of v7 (C = 256, k <= 10, normalised rows, every pair masked) and a mask reach of at most 16 pixels either way (the tag).
#pragma once

namespace fgvc {

constexpr int V8_LDB = P6_END + 16;            // 944 B: 236 dwords = 44 mod 64 -> conflict-free b128 reads by lanes (n, hi)
constexpr int V8_BUFB = 32 * V8_LDB;
constexpr int V8_NSLOT = 5;
constexpr int V8_D = 3;                        // blocks a wave keeps in flight ahead of the one it consumes
constexpr int V8_MAX_BLOCKS = 80;              // a 16 x 16 query tile reaches at most 12 x 6 blocks at reach 16
constexpr int V8_REACH = 16;
constexpr int V8_DMA_LANES = P6_END / 16;      // 58


template <int K, bool PROBE>
__global__ __launch_bounds__(512) void pair_topk_kernel_v8(PairParamsB p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[V8_NSLOT * V8_BUFB];
  __shared__ uint32_t blist[V8_MAX_BLOCKS];      // by | bx << 12 | (query blocks that reach it) << 24
  __shared__ int blist_n;
  __shared__ int counters[2 * V8_NSLOT];           // filled[s] = counters[s]: waves whose rows of the slot's block have landed; done[s] = counters[NSLOT + s]: waves that have finished with it
  int* const filled = counters;
  int* const done = counters + V8_NSLOT;
  __shared__ int wg_dead;
  __shared__ int pipe_lock[4];                     // one per SIMD: the two waves of a SIMD take turns on its matrix pipe (see try_pipe below)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, hi = lane >> 5;
  // waves w and w + 4 share a SIMD: query blocks (br, 0) and (br, 1), which reach the same key rows
  const int br = wave & 3, bc = wave >> 2;

  int g_start = blockIdx.y, g_count = 1;
  if (p.groups) {
    const int2 gr = p.groups[blockIdx.y];
    g_start = gr.x;
    g_count = gr.y;
  }
  // work order: the runs come longest first (blockIdx.y); within a run the tiles of an XCD (linear index mod 8) are a contiguous range
  // in column-major order -- vertical neighbours share two thirds of their key rows and meet in one L2
  const int ntile = p.n_ty * p.n_tx;
  int tile = blockIdx.x;
  if (!(p.debug & 512)) {
    const int per = (ntile + 7) >> 3;
    const int t2 = (blockIdx.x & 7) * per + (blockIdx.x >> 3);      // bijection of [0, 8 per) onto itself; indices >= ntile idle
    tile = t2;
  }
  if (tile >= ntile) return;                                         // (whole workgroup: before any barrier)
  const int tx = tile / p.n_ty, ty = tile - tx * p.n_ty;             // column-major
  const int4 pr = p.pairs[g_start];
  const int qf = pr.x;
  ReachTest reach;
  reach.r2max = p.r2max; reach.ry = p.ry; reach.rx = p.rx;
  const int TY0 = ty * 16, TX0 = tx * 16;
  const int QY0 = TY0 + br * QBH, QX0 = TX0 + bc * QBW;
  const int qy = QY0 + (n >> 3), qx = QX0 + (n & 7);

  // ---- prologue 1: the query block's operands straight from the bank (a lane reads the 16-byte pieces of its own row: 30 KiB per wave,
  //      once per run)
  f16x8 qhs[16];                                           // the query's f16 fragments (fragment j = channels 16 j + 8 hi ..+ 8)
  i32x6 q6l[4], q6h[4];
  int sqH, sqL;
  {
    const unsigned char* qp = reinterpret_cast<const unsigned char*>(p.q_hl) + (size_t)qf * p.Hq * p.Wq * p.rowb +
                              ((size_t)imin(qy, p.Hq - 1) * p.Wq + imin(qx, p.Wq - 1)) * p.rowb + 16 * hi;
#pragma unroll
    for (int j = 0; j < 16; ++j) qhs[j] = *reinterpret_cast<const f16x8*>(qp + 32 * j);
#pragma unroll
    for (int v = 0; v < 4; ++v)
      q6l[v] = v7_cat6(*reinterpret_cast<const i32x4v*>(qp + P6_L6M + 32 * v),
                       *reinterpret_cast<const i32x2v*>(qp + P6_L6T + 32 * (v >> 1) + 8 * (v & 1)));
    const i32x2v sc = *reinterpret_cast<const i32x2v*>(qp + P6_SC);
    sqH = sc[0];
    sqL = sc[1];
    // the query's own h6 operand, exactly as v7 makes it (v_cvt_scalef32_pk32_fp6_f16 of the f16 fragments of a 64-channel group;
    // 2^sh = (scale byte + 4) << 23)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      f16x32 g;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 8; ++i) g[8 * m + i] = qhs[4 * v + m][i];
      u32x6 h6;
      const unsigned int sc_bits = (unsigned int)(((sqH >> (8 * v)) & 255) + 4) << 23;
      asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(h6) : "v"(g), "v"(sc_bits));
      q6h[v] = i32x6{(int)h6[0], (int)h6[1], (int)h6[2], (int)h6[3], (int)h6[4], (int)h6[5]};
    }
  }
  // ---- prologue 2: the key blocks this tile visits
  if (wave == 0) {
    const int ryy = imin(p.reach_y, V8_REACH), rxx = imin(p.reach_x, V8_REACH);
    const int by_lo = imax(0, TY0 - ryy) / QBH, by_hi = imin(p.Hk - 1, TY0 + 15 + ryy) / QBH;
    const int bxl = imax(0, TX0 - rxx) / QBW, bxh = imin(p.Wk - 1, TX0 + 15 + rxx) / QBW;
    const int nbx = bxh - bxl + 1, nby = by_hi - by_lo + 1;
    const int nall = nby * nbx;
    // Order: the first E block rows are reached by the upper query blocks only, the last E by the lower only -- row a is dealt column by
    // column against row nby - E + a, so that every pair of consecutive steps has work for every SIMD; then the rows in between.
    // (debug & 1024: plain row-major.)
    const int E = (p.debug & 1024) ? 0 : imin(3, nby >> 1);
    auto entry = [&](int q) -> uint32_t {
      int rho, c;
      if (q < 2 * E * nbx) {
        const int a = q / (2 * nbx), rem = q - a * 2 * nbx;
        c = rem >> 1;
        rho = (rem & 1) ? nby - E + a : a;
      } else {
        const int q2 = q - 2 * E * nbx;
        const int a = q2 / nbx;
        rho = E + a;
        c = q2 - a * nbx;
      }
      const int by = by_lo + rho, bx = bxl + c;
      uint32_t m = 0;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int wy0 = TY0 + (b & 3) * QBH, wx0 = TX0 + (b >> 2) * QBW;
        m |= (uint32_t)(wy0 < p.Hq && wx0 < p.Wq && reach(wy0, wx0, by * QBH, bx * QBW)) << b;
      }
      return m ? ((uint32_t)by | ((uint32_t)bx << 12) | (m << 24)) : 0u;
    };
    int count = 0;
    for (int base = 0; base < nall; base += 64) {
      const int q = base + lane;
      const uint32_t ent = q < nall ? entry(q) : 0u;
      const unsigned long long bal = __ballot(ent != 0u);
      const int r = count + __popcll(bal & ((1ull << lane) - 1));
      if (ent && r < V8_MAX_BLOCKS) blist[r] = ent;
      count += __popcll(bal);
    }
    // (the host has checked the reach: at most 72 blocks; should a launch get here with more anyway the pair gets EMPTY lists)
    if (lane == 0) blist_n = (count > V8_MAX_BLOCKS || (p.debug & 524288)) ? 0 : count;
  }
  if (tid < V8_NSLOT) {
    filled[tid] = 0;
    done[tid] = 0;
  }
  if (tid == 0) wg_dead = (p.debug & 4096) ? 1 : 0;      // 4096: fault injection for the fail-closed test
  if (tid < 4) pipe_lock[tid] = 0;
  __syncthreads();                                        // the only barrier of the kernel
  const int n_steps = blist_n;
  const int n_total = g_count * n_steps;
  bool dead = false;

  const uint32_t smem_l = lds_addr_of(smem);
  const uint32_t lane16 = 16u * lane;
  const uint32_t lane_off = (uint32_t)(n * V8_LDB + 16 * hi);
  const bool circle_only = reach.ry >= FGVC_NO_LIMIT && reach.rx >= FGVC_NO_LIMIT;

  // A wave alone issues one instruction per ~4 cycles, so what the protocol costs is its instruction count: the block list, the byte
  // offset of this wave's first row of every block and the key frames of the run live in LANES (v_readlane with a scalar index: no LDS
  // round trip, no address arithmetic per step); the two counters of a step are read by one wait; a count is one ds_add under exec = 1.
  const uint32_t blv0 = blist[imin(lane, V8_MAX_BLOCKS - 1)], blv1 = blist[imin(64 + lane, V8_MAX_BLOCKS - 1)];
  const int kfv = p.pairs[g_start + imin(lane, g_count - 1)].y;
  const size_t frame_b = (size_t)p.Hk * p.Wk * p.rowb;
  const uint32_t cnt_l = lds_addr_of(counters);            // filled[s] at + 4 s, done[s] at + 4 (NSLOT + s)
  const uint32_t v_one = 1u;
  auto count_up = [&](uint32_t byte_off) {                 // one LDS add by one lane (every lane of the wave is active here)
    const uint32_t a = cnt_l + byte_off;
    asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(a), "v"(v_one) : "memory");
  };
  auto peek = [&](uint32_t byte_off) -> int {              // a counter, now (LDS round trip)
    const uint32_t a = cnt_l + byte_off;
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
  };

  // The two waves of a SIMD run the same program in step (the ring keeps every wave within a block of the others): left alone they
  // multiply together -- one matrix pipe: each chain takes twice its time -- and then do everything else together with the pipe idle.
  // debug & 128: a lock per SIMD serialises the chains, to put the partner's keys, counters and staging UNDER a chain (measured: 3 % slower).
  // Bounded: a wave that does not get the lock goes ahead without it (a matter of speed only).
  const uint32_t lock_l = lds_addr_of(&pipe_lock[wave & 3]);
  const uint32_t v_zero = 0u;
  auto pipe_acquire = [&]() {
    for (int it = 0; it < 4096; ++it) {
      int old;
      asm volatile("s_mov_b64 exec, 1\n\tds_cmpst_rtn_b32 %0, %1, %2, %3\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(lock_l), "v"(v_zero), "v"(v_one) : "memory");
      if (__builtin_amdgcn_readfirstlane(old) == 0) return;
      __builtin_amdgcn_s_sleep(2);
    }
  };
  auto pipe_release = [&]() { asm volatile("s_mov_b64 exec, 1\n\tds_write_b32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(lock_l), "v"(v_zero) : "memory"); };

  // ---- producer side.  Block G is staged by wave G mod 8, all 32 rows of it (one issue sequence and one count per block instead of
  // eight: what a step costs every wave is a comparison).  A wave issues block G + D at step G and posts it two steps later (its rows
  // have landed by then: the wait is normally free), a step before anybody reads it.
  int Gp = 0, p_pi = 0, p_e = 0, p_slot = 0, p_gen = 0;    // cursor over the blocks in issue order (advanced by every wave, used by the one whose turn it is)
  int mine = -1, mine_slot = 0;                            // the block this wave has in flight
  auto stage = [&]() {                                     // the 32 rows of block Gp -> slot p_slot
    if (p_gen > 0 && peek(4u * (V8_NSLOT + p_slot)) < 8 * p_gen)
      spin_ge<1, false>(&done[p_slot], 8 * p_gen, dead, &wg_dead);                       // block Gp - NSLOT released by all eight waves
    asm volatile("" ::: "memory");
    const int kf = p_pi < 64 ? __builtin_amdgcn_readlane(kfv, p_pi) : __builtin_amdgcn_readfirstlane(p.pairs[g_start + p_pi].y);
    const unsigned char* kbase = reinterpret_cast<const unsigned char*>(p.k_hl) + (size_t)kf * frame_b;
    const uint32_t ent = __builtin_amdgcn_readlane(p_e < 64 ? blv0 : blv1, p_e & 63);
    const int ky0 = (int)(ent & 0xfff) * QBH, kx0 = (int)((ent >> 12) & 0xfff) * QBW;
    uint32_t dst = smem_l + (uint32_t)(p_slot * V8_BUFB);
    if (!(p.debug & 1)) {                                  // (1: ablation, results wrong: the ring's counters only, no bytes moved)
      if (ky0 + QBH <= p.Hk && kx0 + QBW <= p.Wk) {        // all 32 pixels inside the frame: four instructions per row (pair_v8.inc)
        const unsigned char* src = kbase + ((size_t)ky0 * p.Wk + kx0) * p.rowb;
        const int s_rowb = p.rowb, s_rowskip = (p.Wk - (QBW - 1)) * p.rowb;
#define FGVC_V8_PART 4
#include "pair_v8.inc"
#undef FGVC_V8_PART
      } else {                                             // the frame's edge: rows and columns beyond it repeat the last one (masked by the predicate)
#pragma unroll 1
        for (int pr = 0; pr < QBH; ++pr) {
          const unsigned char* rowp = kbase + ((size_t)imin(ky0 + pr, p.Hk - 1) * p.Wk) * p.rowb;
#pragma unroll 1
          for (int c = 0; c < QBW; ++c) {
            const unsigned char* src = rowp + (size_t)imin(kx0 + c, p.Wk - 1) * p.rowb;
            asm volatile("s_mov_b32 exec_hi, 0x3ffffff\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b32 exec_hi, -1"
                         ::"v"(lane16), "s"(src), "s"(dst) : "memory");
            dst += V8_LDB;
          }
        }
      }
    }
    mine = Gp;
    mine_slot = p_slot;
  };
  auto advance_p = [&]() {
    ++Gp;
    if (++p_e == n_steps) { p_e = 0; ++p_pi; }
    if (++p_slot == V8_NSLOT) { p_slot = 0; ++p_gen; }
  };
  auto post_mine = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    count_up(4u * mine_slot);
    mine = -1;
  };

  unsigned int lk[K];                              // running list, ASCENDING: lk[0] = K-th best ... lk[K-1] = best; 0 = empty
  unsigned int ck[16];                             // the keys of the tile whose selection is pending (all 0: none)
#pragma unroll
  for (int j = 0; j < K; ++j) lk[j] = 0u;
#pragma unroll
  for (int r = 0; r < 16; ++r) ck[r] = 0u;
  int v_dy0 = 0, v_dx0 = 0;
  int s_r2lim = FGVC_NO_LIMIT;
  bool pending = false;

  const bool probe = PROBE && blockIdx.x == 40 && blockIdx.y == 0;
  long long pr_t0 = probe ? __builtin_amdgcn_s_memtime() : 0, pr_fill = 0, pr_chain = 0, pr_sel = 0, pr_issue = 0;
  int pr_tiles = 0;

  // vmcnt(0), as the BUILTIN: the compiler's wait-count pass must know that the query operands have landed -- it does not see the DMAs
  // of the loop, and put `s_waitcnt vmcnt(0)` (= every row this wave has in flight) in front of every tile's first use of them
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (int i = 0; i < V8_D; ++i) {
    if (Gp < n_total && (Gp & 7) == wave) stage();
    if (Gp < n_total) advance_p();
  }
  // The two waves of a SIMD run the same program: started together they multiply together (one matrix pipe) and select together (one
  // vector port).  debug & 2048: waves 4-7 start half a tile late (measured: the ring re-aligns them within a few blocks; no gain).
  if (wave >= 4 && (p.debug & 2048)) __builtin_amdgcn_s_sleep(12);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  int c_e = 0, c_pi = 0, c_slot = 0, c_gen = 0;
  const bool my_valid = QY0 < p.Hq && QX0 < p.Wq;
  for (int G = 0; G < n_total; ++G) {
    // ---- produce: post the block that is read next step, stage block G + D when it is this wave's turn
    {
      const long long i0 = probe ? __builtin_amdgcn_s_memtime() : 0;
      if (mine >= 0 && G + ((p.debug & 64) ? 1 : 2) >= mine) post_mine();      // (64: a step later)
      if (Gp < n_total) {
        if ((Gp & 7) == wave) stage();
        advance_p();
      }
      if (probe) pr_issue += __builtin_amdgcn_s_memtime() - i0;
    }
    const uint32_t ent = __builtin_amdgcn_readlane(c_e < 64 ? blv0 : blv1, c_e & 63);
    // block G complete?  (Also asked by a wave that does not read it: a release must never be sent for a block that is not staged yet --
    // the counters are per slot, and a wave five blocks ahead would be counted for the block before.)
    const long long f0 = probe ? __builtin_amdgcn_s_memtime() : 0;
    if (peek(4u * c_slot) < c_gen + 1) {
      spin_ge<1, false>(&filled[c_slot], c_gen + 1, dead, &wg_dead);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (the hand-counted waits of the chain count its own reads only)
    }
    if (((ent >> (24 + wave)) & 1u) != 0u) {
      // ---- consume: this wave's query block x key block G
      const long long c0 = probe ? __builtin_amdgcn_s_memtime() : 0;
      if (probe) pr_fill += c0 - f0;
      const uint32_t ka = smem_l + (uint32_t)(c_slot * V8_BUFB) + lane_off;
      if (pending && (p.debug & (2 | 8))) {                    // (8: selection in front of the chain instead of inside it; 2: ablation, results wrong: no matrix chain)
#define FGVC_V8_PART 3
        if constexpr (K == 10) {
#define FGVC_V8_K 10
#include "pair_v8.inc"
#undef FGVC_V8_K
        } else {
#define FGVC_V8_K 5
#include "pair_v8.inc"
#undef FGVC_V8_K
        }
#undef FGVC_V8_PART
        pending = false;
      }
      if (!(p.debug & 2)) {
        if (p.debug & 128) pipe_acquire();
        if (pending) {
#define FGVC_V8_PART 2
          if constexpr (K == 10) {
#define FGVC_V8_K 10
#include "pair_v8.inc"
#undef FGVC_V8_K
          } else {
#define FGVC_V8_K 5
#include "pair_v8.inc"
#undef FGVC_V8_K
          }
#undef FGVC_V8_PART
          pending = false;
        } else {
#define FGVC_V8_PART 1
#include "pair_v8.inc"
#undef FGVC_V8_PART
        }
        if (p.debug & 128) pipe_release();
      }
      // every LDS read of the block has returned: release the slot
      count_up(4u * (V8_NSLOT + c_slot));
      const long long s0 = probe ? __builtin_amdgcn_s_memtime() : 0;
      if (probe) { pr_chain += s0 - c0; ++pr_tiles; }
      // ---- the tile's keys: (bits(acc + 2^19 + 2^17) << 10) | tag, tag = 4 ((35 - 4 j' - row) 5 + 4 - i') + 3 - column
      {
        const int ky0 = (int)(ent & 0xfff) * QBH, kx0 = (int)((ent >> 12) & 0xfff) * QBW;
        const int jp = (ky0 - QY0) / QBH + 4, ip = (kx0 - QX0) / QBW + 2;      // 0..8, 0..4 (exact divisions)
        const int tagbase = 4 * ((35 - 4 * jp) * 5 + 4 - ip) + 3;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ck[r] = (__builtin_bit_cast(unsigned int, acc[r] + V7_BIAS) << 10) | (unsigned int)(tagbase - (20 * (r >> 2) + (r & 3)));
        const bool interior = ky0 + QBH <= p.Hk && kx0 + QBW <= p.Wk;
        if (interior && circle_only) {              // wave-uniform
          v_dy0 = ky0 - qy;
          v_dx0 = kx0 + 4 * hi - qx;
          s_r2lim = reach.r2max;
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dy = ky0 - qy + (r >> 2), dx = kx0 + 4 * hi - qx + (r & 3);
            const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
            const int cy = imin(ady, 32767), cx = imin(adx, 32767);
            const bool ok = (ky0 + (r >> 2) < p.Hk) & (kx0 + 4 * hi + (r & 3) < p.Wk) &
                            (cy * cy + cx * cx <= reach.r2max) & (ady <= reach.ry) & (adx <= reach.rx);
            ck[r] = ok ? ck[r] : 0u;
          }
          v_dy0 = 0;
          v_dx0 = 0;
          s_r2lim = FGVC_NO_LIMIT;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(ck[r]));
      }
      pending = !(p.debug & 1048576);                // (1048576: ablation, results wrong: no selection at all)
      if (probe) pr_sel += __builtin_amdgcn_s_memtime() - s0;
    } else {
      // not this wave's: released as soon as it is complete
      count_up(4u * (V8_NSLOT + c_slot));
    }
    if (++c_slot == V8_NSLOT) { c_slot = 0; ++c_gen; }
    if (++c_e == n_steps) {
      // ---- end of a pair: the pending tile, then two partial lists per query (the two lane halves) -> canonical top-K
      c_e = 0;
      if (pending) {
        const long long s0 = probe ? __builtin_amdgcn_s_memtime() : 0;
#define FGVC_V8_PART 3
        if constexpr (K == 10) {
#define FGVC_V8_K 10
#include "pair_v8.inc"
#undef FGVC_V8_K
        } else {
#define FGVC_V8_K 5
#include "pair_v8.inc"
#undef FGVC_V8_K
        }
#undef FGVC_V8_PART
        pending = false;
        if (probe) pr_sel += __builtin_amdgcn_s_memtime() - s0;
      }
      long long L[K];
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const unsigned int key = lk[i];
        const int tag = (int)(key & 1023u);
        const int T = tag >> 2, col = 3 - (tag & 3);
        const int q5 = (T * 52429) >> 18;            // T / 5 for T < 2^14
        const int ip = 4 - (T - 5 * q5);
        const int ky = QY0 - 16 + (35 - q5), kx = QX0 - 16 + 8 * ip + 4 * hi + col;
        const int pix = ky * p.Wk + kx;
        const bool em = key == 0u;
        L[i] = em ? 0ll : (long long)(((unsigned long long)(key >> 10) << 32) | (unsigned long long)(~(uint32_t)pix));
      }
      // a lane's list is ascending in (score, tag) = (score, -pixel): already canonical; the other lane half holds the other columns
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
      if (hi == 0 && my_valid && qy < p.Hq && qx < p.Wq) {
        const size_t oo = ((size_t)(g_start + c_pi) * p.Hq * p.Wq + (size_t)qy * p.Wq + qx) * p.kout;
#pragma unroll
        for (int j = 0; j < K; ++j) {
          if (j < p.kout) {
            const long long v = L[K - 1 - j];
            const int sk = (int)(v >> 32);              // 22-bit score: (cos + 2) 2^20
            const bool em = sk == 0;
            p.idx_out[oo + j] = poison ? 0 : em ? -1 : (int)~(uint32_t)v;
            p.score_out[oo + j] = poison ? PAIR_POISON_SCORE : em ? -INFINITY : (float)sk * 0x1p-20f - 2.0f;
          }
        }
      }
      ++c_pi;
#pragma unroll
      for (int j = 0; j < K; ++j) lk[j] = 0u;
    }
  }
  if (n_steps == 0 && hi == 0 && my_valid && qy < p.Hq && qx < p.Wq) {
    // (no key block at all -- a launch beyond the host's checks: EMPTY lists for every pair of the run)
    for (int pi = 0; pi < g_count; ++pi) {
      const size_t oo = ((size_t)(g_start + pi) * p.Hq * p.Wq + (size_t)qy * p.Wq + qx) * p.kout;
      for (int j = 0; j < p.kout; ++j) {
        p.idx_out[oo + j] = -1;
        p.score_out[oo + j] = -INFINITY;
      }
    }
  }
  if (probe && lane == 0 && wave < 4) {      // fgvc_pair_topk_f16x3_probe reads the 32 words (tools/experiments/time_pair_v8.py)
    long long* o = &g_pair_v5_probe[8 * wave];
    o[0] = __builtin_amdgcn_s_memtime() - pr_t0; o[1] = pr_fill; o[2] = pr_chain; o[3] = pr_sel; o[4] = pr_issue; o[5] = pr_tiles; o[6] = n_steps; o[7] = g_count;
  }
  if (dead) g_pair_v5_timeout = 1;
}

// whether pair_topk_kernel_v8 takes a launch: v7's conditions (checked by the caller) and a reach of at most 16 pixels either way
static bool pair_v8_applies(int reach_y, int reach_x) { return reach_y <= V8_REACH && reach_x <= V8_REACH; }

int pair_topk_v8_launch(const PairParamsB& p0, int n_pairs, int n_groups, int topk, hipStream_t s) {
  PairParamsB p = p0;
  p.n_ty = cdiv(p.Hq, 16);
  p.n_tx = cdiv(p.Wq, 16);
  const int ntile = p.n_ty * p.n_tx;
  dim3 grid(((ntile + 7) / 8) * 8, p.groups ? n_groups : n_pairs);
  if (topk <= 5) pair_topk_kernel_v8<5, false><<<grid, 512, 0, s>>>(p);
  else if (p.debug & 256) pair_topk_kernel_v8<10, true><<<grid, 512, 0, s>>>(p);
  else pair_topk_kernel_v8<10, false><<<grid, 512, 0, s>>>(p);
  FGVC_CHECK_LAUNCH("fgvc_pair_topk_f16f6");
  return FGVC_OK;
}

}  // namespace fgvc
