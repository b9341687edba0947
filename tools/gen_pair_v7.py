"""Generate fgvc_amd/csrc/pair_v7.inc for fgvc_pair_topk_f16f6 (pair_topk_v7.hip):

  PART 1  the consumer's matrix chain of one 32 x 32 tile at 1.5 pipe units: 16 v_mfma_f32_32x32x16_f16 (h_k h_q) + 8
          v_mfma_scale_f32_32x32x64_f8f6f4 on FP6 operands (h6_k l6_q and l6_k h6_q per 64-channel group), every LDS read, every wait
          and the on-the-fly FP6 form of the query's h part (v_cvt_scalef32_pk32_fp6_f16) as volatile assembly statements in a fixed
          order.  The LDS returns a wave's reads in order, so a wait before an instruction = the number of reads issued after the
          youngest one it needs; this script counts them (and fails if a register would be overwritten while a reader is pending).
  PART 2  the selector's stream per tile (K = 5, 10): disc predicate, 60-comparator selection network on 32-bit unsigned keys, merge
          into the running list.  Keys carry (score : 22 bits | 63 - list position : 6 | 15 - register : 4), so the list has no payload:
          a compare-exchange is v_max_u32 + v_min_u32 and the merge 10 + 30 operations (the f16x3 kernel's: 30 + 75).  212 vector
          operations per tile (f16x3: 293).  List-scheduled and executed in Python against a direct evaluation before it is written
          (the machinery of tools/gen_pair_v5_chain.py).

    python tools/gen_pair_v7.py        # rewrites the .inc; the build does not run it
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_pair_v5_chain as g5          # noqa: E402  (net, versioned, schedule, render_slice, declarations, cut, live_outs, cvar)

OUT = os.path.join(os.path.dirname(HERE), "fgvc_amd", "csrc", "pair_v7.inc")

# ---- row format (fgvc_split_f16f6p), byte offsets inside a 1 KiB pixel row; a lane (n, hi) adds 16 hi to every one of them
OFF_H = 0            # 256 f16: fragment j (K-16 step) at 32 j
OFF_H6M = 512        # h6 mains: group v at + 32 v  (16 B per lane half)
OFF_H6T = 640        # h6 tails: groups 2 w, 2 w + 1 at + 32 w (+ 8 for the odd one; 8 B per lane half and group)
OFF_L6M = 704
OFF_L6T = 832
OFF_SC = 896         # 8 B per lane half: scale bytes H(v = 0..3), L(v = 0..3)
RING = int(os.environ.get("V7_GEN_RING", "4"))             # key fragments in flight (divides 16: the ring continues from one tile into the next; 8: micro-benchmarks only)


MICRO = set(os.environ.get("V7_GEN_MICRO", "").split(","))   # timing-only toggles for tools/micro/pair_v7_stream.hip: nowait, nocvt, nop6
HEAD = (["A0", "A1", "HM0", "HT0", "A2", "A3", "LM0", "LT0", "QM0", "QT0", "SC"] if RING == 4 else
        ["A0", "A1", "A2", "A3", "A4", "A5", "HM0", "HT0", "A6", "A7", "LM0", "LT0", "QM0", "QT0", "SC"])     # the first reads of a tile, in the order they are ALWAYS issued


def chain(ind="        "):
    """The consumer's stream, continuous over tiles.  PART 1 = the first reads of a tile (HEAD) from `ka_l`, for the first tile of a run
    only; PART 3 = a tile's body: it finds HEAD issued (by PART 1 or by the body before it), and issues the NEXT tile's HEAD from `ka_n`
    in its last quarter -- the fragment ring simply runs on (A16 + j lands where A_j did), so that no chain starts with an LDS round trip
    and none ends with a drain.  Expects in scope: f32x16 acc; f16x8 ah[4]; i32x4v xm, ym, qm; i32x2v xt, yt, qt, ksc; f16x32 qh4[4];
    i32x6 q6l[4]; int sqH, sqL; uint32_t ka_l, ka_n, a_q6h, a_q6t, a_hand_free, a_next_filled; int peek_free, peek_fill; V7_RELEASE() = release this
    tile's ring slot (every read of it has been issued); V7_LOOKAHEAD() = find the next tile, wait for its key block, set ka_n."""
    out = []
    issued = []
    done_upto = [0]

    def rd(kind, dst, off, name, addr="ka_l"):
        out.append(f'{ind}asm volatile("ds_read_{kind} %0, %1 offset:{off}" : "=v"({dst}) : "v"({addr}) : "memory");')
        issued.append(name)

    def need(*names):
        last = max(issued.index(n) for n in names)
        if last < done_upto[0]:
            return
        n_after = len(issued) - 1 - last
        assert n_after <= 15, n_after
        if "nowait" not in MICRO:
            out.append(f'{ind}asm volatile("s_waitcnt lgkmcnt({n_after})" ::: "memory");')
        done_upto[0] = last + 1

    def A(j, nxt=False):
        rd("b128", f"ah[{j % RING}]", OFF_H + 32 * (j % 16), ("n" if nxt else "") + f"A{j % 16}", "ka_n" if nxt else "ka_l")

    def HM(v, nxt=False):
        a, p = ("ka_n", "n") if nxt else ("ka_l", "")
        rd("b128", "xm", OFF_H6M + 32 * v, f"{p}HM{v}", a)
        rd("b64", "xt", OFF_H6T + 32 * (v >> 1) + 8 * (v & 1), f"{p}HT{v}", a)

    def LM(v, nxt=False):
        a, p = ("ka_n", "n") if nxt else ("ka_l", "")
        rd("b128", "ym", OFF_L6M + 32 * v, f"{p}LM{v}", a)
        rd("b64", "yt", OFF_L6T + 32 * (v >> 1) + 8 * (v & 1), f"{p}LT{v}", a)

    def QM(v, nxt=False):
        # the query's own h6 piece of group v: the same bytes for every tile, parked in the LDS by the prologue (6 KiB per consumer).  Made
        # on the fly from the resident f16 fragments it cost 53 cycles per v_cvt_scalef32_pk32_fp6_f16 with the matrix pipe idle behind
        # it -- 212 of a tile's 1 140 (tools/micro/pair_v7_stream_variants.sh); resident it would need 24 registers the wave has not got
        p = "n" if nxt else ""
        rd("b128", "qm", 1024 * v, f"{p}QM{v}", "a_q6h")
        rd("b64", "qt", 512 * v, f"{p}QT{v}", "a_q6t")

    def SC(nxt=False):
        rd("b64", "ksc", OFF_SC, "nSC" if nxt else "SC", "ka_n" if nxt else "ka_l")

    def mfma_f(j):
        v, m = j // 4, j % 4
        b = f"V7_SUB8(qh4[{v}], {m})"
        if j == 0:
            # SrcC is the constant 0; `acc` is declared read-write all the same so that the compiler keeps the accumulator in the registers
            # of the tile before it (an early-clobber output got 16 registers of its own: two accumulators alive, operands spilled)
            out.append(f'{ind}asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "+v"(acc) : "v"(ah[{j % RING}]), "v"({b}));')
        else:
            out.append(f'{ind}asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah[{j % RING}]), "v"({b}));')

    def mfma_s(a6, b6, sa, sb, v):
        if "nop6" in MICRO:
            return
        sel = f"op_sel:[{v & 1},{v & 1},0] op_sel_hi:[{v >> 1},{v >> 1},0]"
        out.append(f'{ind}asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 {sel} cbsz:2 blgp:2" '
                   f': "+v"(acc) : "v"({a6}), "v"({b6}), "v"({sa}), "v"({sb}));')

    # ---- PART 1: HEAD from ka_l (first tile of a run)
    if RING == 4:
        A(0); A(1); HM(0); A(2); A(3); LM(0); QM(0); SC()
    else:
        A(0); A(1); A(2); A(3); A(4); A(5); HM(0); A(6); A(7); LM(0); QM(0); SC()
    assert issued == HEAD
    out.append("#elif FGVC_V7_PART == 3")
    # ---- PART 3: the body (HEAD in flight)
    for v in range(4):
        for half in range(2):
            for m in (2 * half, 2 * half + 1):
                j = 4 * v + m
                need(f"A{j}")
                mfma_f(j)
                A(j + RING, nxt=j + RING >= 16)          # the ring runs on into the next tile's first fragments
            if half == 0:
                need("SC", f"HM{v}", f"HT{v}")
                mfma_s("V7_CAT6(xm, xt)", f"q6l[{v}]", "ksc[0]", "sqL", v)       # h6_k x l6_q
                HM((v + 1) % 4, nxt=v == 3)
            else:
                need(f"LM{v}", f"LT{v}", f"QM{v}", f"QT{v}")
                mfma_s("V7_CAT6(ym, yt)", "V7_CAT6(qm, qt)", "ksc[1]", "sqH", v)  # l6_k x h6_q
                LM((v + 1) % 4, nxt=v == 3) if v < 3 else None
                if v == 1:
                    # two counters asked for now: has the selector read the tile before this one (needed at the hand-over behind this
                    # chain), has the NEXT tile's key block landed (needed six MFMAs on: normally it has, and nothing is waited for)
                    out.append(f'{ind}asm volatile("ds_read_b32 %0, %1" : "=v"(peek_free) : "v"(a_hand_free) : "memory");')
                    issued.append("PK")
                    out.append(f'{ind}asm volatile("ds_read_b32 %0, %1" : "=v"(peek_fill) : "v"(a_next_filled) : "memory");')
                    issued.append("PF")
                if v == 2:
                    # A15, HM3 and now LM3: every read of this key block has been issued -> release its slot; then the next tile
                    # (the reads that follow go to ITS key block)
                    out.append(f"{ind}V7_RELEASE();")
                    need("PF")
                    out.append(f"{ind}V7_LOOKAHEAD();")
                if v < 3:
                    QM(v + 1)
                if v == 3:
                    # (the rest of the next tile's first reads follows the hand-over: a wave may have 15 LDS operations in flight, and the
                    # hand-over's four stores and its count stalled behind eleven reads until some of them had returned)
                    out.append("#elif FGVC_V7_PART == 4")
                    LM(0, nxt=True)
                    QM(0, nxt=True)
                    SC(nxt=True)
                    out.append("#elif FGVC_V7_PART == 5")
    # (PART 5 = the wait for the counter the hand-over reads: placed by the kernel in front of PART 4)
    last = issued.index("PK")
    n_after = len([x for x in issued[last + 1:] if x not in ("nLM0", "nLT0", "nQM0", "nQT0", "nSC")])
    if last >= done_upto[0]:
        out.append(f'{ind}asm volatile("s_waitcnt lgkmcnt({n_after})" ::: "memory");')
        done_upto[0] = last + 1
    rest = issued[done_upto[0]:]
    nxt = [x for x in issued if x.startswith("n")]
    assert nxt == ["n" + h for h in HEAD] and rest == nxt[len(nxt) - len(rest):], (nxt, rest)   # the next tile's HEAD, in HEAD's order; what is in flight is its tail
    n_reads = len([x for x in issued if not x.startswith("n")])
    out.insert(0, f"{ind}// one tile: {n_reads} LDS reads (+ the next tile's first 9), 16 f16 + 8 scaled FP6 MFMAs (generated by tools/gen_pair_v7.py -- do not edit)")
    return "\n".join(out) + "\n"


# ------------------------------------------------------------------------------------------------------------------ selector stream
def micro_ops7(K):
    sel = g5.net(f"FGVC_SELNET_16_TOP{K}")
    vm = g5.net(f"FGVC_VMERGE_ASC_{K}")
    ops = []
    kind = {"v_dx0": "v", "v_dy0": "v", "s_r2lim": "S"}
    for r in range(16):
        kind[f"ck[{r}]"] = "v"
    for i in range(K):
        kind[f"lk[{i}]"] = "v"

    def op(mn, dst, args, k="v"):
        kind.setdefault(dst, k)
        ops.append((mn, dst, args))

    for a in range(4):
        op("v_add_u32", f"xs{a}", [a, "v_dx0"])
        op("v_mul_i32_i24", f"xsq{a}", [f"xs{a}", f"xs{a}"])
    for r in range(16):
        a = r // 4
        if r % 4 == 0:
            op("v_add_u32", f"ys{a}", [a, "v_dy0"])
            op("v_mul_i32_i24", f"ysq{a}", [f"ys{a}", f"ys{a}"])
            op("v_sub_u32", f"ylim{a}", ["s_r2lim", f"ysq{a}"])
        op("v_cmp_le_i32", f"kc{r}", [f"xsq{r & 3}", f"ylim{a}"], "s")
        op("v_cndmask_b32_e64", f"ck[{r}]", [0, f"ck[{r}]", f"kc{r}"])
    for i, j in sel:                       # descending: ck[i] >= ck[j]
        op("v_max_u32", f"ck[{i}]", [f"ck[{i}]", f"ck[{j}]"])
        op("v_min_u32", f"ck[{j}]", [f"ck[{i}]", f"ck[{j}]"])
    for i in range(K):                     # top K of (sorted candidates) U (ascending list): V-shaped
        op("v_max_u32", f"lk[{i}]", [f"ck[{i}]", f"lk[{i}]"])
    for i, j in vm:                        # bitonic merger, ascending: lk[i] <= lk[j]
        op("v_min_u32", f"lk[{i}]", [f"lk[{i}]", f"lk[{j}]"])
        op("v_max_u32", f"lk[{j}]", [f"lk[{i}]", f"lk[{j}]"])
    return ops, kind


def prepare7(K):
    ops, kind = micro_ops7(K)
    sched = g5.schedule(g5.versioned(ops))
    g5.FINAL.clear()
    for mn, dst, srcs in sched:
        g5.FINAL[dst[0]] = max(g5.FINAL.get(dst[0], 0), dst[1])
    return sched, kind


def declarations7(sched, kind, ind):
    ints, masks = [], []
    for mn, dst, srcs in sched:
        c = g5.cvar(dst)
        if c == dst[0] and dst[0].startswith(("ck[", "lk[")):
            continue
        (masks if kind[dst[0]] == "s" else ints).append(c)
    s = ""
    for i in range(0, len(ints), 12):
        s += ind + "int " + ", ".join(f"{v} = 0" for v in ints[i:i + 12]) + ";\n"
    for i in range(0, len(masks), 8):
        s += ind + "unsigned long long " + ", ".join(f"{v} = 0" for v in masks[i:i + 8]) + ";\n"
    return s


def select(K, ind="    "):
    sched, kind = prepare7(K)
    chunks = g5.cut(sched, (len(sched) + 7) // 8)
    lo = g5.live_outs(chunks)
    s = f"{ind}// K = {K}: {len(sched)} vector operations (generated by tools/gen_pair_v7.py -- do not edit)\n"
    s += declarations7(sched, kind, ind)
    for c, l in zip(chunks, lo):
        s += g5.render_slice(c, kind, l, ind)
    return s


def self_check7(K):
    sched, kind = prepare7(K)
    rng = random.Random(70 + K)
    M = 0xFFFFFFFF
    for _ in range(300):
        env = {"v_dx0": rng.randint(-20, 20), "v_dy0": rng.randint(-20, 20), "s_r2lim": rng.choice([-1, 225, 0x3fffffff])}
        seq = rng.randint(0, 63)
        for r in range(16):
            env[f"ck[{r}]"] = (rng.randint(1 << 20, 3 << 20) << 10) | (seq << 4) | (15 - r)
        lk = sorted(((rng.randint(1 << 20, 3 << 20) << 10) | rng.randint(0, 1023)) if rng.random() < 0.8 else 0 for _ in range(K))
        for i in range(K):
            env[f"lk[{i}]"] = lk[i]
        vals = {(k, 0): v for k, v in env.items()}
        for mn, dst, srcs in sched:
            a = [vals[v] if isinstance(v, tuple) else v for v in srcs]
            if mn == "v_add_u32":
                res = a[0] + a[1]
            elif mn == "v_mul_i32_i24":
                res = a[0] * a[1]
            elif mn == "v_sub_u32":
                res = a[0] - a[1]
            elif mn == "v_cmp_le_i32":
                res = a[0] <= a[1]
            elif mn == "v_cndmask_b32_e64":
                res = a[1] if a[2] else a[0]
            elif mn == "v_max_u32":
                res = max(a[0] & M, a[1] & M)
            elif mn == "v_min_u32":
                res = min(a[0] & M, a[1] & M)
            else:
                raise AssertionError(mn)
            vals[dst] = res
        keys = []
        for r in range(16):
            ok = (env["v_dx0"] + (r & 3)) ** 2 <= env["s_r2lim"] - (env["v_dy0"] + (r >> 2)) ** 2
            keys.append(env[f"ck[{r}]"] if ok else 0)
        want = sorted(keys + lk)[-K:]
        got = [vals[(f"lk[{i}]", g5.FINAL[f"lk[{i}]"])] for i in range(K)]
        assert got == want, (got, want)
    print(f"K = {K}: selector stream verified on 300 random tiles; {len(sched)} operations")


def main():
    txt = "// GENERATED by tools/gen_pair_v7.py -- do not edit.  Included by pair_topk_v7.hpp (FGVC_V7_PART = 1: the first reads of a tile, 3: its chain, 2: the selector's stream).\n"
    txt += "#if FGVC_V7_PART == 1\n" + chain() + "#endif\n"
    for K in (5, 10):
        self_check7(K)
        txt += f"#if FGVC_V7_K == {K} && FGVC_V7_PART == 2\n" + select(K) + "#endif\n"
    open(OUT, "w").write(txt)
    print("wrote", OUT, len(txt.splitlines()), "lines")


if __name__ == "__main__":
    main()
