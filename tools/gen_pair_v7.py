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
RING = int(os.environ.get("V7_GEN_RING", "4"))             # key fragments in flight
DBG = set(os.environ.get("V7_GEN_DBG", "").split(","))     # bisection toggles: f0early, qscres, nopeek, headall


def chain(ind="        "):
    """One tile.  Expects in scope: f32x16 acc; f16x8 ah[4]; i32x4 xm, ym; i32x2 xt, yt, ksc; u32x6 q6h; f16x32 qh4[4]; i32x6 q6l[4];
    int sqH, sqL; float qsc[4]; uint32_t ka_l (LDS address of the lane's key row); V7_RELEASE() = the statement that releases the
    ring slot (placed after the last read of the block has been issued)."""
    out = []
    issued = []            # names of reads in issue order
    done_upto = [0]        # reads [0, done_upto) are known complete after the last wait

    def rd(kind, dst, off, name):
        out.append(f'{ind}asm volatile("ds_read_{kind} %0, %1 offset:{off}" : "=v"({dst}) : "v"(ka_l) : "memory");')
        issued.append(name)

    def need(*names):
        last = max(issued.index(n) for n in names)
        if last < done_upto[0]:
            return
        n_after = len(issued) - 1 - last
        out.append(f'{ind}asm volatile("s_waitcnt lgkmcnt({n_after})" ::: "memory");')
        done_upto[0] = last + 1

    def A(j):
        rd("b128", f"ah[{j % RING}]", OFF_H + 32 * j, f"A{j}")

    def HM(v):
        if "nofp6" in DBG:
            return
        rd("b128", "xm", OFF_H6M + 32 * v, f"HM{v}")
        rd("b64", "xt", OFF_H6T + 32 * (v >> 1) + 8 * (v & 1), f"HT{v}")

    def LM(v):
        if "nofp6" in DBG:
            return
        rd("b128", "ym", OFF_L6M + 32 * v, f"LM{v}")
        rd("b64", "yt", OFF_L6T + 32 * (v >> 1) + 8 * (v & 1), f"LT{v}")

    def mfma_f(j):
        v, m = j // 4, j % 4
        b = f"V7_SUB8(qh4[{v}], {m})"
        if j == 0:
            # SrcC is the constant 0; `acc` is declared read-write all the same so that the compiler keeps the accumulator in the registers
            # of the tile before it (an early-clobber output got 16 registers of its own: two accumulators alive, operands spilled)
            if "f0early" in DBG:
                out.append(f'{ind}asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(ah[{j % RING}]), "v"({b}));')
            else:
                out.append(f'{ind}asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "+v"(acc) : "v"(ah[{j % RING}]), "v"({b}));')
        else:
            out.append(f'{ind}asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(ah[{j % RING}]), "v"({b}));')

    def mfma_s(a6, b6, sa, sb, v, bcls="v"):
        if "nofp6" in DBG:
            return
        sel = f"op_sel:[{v & 1},{v & 1},0] op_sel_hi:[{v >> 1},{v >> 1},0]"
        out.append(f'{ind}asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 {sel} cbsz:2 blgp:2" '
                   f': "+v"(acc) : "v"({a6}), "{bcls}"({b6}), "v"({sa}), "v"({sb}));')

    # head of the tile: the first reads (the kernel hands the PREVIOUS tile over between these and the first MFMA: the latency of the
    # reads and the conversion / LDS stores of the hand-over cover each other; LDS operations nobody counts here only make the waits
    # stricter than they need be)
    if "nofp6" not in DBG:
        rd("b64", "ksc", OFF_SC, "SC")
    if "headall" in DBG:
        A(0); A(1); HM(0); A(2); A(3); LM(0)
        out.append("#elif FGVC_V7_PART == 3")
    else:
        A(0); A(1)
        out.append("#elif FGVC_V7_PART == 3")
        HM(0)                        # (after the hand-over: with more reads in front of it the kernel spilled query operands)
        for j in range(2, RING):
            A(j)
        LM(0)
    for v in range(4):
        # the scale of this group's conversion, 2^sh from the query's scale byte: (byte + 4) << 23 as a float (two operations instead of
        # four resident registers), made two MFMAs ahead of the conversion that reads it
        if "qscres" in DBG:
            out.append(f'{ind}qsc_t = __builtin_bit_cast(unsigned int, qsc[{v}]);')
        else:
            out.append(f'{ind}asm volatile("v_bfe_u32 %0, %1, {8 * v}, 8\\n\\tv_lshl_add_u32 %0, %0, 23, %2" : "=&v"(qsc_t) : "v"(sqH), "v"(c_exp4));')
            if "chkqsc" in DBG:
                out.append(f'{ind}if (qsc_t != __builtin_bit_cast(unsigned int, qsc[{v}])) {{ g_pair_v5_timeout = 1; g_pair_v5_probe[{v}] = ((long long)qsc_t << 32) | __builtin_bit_cast(unsigned int, qsc[{v}]); g_pair_v5_probe[4 + {v}] = sqH; }}')
        for half in range(2):
            for m in (2 * half, 2 * half + 1):
                j = 4 * v + m
                need(f"A{j}")
                mfma_f(j)
                if j + RING < 16:
                    A(j + RING)
            if half == 0:
                # the query's h part of this group in FP6, made two f16 MFMAs before its reader (a VALU result needs a few issue
                # slots before a matrix instruction may read it; nothing in an asm statement is padded by the compiler)
                # (early-clobber: given the chance the allocator puts the six result registers on top of the scale operand, and the
                # instruction -- several passes over its 32 elements -- then reads a scale it has already overwritten: measured, 1e-4 cosine)
                out.append(f'{ind}asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(q6h) : "v"(qh4[{v}]), "v"(qsc_t));')
                if "nofp6" not in DBG:
                    need("SC", f"HM{v}", f"HT{v}")
                mfma_s("V7_CAT6(xm, xt)", f"q6l[{v}]", "ksc[0]", "sqL", v)       # h6_k x l6_q
                if v + 1 < 4:
                    HM(v + 1)
            else:
                if "nofp6" not in DBG:
                    need(f"LM{v}", f"LT{v}")
                mfma_s("V7_CAT6(ym, yt)", "q6h", "ksc[1]", "sqH", v)              # l6_k x h6_q
                if v + 1 < 4:
                    LM(v + 1)
                if v + 1 == 3:                 # A15, HM3 and now LM3: every read of the block has been issued
                    out.append(f"{ind}V7_RELEASE();")
                    # two counters asked for now, needed after the chain: has the selector read the tile before this one, has the next
                    # key block landed
                    if "nopeek" in DBG:
                        out.append(f"{ind}peek_free = 0; peek_fill = -1;")
                    else:
                        out.append(f'{ind}asm volatile("ds_read_b32 %0, %1" : "=v"(peek_free) : "v"(a_hand_free) : "memory");')
                        issued.append("PK1")
                        out.append(f'{ind}asm volatile("ds_read_b32 %0, %1" : "=v"(peek_fill) : "v"(a_next_filled) : "memory");')
                        issued.append("PK2")
    if "nopeek" not in DBG:
        need("PK1", "PK2")
    assert done_upto[0] == len(issued), "a read nobody waited for"
    n_reads = len(issued)
    out.insert(0, f"{ind}// one tile: {n_reads} LDS reads, 16 f16 + 8 scaled FP6 MFMAs (generated by tools/gen_pair_v7.py -- do not edit)")
    # sanity: a ring register is re-read only after its reader has been issued (program order of the statements above)
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
