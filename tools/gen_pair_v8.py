"""Generate fgvc_amd/csrc/pair_v8.inc for pair_topk_kernel_v8 (pair_topk_v8.hpp): a 32 x 32 tile of fgvc_pair_topk_f16f6 as ONE
assembly statement per tile --

  the matrix chain: the SAME 16 v_mfma_f32_32x32x16_f16 + 8 v_mfma_scale_f32_32x32x64_f8f6f4 in the SAME order on the same operands as
  tools/gen_pair_v7.py writes them (scores bit-identical to pair_topk_kernel_v7's), for a wave that owns its query block outright: all
  of the query's operands are resident (h, l6 AND h6), every LDS read of the tile goes to the key block, nothing is handed over;

  and, dealt behind its matrix instructions, the SELECTION of the tile before it: mask predicate, 60-comparator selection network on
  32-bit keys, merge into the running list (the operations of tools/gen_pair_v7.py's selector stream).  A wave alone issues one
  instruction per ~4 cycles; a matrix instruction occupies the pipe for 32 and the issue port for 8 -- what is issued behind it is free.

Why one statement: with a statement per instruction (the first build) the register allocator rotated the query's 6-register FP6 operands
through v_mov chains, copied the accumulator and spilled (270 registers wanted, 256 there); operands pinned to physical registers were
COPIED into them at every statement.  Inside one statement this script allocates: the chain's buffers and the selection's temporaries
are physical registers named in the text and listed as clobbers (they live and die inside the statement), everything that crosses the
statement is an ordinary operand.  The selection's values are renamed, not moved: a compare-exchange writes its maximum to a free
register and its minimum over a dying source; where the final list lies is told to the C++ side as a permutation.

  FGVC_V8_PART 1      chain only (a pair's first tile)
  FGVC_V8_PART 2, K   chain + selection of the pending tile
  FGVC_V8_PART 3, K   selection only (a pair's last tile)

The LDS returns a wave's reads in order, so a wait before an instruction = the number of reads issued after the youngest one it needs;
this script counts them.

    python tools/gen_pair_v8.py        # rewrites the .inc; the build does not run it
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_pair_v5_chain as g5          # noqa: E402  (net: the comparator lists of csrc/sortnet.hpp)

OUT = os.path.join(os.path.dirname(HERE), "fgvc_amd", "csrc", "pair_v8.inc")
# row format (fgvc_split_f16f6p), byte offsets inside a pixel row; a lane (n, hi) adds 16 hi to every one of them
OFF_H, OFF_H6M, OFF_H6T, OFF_L6M, OFF_L6T, OFF_SC = 0, 512, 640, 704, 832, 896
RING = int(os.environ.get('V8_RING', '2'))                                 # key fragments in flight
N_MFMA = 24

# ---- physical registers that live inside a statement only (clobbers): the chain's buffers, then the selection's temporaries
ONE6 = os.environ.get('V8_ONE6', '1') == '1'      # one buffer for the key block's two FP6 operands
T_KSC = 254          # scale bytes: v[254:255]
T_Y6 = 248           # key l6 operand: v[248:253]
T_X6 = T_Y6 if ONE6 else 242           # key h6 operand
T_AH = T_X6 - 4 * RING           # ah[i]: 4 registers each
T_SEL_TOP = 225      # selection temporaries: v225 downwards


def vr(base, n):
    return f"v[{base}:{base + n - 1}]" if n > 1 else f"v{base}"


# ------------------------------------------------------------------------------------------------------------------ the selection
def select_ops(K):
    """program order; values are names, every write makes a new version (name, k).  ('cmpmask', dst, [a, b, src]) = dst <- a <= b ? src : 0"""
    sel = g5.net(f"FGVC_SELNET_16_TOP{K}")
    vm = g5.net(f"FGVC_VMERGE_ASC_{K}")
    ops = []
    for a in range(4):
        ops.append(("v_add_u32", f"xs{a}", [a, "v_dx0"]))
        ops.append(("v_mul_i32_i24", f"xsq{a}", [f"xs{a}", f"xs{a}"]))
    for rr in range(4):
        ops.append(("v_add_u32", f"ys{rr}", [rr, "v_dy0"]))
        ops.append(("v_mul_i32_i24", f"ysq{rr}", [f"ys{rr}", f"ys{rr}"]))
        ops.append(("v_sub_u32", f"ylim{rr}", ["s_r2lim", f"ysq{rr}"]))
        for c in range(4):
            r = 4 * rr + c
            ops.append(("cmpmask", f"ck{r}", [f"xsq{c}", f"ylim{rr}", f"ck{r}"]))
    for i, j in sel:                       # descending: ck[i] >= ck[j]
        ops.append(("v_max_u32", f"ck{i}", [f"ck{i}", f"ck{j}"]))
        ops.append(("v_min_u32", f"ck{j}", [f"ck{i}", f"ck{j}"]))
    for i in range(K):                     # top K of (sorted candidates) U (ascending list): V-shaped
        ops.append(("v_max_u32", f"lk{i}", [f"ck{i}", f"lk{i}"]))
    for i, j in vm:                        # bitonic merger, ascending: lk[i] <= lk[j]
        ops.append(("v_min_u32", f"lk{i}", [f"lk{i}", f"lk{j}"]))
        ops.append(("v_max_u32", f"lk{j}", [f"lk{i}", f"lk{j}"]))
    return ops


def versioned(ops):
    """SSA: sources refer to the version current when the op is read IN PROGRAM ORDER -- a compare-exchange's second op reads the
    versions its first op read (the pair is written (max, min) on the OLD values)"""
    cur = {}
    out = []
    k = 0
    while k < len(ops):
        mn, dst, srcs = ops[k]
        pair = (k + 1 < len(ops) and mn in ("v_max_u32", "v_min_u32") and ops[k + 1][0] in ("v_max_u32", "v_min_u32")
                and ops[k + 1][2] == srcs and ops[k + 1][1] != dst)
        group = [ops[k], ops[k + 1]] if pair else [ops[k]]
        vs = [(s, cur.get(s, 0)) if isinstance(s, str) else s for s in srcs]
        for mn2, dst2, _ in group:
            cur[dst2] = cur.get(dst2, 0) + 1
            out.append((mn2, (dst2, cur[dst2]), vs))
        k += len(group)
    return out, cur


def allocate(K):
    """-> (lines per op, perm, n_temps): every value gets a location: operand w[k] ('%k' filled in later as {wK}) or a physical temporary"""
    sops, final = versioned(select_ops(K))
    last_use = {}
    for idx, (mn, dst, srcs) in enumerate(sops):
        for s in srcs:
            if isinstance(s, tuple):
                last_use[s] = idx
    finals = {(f"lk{i}", final[f"lk{i}"]) for i in range(K)}
    loc = {}
    for r in range(16):
        loc[(f"ck{r}", 0)] = f"{{w{r}}}"
    for i in range(K):
        loc[(f"lk{i}", 0)] = f"{{w{16 + i}}}"
    for nm in ("v_dx0", "v_dy0", "s_r2lim"):
        loc[(nm, 0)] = "{" + nm + "}"
    free_ops, free_phys = [], []
    next_phys = [T_SEL_TOP]
    max_phys = [0]

    def is_operand(l):
        return l.startswith("{w")

    def take(final_value, dying):
        # a dying source's place first (the instruction reads before it writes); a final value must end in an operand
        for l in dying:
            if not final_value or is_operand(l):
                dying.remove(l)
                return l
        if final_value:
            assert free_ops, "no free operand register for a final value"
            return free_ops.pop()
        if free_phys:
            return free_phys.pop()
        if free_ops:
            return free_ops.pop()
        l = "{t%d}" % max_phys[0]
        max_phys[0] += 1
        return l

    lines = []
    for idx, (mn, dst, srcs) in enumerate(sops):
        src_locs = [loc[s] if isinstance(s, tuple) else str(s) for s in srcs]
        dying = []
        for s in sorted(set(x for x in srcs if isinstance(x, tuple))):
            if last_use[s] == idx and s[0] not in ("v_dx0", "v_dy0", "s_r2lim"):
                dying.append(loc[s])
        dead_result = dst not in last_use and dst not in finals
        d = take(dst in finals, dying)
        loc[dst] = d
        if mn == "cmpmask":
            lines.append([f"v_cmp_le_i32 vcc, {src_locs[0]}, {src_locs[1]}", f"v_cndmask_b32 {d}, 0, {src_locs[2]}, vcc"])
        else:
            lines.append([f"{mn} {d}, {src_locs[0]}, {src_locs[1]}"])
        for l in dying:                     # places of sources that died here and were not taken
            (free_ops if is_operand(l) else free_phys).append(l)
        if dead_result:
            (free_ops if is_operand(d) else free_phys).append(d)
    perm = [int(loc[(f"lk{i}", final[f"lk{i}"])][2:-1]) for i in range(K)]
    return lines, perm, max_phys[0], sops


def self_check(K):
    """run the allocated text on random integers against a direct evaluation of the selection"""
    lines, perm, n_phys, sops = allocate(K)
    rng = random.Random(80 + K)
    M = 0xFFFFFFFF
    for _ in range(300):
        env = {"{v_dx0}": rng.randint(-20, 20), "{v_dy0}": rng.randint(-20, 20), "{s_r2lim}": rng.choice([-1, 224, 0x3fffffff])}
        ck = [(rng.randint(1 << 20, 3 << 20) << 10) | rng.randint(0, 1023) for _ in range(16)]
        if rng.random() < 0.2:
            ck[rng.randrange(16)] = ck[rng.randrange(16)]
        lk = sorted(((rng.randint(1 << 20, 3 << 20) << 10) | rng.randint(0, 1023)) if rng.random() < 0.8 else 0 for _ in range(K))
        for r in range(16):
            env[f"{{w{r}}}"] = ck[r]
        for i in range(K):
            env[f"{{w{16 + i}}}"] = lk[i]
        vcc = False

        def val(t):
            return env[t] if t in env else int(t)
        for group in lines:
            for ln in group:
                mn, rest = ln.split(" ", 1)
                a = [x.strip() for x in rest.split(",")]
                if mn == "v_cmp_le_i32":
                    vcc = val(a[1]) <= val(a[2])
                elif mn == "v_cndmask_b32":
                    env[a[0]] = val(a[2]) if vcc else val(a[1])
                elif mn == "v_add_u32":
                    env[a[0]] = val(a[1]) + val(a[2])
                elif mn == "v_mul_i32_i24":
                    env[a[0]] = val(a[1]) * val(a[2])
                elif mn == "v_sub_u32":
                    env[a[0]] = val(a[1]) - val(a[2])
                elif mn == "v_max_u32":
                    env[a[0]] = max(val(a[1]) & M, val(a[2]) & M)
                elif mn == "v_min_u32":
                    env[a[0]] = min(val(a[1]) & M, val(a[2]) & M)
                else:
                    raise AssertionError(mn)
        keys = []
        for r in range(16):
            ok = (env["{v_dx0}"] + (r & 3)) ** 2 <= env["{s_r2lim}"] - (env["{v_dy0}"] + (r >> 2)) ** 2
            keys.append(ck[r] if ok else 0)
        want = sorted(keys + lk)[-K:]
        got = [env[f"{{w{perm[i]}}}"] for i in range(K)]
        assert got == want, (got, want)
    n_ops = sum(len(g) for g in lines)
    print(f"K = {K}: selection verified on 300 random tiles; {n_ops} vector operations, {n_phys} temporaries, final list in w{perm}")
    return n_ops


# ------------------------------------------------------------------------------------------------------------------ the statement
def statement(K, with_chain, with_select, ind="        "):
    """ONE asm statement.  Operands: %0 acc (out); w[0..25] (in/out: ck[0..15], lk[0..9] -- K = 5: lk[0..4]) when selecting; inputs ka,
    qhs[16], q6l[4], q6h[4], sqH, sqL (chain) and v_dx0, v_dy0, s_r2lim (selection)."""
    outs, ins = [], []
    name_of = {}

    def add_out(c, expr, key):
        name_of[key] = f"%{len(outs)}"
        outs.append(f'"{c}"({expr})')

    if with_chain:
        add_out("=&v", "acc", "acc")
    nw = 16 + K
    n_tmp = allocate(K)[2] if with_select else 0
    if with_select:
        for k in range(nw):
            add_out("+v", f"w[{k}]", f"w{k}")
        for k in range(n_tmp):
            add_out("=&v", f"t[{k}]", f"t{k}")
    n_out = len(outs)

    def add_in(c, expr, key):
        name_of[key] = f"%{n_out + len(ins)}"
        ins.append(f'"{c}"({expr})')

    if with_chain:
        add_in("v", "ka", "ka")
        for j in range(16):
            add_in("v", f"qhs[{j}]", f"qh{j}")
        for v in range(4):
            add_in("v", f"q6l[{v}]", f"q6l{v}")
        for v in range(4):
            add_in("v", f"q6h[{v}]", f"q6h{v}")
        add_in("v", "sqH", "sqH")
        add_in("v", "sqL", "sqL")
    if with_select:
        add_in("v", "v_dx0", "v_dx0")
        add_in("v", "v_dy0", "v_dy0")
        add_in("s", "s_r2lim", "s_r2lim")

    text = []
    sel_groups, perm, n_phys = [], None, 0
    if with_select:
        lines, perm, n_phys, _ = allocate(K)
        sel_groups = [[ln.format(**name_of) for ln in g] for g in lines]
    n_slots = N_MFMA if with_chain else 1
    # deal the selection's operations behind the matrix instructions (a cmp + cndmask pair stays together)
    per_slot = [[] for _ in range(n_slots)]
    total = sum(len(g) for g in sel_groups)
    done = 0
    for g in sel_groups:
        per_slot[min(n_slots - 1, done * n_slots // max(total, 1))].extend(g)
        done += len(g)

    if with_chain:
        issued = []
        done_upto = [0]
        n_mfma = [0]
        ACC, KA = name_of["acc"], name_of["ka"]

        def rd(kind, reg, off, name):
            text.append(f"ds_read_{kind} {reg}, {KA} offset:{off}")
            issued.append(name)

        def need(*names):
            last = max(issued.index(n) for n in names)
            if last < done_upto[0]:
                return
            n_after = len(issued) - 1 - last
            assert n_after <= 15, n_after
            text.append(f"s_waitcnt lgkmcnt({n_after})")
            done_upto[0] = last + 1

        def A(j):
            if j < 16:
                rd("b128", vr(T_AH + 4 * (j % RING), 4), OFF_H + 32 * j, f"A{j}")

        def HM(v):
            rd("b128", vr(T_X6, 4), OFF_H6M + 32 * v, f"HM{v}")
            rd("b64", vr(T_X6 + 4, 2), OFF_H6T + 32 * (v >> 1) + 8 * (v & 1), f"HT{v}")

        def LM(v):
            rd("b128", vr(T_Y6, 4), OFF_L6M + 32 * v, f"LM{v}")
            rd("b64", vr(T_Y6 + 4, 2), OFF_L6T + 32 * (v >> 1) + 8 * (v & 1), f"LT{v}")

        def after_mfma():
            text.extend(per_slot[n_mfma[0]])
            n_mfma[0] += 1

        def mfma_f(j):
            c = "0" if j == 0 else ACC
            text.append(f"v_mfma_f32_32x32x16_f16 {ACC}, {vr(T_AH + 4 * (j % RING), 4)}, {name_of[f'qh{j}']}, {c}")

        def mfma_s(a_base, b_name, sa, sb_name, v):
            sel = f"op_sel:[{v & 1},{v & 1},0] op_sel_hi:[{v >> 1},{v >> 1},0]"
            text.append(f"v_mfma_scale_f32_32x32x64_f8f6f4 {ACC}, {vr(a_base, 6)}, {name_of[b_name]}, {ACC}, v{sa}, {name_of[sb_name]} {sel} cbsz:2 blgp:2")

        if ONE6:
            A(0); A(1); HM(0)
            rd("b64", vr(T_KSC, 2), OFF_SC, "SC")
        else:
            A(0); A(1); HM(0); A(2); A(3); LM(0)
            rd("b64", vr(T_KSC, 2), OFF_SC, "SC")
        for v in range(4):
            for half in range(2):
                for m in (2 * half, 2 * half + 1):
                    j = 4 * v + m
                    need(f"A{j}")
                    mfma_f(j)
                    A(j + RING)
                    after_mfma()
                if half == 0:
                    need("SC", f"HM{v}", f"HT{v}")
                    mfma_s(T_X6, f"q6l{v}", T_KSC, "sqL", v)           # h6_k x l6_q
                    if ONE6:
                        LM(v)
                    elif v < 3:
                        HM(v + 1)
                else:
                    need(f"LM{v}", f"LT{v}")
                    mfma_s(T_Y6, f"q6h{v}", T_KSC + 1, "sqH", v)       # l6_k x h6_q
                    if v < 3:
                        HM(v + 1) if ONE6 else LM(v + 1)
                after_mfma()
        assert n_mfma[0] == N_MFMA and done_upto[0] == len(issued), (n_mfma, done_upto, len(issued))
        text.append("s_nop 15")           # MFMA result -> vector read: the last MFMA's passes must have written back
        text.append("s_nop 7")
    else:
        text.extend(per_slot[0])

    clob = ['"memory"']
    if with_chain:
        clob += [f'"v{r}"' for r in range(T_AH, 256)]
    if with_select:
        clob += ['"vcc"']
    body = "\\n\\t".join(text)
    s = ""
    if with_select:
        s += f"{ind}unsigned int w[{nw}], t[{n_tmp}];\n"
        s += f"{ind}" + " ".join(f"w[{r}] = ck[{r}];" for r in range(16)) + "\n"
        s += f"{ind}" + " ".join(f"w[{16 + i}] = lk[{i}];" for i in range(K)) + "\n"
    s += f'{ind}asm volatile("{body}"\n{ind}             : {", ".join(outs)}\n{ind}             : {", ".join(ins)}\n{ind}             : {", ".join(clob)});\n'
    if with_select:
        s += f"{ind}" + " ".join(f"lk[{i}] = w[{perm[i]}];" for i in range(K)) + "\n"
    return s, len(text)


def staging(ind="      "):
    """the 32 rows of an interior key block by LDS-DMA: 4 instructions per row, no register but vcc -- m0 walks the slot's rows, vcc the
    bank's.  Operands: %0 lane offset (16 lane), %1 pointer to the block's first pixel row, %2 LDS address of the slot, %3 bytes from
    one pixel's row to the next, %4 bytes from a block row's last pixel to the next block row's first."""
    t = ["s_mov_b32 exec_hi, 0x3ffffff", "s_mov_b32 m0, %2", "s_mov_b64 vcc, %1", "global_load_lds_dwordx4 %0, vcc"]
    for pr in range(4):
        for c in range(8):
            if pr == 0 and c == 0:
                continue
            t += ["s_add_u32 m0, m0, 0x3b0", f"s_add_u32 vcc_lo, vcc_lo, {'%3' if c else '%4'}", "s_addc_u32 vcc_hi, vcc_hi, 0", "global_load_lds_dwordx4 %0, vcc"]
    t.append("s_mov_b32 exec_hi, -1")
    body = "\\n\\t".join(t)
    return f'{ind}asm volatile("{body}"\n{ind}             :: "v"(lane16), "s"(src), "s"(dst), "s"(s_rowb), "s"(s_rowskip) : "memory", "vcc", "scc");\n', len(t)


def main():
    txt = ("// GENERATED by tools/gen_pair_v8.py -- do not edit.  Included by pair_topk_v8.hpp (FGVC_V8_PART = 1: a tile's matrix chain, 2: the chain with\n"
           "// the selection of the tile before it behind its matrix instructions, 3: the selection alone), one assembly statement each.\n")
    s, n = statement(10, True, False)
    txt += f"#if FGVC_V8_PART == 1\n        // {n} instructions\n" + s + "#endif\n"
    for K in (5, 10):
        self_check(K)
        s, n = statement(K, True, True)
        txt += f"#if FGVC_V8_K == {K} && FGVC_V8_PART == 2\n        // {n} instructions\n" + s + "#endif\n"
        s, n = statement(K, False, True)
        txt += f"#if FGVC_V8_K == {K} && FGVC_V8_PART == 3\n        // {n} instructions\n" + s + "#endif\n"
    st, n = staging()
    txt += f"#if FGVC_V8_PART == 4\n      // {n} instructions\n" + st + "#endif\n"
    open(OUT, "w").write(txt)
    print("wrote", OUT, len(txt.splitlines()), "lines")


if __name__ == "__main__":
    main()
