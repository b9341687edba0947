"""Generate fgvc_amd/csrc/pair_v5_chain.inc: the 48-MFMA chain of fgvc_pair_topk_f16x3 with the selection of the pending tile cut
into one slice of vector instructions per MFMA, every slice ONE volatile inline-asm statement.

Why generated assembly slices: the slices are pure register arithmetic.  Written as C++, instruction selection places them anywhere
in the loop body (first build: all 300 operations in one gap between two MFMAs); pinned by scheduling barriers and empty asm
fences they keep their gap, but every fence that reads a register a vector operation has just written is padded with wait states
by the compiler (~200 s_nop per tile) and the chain ran at 62 cycles per MFMA (tools/micro/mfma_valu_fill.hip: 4 vector operations
per gap are free, 6 cost 37 cycles per MFMA, 10 cost 61).  Volatile asm statements keep their program order against each other,
need no fences, and get no padding.

Instruction order inside the stream: the selection is a list of micro-operations with register dependencies (sorting networks have
~6 independent comparators per layer); a dependent vector operation issued right behind its producer waits ~2.6 cycles (measured:
the network in Batcher's order ran at 5.5 cycles per operation, 4 is the issue rate), so the generator list-schedules the
micro-operations inside a sliding window -- an operation is emitted only when its operands are at least two instructions old, if
any such operation is ready -- and compare results travel in SGPR pairs (not VCC) so that compares of different comparators may
overlap.  The scheduled stream is then cut into 48 near-equal slices.  `self_check` executes the scheduled stream in Python on
random tiles against a direct evaluation of the selection before anything is written.

The include expects in scope: f32x16 acc; f16x8 ah[3], al[3] (steps 0 and 1 already requested), qh[16], ql[16];
const unsigned char* ka; int ck[16] (raw fixed-point scores of the pending tile), lk[K], lb[K]; int v_dy0, v_dx0 (key block origin
minus query position, per lane), v_base, v_empty (= 0x80000000) in VGPRs; int s_r2lim (wave-uniform, SGPR); K via the macro
FGVC_V5_K (5 or 10); FGVC_V5_RELEASE() = the statement that releases the ring slot; constexpr bool do_sel.

    python tools/gen_pair_v5_chain.py        # rewrites the .inc; the build does not run it
"""
import os
import random
import re

HERE = os.path.dirname(os.path.abspath(__file__))
SORTNET = os.path.join(os.path.dirname(HERE), "fgvc_amd", "csrc", "sortnet.hpp")
OUT = os.path.join(os.path.dirname(HERE), "fgvc_amd", "csrc", "pair_v5_chain.inc")
WINDOW = 10          # look-ahead of the list scheduler, in micro-operations of the original order
MIN_AGE = 2          # an operand should be at least this many instructions old


def net(name):
    txt = open(SORTNET).read()
    m = re.search(r"#define %s\(X\) (.*)" % re.escape(name), txt)
    return [(int(a), int(b)) for a, b in re.findall(r"X\((\d+),(\d+)\)", m.group(1))]


def micro_ops(K):
    """[(mnemonic, dst, [operands])]: an operand is a variable name or an int literal; kinds: 'v' VGPR int, 's' SGPR pair
    (compare mask), 'S' SGPR int (read only)"""
    sel = net(f"FGVC_SELNET_16_TOP{K}")
    vm = net(f"FGVC_VMERGE_ASC_{K}")
    ops = []
    kind = {"v_dx0": "v", "v_dy0": "v", "s_r2lim": "S", "v_empty": "v", "v_base": "v"}
    for r in range(16):
        kind[f"ck[{r}]"] = "v"
    for i in range(K):
        kind[f"lk[{i}]"] = "v"
        kind[f"lb[{i}]"] = "v"

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
        op("v_and_or_b32", f"kt{r}", [f"ck[{r}]", -16, 15 - r])
        op("v_cndmask_b32_e64", f"ck[{r}]", ["v_empty", f"kt{r}", f"kc{r}"])
    for i, j in sel:                       # descending: ck[i] >= ck[j]
        op("v_max_i32", f"ck[{i}]", [f"ck[{i}]", f"ck[{j}]"])
        op("v_min_i32", f"ck[{j}]", [f"ck[{i}]", f"ck[{j}]"])
    for i in range(K):                     # top K of (sorted candidates) U (ascending list): V-shaped
        op("v_cmp_gt_i32", f"lc{i}", [f"ck[{i}]", f"lk[{i}]"], "s")
        op("v_max_i32", f"lk[{i}]", [f"ck[{i}]", f"lk[{i}]"])
        op("v_cndmask_b32_e64", f"lb[{i}]", [f"lb[{i}]", "v_base", f"lc{i}"])
    for n, (i, j) in enumerate(vm):        # bitonic merger, ascending: lk[i] <= lk[j], payload carried
        op("v_cmp_gt_i32", f"vc{n}", [f"lk[{i}]", f"lk[{j}]"], "s")
        op("v_min_i32", f"lk[{i}]", [f"lk[{i}]", f"lk[{j}]"])
        op("v_max_i32", f"lk[{j}]", [f"lk[{i}]", f"lk[{j}]"])
        op("v_cndmask_b32_e64", f"lb[{i}]", [f"lb[{i}]", f"lb[{j}]", f"vc{n}"])
        op("v_cndmask_b32_e64", f"lb[{j}]", [f"lb[{j}]", f"lb[{i}]", f"vc{n}"])
    # NOTE the pairs (max, min) and (min, max) above read the OLD values of both wires: versioning below takes care of it
    return ops, kind


def versioned(ops):
    """SSA-rename: a write makes a new version of its variable; operands become (name, version) or ints.  The two halves of a
    compare-exchange read the versions current BEFORE the first half wrote."""
    ver = {}
    out = []
    i = 0
    while i < len(ops):
        mn, dst, args = ops[i]
        group = [ops[i]]
        # a compare-exchange = consecutive operations with identical operand lists writing different wires (max/min, min/max, or
        # the two payload selects of a merger comparator): all of them read the pre-group versions
        while (i + len(group) < len(ops) and mn.split("_")[1] in ("max", "min", "cndmask")
               and sorted(map(str, ops[i + len(group)][2])) == sorted(map(str, args))
               and ops[i + len(group)][0].split("_")[1] in ("max", "min", "cndmask")):
            group.append(ops[i + len(group)])
        snap = dict(ver)
        for g_mn, g_dst, g_args in group:
            srcs = [(a, snap.get(a, 0)) if isinstance(a, str) else a for a in g_args]
            ver[g_dst] = ver.get(g_dst, 0) + 1
            out.append((g_mn, (g_dst, ver[g_dst]), srcs))
        i += len(group)
    return out


def schedule(sops):
    """windowed list scheduling: prefer operations whose operands are at least MIN_AGE instructions old"""
    n = len(sops)
    done_at = {}                           # version -> position in the scheduled stream
    emitted = [False] * n
    order = []
    lo = 0
    while len(order) < n:
        while lo < n and emitted[lo]:
            lo += 1
        best, best_key = None, None
        for i in range(lo, min(n, lo + WINDOW)):
            if emitted[i]:
                continue
            mn, dst, srcs = sops[i]
            vs = [v for v in srcs if isinstance(v, tuple)]
            if any(v[1] > 0 and v not in done_at for v in vs):       # a source version that does not exist yet
                continue
            age = min([len(order) - done_at[v] for v in vs if v[1] > 0] or [99])
            key = (0 if age >= MIN_AGE else 1, i)
            if best_key is None or key < best_key:
                best, best_key = i, key
        assert best is not None
        emitted[best] = True
        done_at[sops[best][1]] = len(order)
        order.append(best)
    return [sops[i] for i in order]


def render_slice(chunk, kind, live_out, ind):
    """one asm statement for a list of scheduled operations.  Values are SSA versions: a version produced inside the statement is an
    output operand; one produced earlier comes in through the C++ variable `name__version`; `live_out` = versions read by later
    statements (or final) -- only those are stored."""
    if not chunk:
        return ""
    produced = {}
    outs, ins, lines = [], [], []
    for mn, dst, srcs in chunk:
        refs = []
        for v in srcs:
            if not isinstance(v, tuple):
                refs.append(str(v))
            elif v in produced:
                refs.append(f"%{produced[v]}")
            else:
                k = kind[v[0]]
                key = (v, "s" if k in ("s", "S") else "v")
                if key not in ins:
                    ins.append(key)
                refs.append("{in%d}" % ins.index(key))
        k = kind[dst[0]]
        produced[dst] = len(outs)
        outs.append(("unsigned long long" if k == "s" else "int", dst, "s" if k == "s" else "v"))
        lines.append(f"{mn} %{produced[dst]}, " + ", ".join(refs))
    body = "\\n\\t".join(lines)
    for i in range(len(ins)):
        body = body.replace("{in%d}" % i, f"%{len(outs) + i}")
    s = ind + "{\n"
    for i, (ctype, _, _) in enumerate(outs):
        s += ind + f"  {ctype} t{i}_;\n"
    s += ind + f'  asm volatile("{body}"\n'
    s += ind + "               : " + ", ".join(f'"=&{c}"(t{i}_)' for i, (_, _, c) in enumerate(outs)) + "\n"
    s += ind + "               : " + ", ".join(f'"{c}"({cvar(v)})' for v, c in ins) + ");\n"
    for i, (_, dst, _) in enumerate(outs):
        if dst in live_out:
            s += ind + f"  {cvar(dst)} = t{i}_;\n"
    s += ind + "}\n"
    return s


FINAL = {}


def cvar(v):
    """C++ lvalue of a version: live-ins (version 0) and the final versions of the state arrays are the kernel's own variables, every
    intermediate version has a generated scalar"""
    name, ver = v
    if ver == 0 or FINAL.get(name) == ver:
        return name
    return re.sub(r"[\[\]]", "_", name) + f"_v{ver}"


def prepare(K):
    ops, kind = micro_ops(K)
    sched = schedule(versioned(ops))
    FINAL.clear()
    for mn, dst, srcs in sched:
        FINAL[dst[0]] = max(FINAL.get(dst[0], 0), dst[1])
    return sched, kind


def declarations(sched, kind, ind):
    ints, masks = [], []
    for mn, dst, srcs in sched:
        c = cvar(dst)
        if c == dst[0] and (dst[0].startswith(("ck[", "lk[", "lb["))):
            continue
        (masks if kind[dst[0]] == "s" else ints).append(c)
    s = ""
    for i in range(0, len(ints), 12):
        s += ind + "int " + ", ".join(f"{v} = 0" for v in ints[i:i + 12]) + ";\n"
    for i in range(0, len(masks), 8):
        s += ind + "unsigned long long " + ", ".join(f"{v} = 0" for v in masks[i:i + 8]) + ";\n"
    return s


def cut(sched, n_slices):
    n = len(sched)
    return [sched[n * i // n_slices: n * (i + 1) // n_slices] for i in range(n_slices)]


def live_outs(chunks):
    """for every chunk: the versions it produces that a LATER chunk reads, or that are final"""
    res = []
    for ci, chunk in enumerate(chunks):
        later = set()
        for c2 in chunks[ci + 1:]:
            for mn, dst, srcs in c2:
                later.update(v for v in srcs if isinstance(v, tuple))
        res.append({dst for mn, dst, srcs in chunk if dst in later or FINAL.get(dst[0]) == dst[1]})
    return res


def chain(K, ind="      "):
    sched, kind = prepare(K)
    chunks = cut(sched, 48)
    lo = live_outs(chunks)
    s = f"{ind}// K = {K}: {len(sched)} vector operations in 48 slices (generated by tools/gen_pair_v5_chain.py -- do not edit)\n"
    s += declarations(sched, kind, ind)
    for j in range(16):
        s += ind + f"// ---- K-16 step {j}\n"
        if j + 2 < 16:                  # fragments of step j + 2 (ring of three)
            s += ind + f"ah[{(j + 2) % 3}] = *reinterpret_cast<const f16x8*>(ka + {32 * (j + 2)});\n"
            s += ind + f"al[{(j + 2) % 3}] = *reinterpret_cast<const f16x8*>(ka + 2 * 256 + {32 * (j + 2)});\n"
        if j == 13:
            s += ind + "FGVC_V5_RELEASE();\n"
        for m in range(3):
            slot = 3 * j + m
            a = f"al[{j % 3}]" if m == 2 else f"ah[{j % 3}]"
            b = f"ql[{j}]" if m == 1 else f"qh[{j}]"
            if slot == 0:
                s += ind + f'asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"({a}), "v"({b}));\n'
            else:
                s += ind + f'asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"({a}), "v"({b}));\n'
            if chunks[slot]:
                s += ind + "if constexpr (do_sel) {\n" + render_slice(chunks[slot], kind, lo[slot], ind + "  ") + ind + "}\n"
    s += ind + f"// vector operations per slice: {[len(c) for c in chunks]}\n"
    return s


def flush(K, ind="    "):
    """the same stream without the MFMAs (last pending tile, ablations)"""
    sched, kind = prepare(K)
    chunks = cut(sched, (len(sched) + 7) // 8)
    lo = live_outs(chunks)
    s = declarations(sched, kind, ind)
    for c, l in zip(chunks, lo):
        s += render_slice(c, kind, l, ind)
    return s


def self_check(K):
    """run the scheduled stream on random integers against a direct evaluation of the selection"""
    sched, kind = prepare(K)
    rng = random.Random(K)
    for _ in range(300):
        env = {"v_dx0": rng.randint(-20, 20), "v_dy0": rng.randint(-20, 20), "s_r2lim": rng.choice([-1, 225, 0x3fffffff]),
               "v_empty": -2 ** 31, "v_base": rng.randint(0, 5000)}
        for r in range(16):
            env[f"ck[{r}]"] = rng.randint(-2 ** 28, 2 ** 28)
        lk = sorted((rng.randint(-2 ** 28, 2 ** 28) & ~15) | rng.randint(0, 15) for _ in range(K))
        for i in range(K):
            env[f"lk[{i}]"] = lk[i]
            env[f"lb[{i}]"] = 10000 + i
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
            elif mn == "v_cmp_gt_i32":
                res = a[0] > a[1]
            elif mn == "v_and_or_b32":
                res = (a[0] & a[1]) | a[2]
            elif mn == "v_cndmask_b32_e64":
                res = a[1] if a[2] else a[0]
            elif mn == "v_max_i32":
                res = max(a[0], a[1])
            elif mn == "v_min_i32":
                res = min(a[0], a[1])
            else:
                raise AssertionError(mn)
            vals[dst] = res
        keys = []
        for r in range(16):
            ok = (env["v_dx0"] + (r & 3)) ** 2 <= env["s_r2lim"] - (env["v_dy0"] + (r >> 2)) ** 2
            keys.append(((env[f"ck[{r}]"] & ~15) | (15 - r)) if ok else -2 ** 31)
        want = sorted([(k, env["v_base"]) for k in keys] + [(lk[i], 10000 + i) for i in range(K)], key=lambda t: t[0])[-K:]
        got = [(vals[(f"lk[{i}]", FINAL[f"lk[{i}]"])], vals[(f"lb[{i}]", FINAL[f"lb[{i}]"])]) for i in range(K)]
        assert [g[0] for g in got] == [w[0] for w in want], (got, want)
        allkeys = keys + lk
        for g, w in zip(got, want):
            if allkeys.count(w[0]) == 1:
                assert g[1] == w[1], (got, want)
    ages = []
    pos = {}
    for p, (mn, dst, srcs) in enumerate(sched):
        ages += [p - pos[v] for v in srcs if isinstance(v, tuple) and v in pos]
        pos[dst] = p
    print(f"K = {K}: scheduled stream verified on 300 random tiles; {len(sched)} operations, "
          f"{sum(1 for a in ages if a < MIN_AGE)} operand reads younger than {MIN_AGE} instructions")


def main():
    txt = "// GENERATED by tools/gen_pair_v5_chain.py -- do not edit.  Included by pair_topk_v5.hip (FGVC_V5_PART = 1: the chain, 2: the flush).\n"
    for K in (5, 10):
        self_check(K)
        txt += f"#if FGVC_V5_K == {K} && FGVC_V5_PART == 1\n" + chain(K) + "#endif\n"
        txt += f"#if FGVC_V5_K == {K} && FGVC_V5_PART == 2\n" + flush(K) + "#endif\n"
    open(OUT, "w").write(txt)
    print("wrote", OUT, len(txt.splitlines()), "lines")


if __name__ == "__main__":
    main()
