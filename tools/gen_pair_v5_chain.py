"""Generate fgvc_amd/csrc/pair_v5_chain.inc: the 48-MFMA chain of fgvc_pair_topk_f16x3 with the selection of the pending tile cut
into one slice of vector instructions per MFMA, every slice ONE volatile inline-asm statement.

Why generated assembly slices: the slices are pure register arithmetic.  Written as C++, instruction selection places them anywhere
in the loop body (first build: all 300 operations in one gap between two MFMAs); pinned by scheduling barriers and empty asm
fences they keep their gap, but every fence that reads a register a vector operation has just written is padded with wait states
by the compiler (~200 s_nop per tile) and the chain ran at 62 cycles per MFMA (tools/micro/mfma_valu_fill.hip: 4 vector operations
per gap are free, 6 cost 37 cycles per MFMA, 10 cost 61).  Volatile asm statements keep their program order against each other,
need no fences, and get no padding.

The include expects in scope: f32x16 acc; f16x8 ah[3], al[3] (steps 0 and 1 already requested), qh[16], ql[16]; const unsigned char* ka; int ck[16] (raw fixed-point
scores of the pending tile), lk[K], lb[K]; int v_dy0, v_dx0 (key block origin minus query position, per lane); int s_r2lim, s_base
(s_r2lim wave-uniform in an SGPR, v_base and v_empty = 0x80000000 in VGPRs); K via the macro FGVC_V5_K (5 or 10); FGVC_V5_RELEASE() = the statement that releases the ring slot.
Slices (slot = MFMA index): column offsets^2 and row limits, keys (3 ops per candidate), selection network (2 per comparator),
list maximum (3 per entry), bitonic merger with payload (5 per comparator).

    python tools/gen_pair_v5_chain.py        # rewrites the .inc; the build does not run it
"""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
SORTNET = os.path.join(os.path.dirname(HERE), "fgvc_amd", "csrc", "sortnet.hpp")
OUT = os.path.join(os.path.dirname(HERE), "fgvc_amd", "csrc", "pair_v5_chain.inc")


def net(name):
    txt = open(SORTNET).read()
    m = re.search(r"#define %s\(X\) (.*)" % re.escape(name), txt)
    return [(int(a), int(b)) for a, b in re.findall(r"X\((\d+),(\d+)\)", m.group(1))]


class Slice:
    """one asm statement under construction: instruction lines + operand lists"""

    def __init__(self):
        self.lines, self.outs, self.ins, self.post, self.decl, self.vcc = [], [], [], [], [], False
        self.nops = 0
        self.cur = {}             # variable -> placeholder of the temp that holds its newest value inside this statement

    def out(self, var):           # fresh int temp bound to an output operand; returns its %n placeholder name
        t = f"t{len(self.decl)}_"
        self.decl.append(t)
        self.outs.append((t, var))
        self.cur[var] = "{" + t + "}"
        return "{" + t + "}"

    def inp(self, expr, cons="v"):
        if expr in self.cur:      # produced earlier in this statement: read the temp, not the C++ variable
            return self.cur[expr]
        key = (expr, cons)
        if key not in self.ins:
            self.ins.append(key)
        return "{in:" + expr + "}"

    def emit(self, text, n=1):
        self.lines.append(text)
        self.nops += n

    def render(self, ind):
        if not self.lines:
            return ""
        names = {}
        for i, (t, _) in enumerate(self.outs):
            names["{" + t + "}"] = f"%{i}"
        for j, (e, _) in enumerate(self.ins):
            names["{in:" + e + "}"] = f"%{len(self.outs) + j}"
        body = "\\n\\t".join(self.lines)
        for k, v in names.items():
            body = body.replace(k, v)
        outs = ", ".join(f'"=&v"({t})' for t, _ in self.outs)
        ins = ", ".join(f'"{c}"({e})' for e, c in self.ins)
        clob = ' : "vcc"' if self.vcc else ""
        s = ind + "{\n"
        s += ind + "  int " + ", ".join(self.decl) + ";\n"
        s += ind + f'  asm volatile("{body}"\n{ind}               : {outs}\n{ind}               : {ins}{clob});\n'
        for t, var in self.outs:
            s += ind + f"  {var} = {t};\n"
        s += ind + "}\n"
        return s


def macros(K):
    """the selection of one tile as a list of (n_ops, fn(slice)) in dependency order"""
    sel = net(f"FGVC_SELNET_16_TOP{K}")
    vm = net(f"FGVC_VMERGE_ASC_{K}")
    out = []

    def geom_x(a):
        def f(s):
            d = s.inp("v_dx0")
            t = s.out(f"xsq{a}")
            s.emit(f"v_add_u32 {t}, {a}, {d}")
            s.emit(f"v_mul_i32_i24 {t}, {t}, {t}")
        return (2, f)

    def geom_y(a):
        def f(s):
            d, lim = s.inp("v_dy0"), s.inp("s_r2lim", "s")
            t = s.out(f"ylim{a}")
            s.emit(f"v_add_u32 {t}, {a}, {d}")
            s.emit(f"v_mul_i32_i24 {t}, {t}, {t}")
            s.emit(f"v_sub_u32 {t}, {lim}, {t}")
        return (3, f)

    def key(r):
        def f(s):
            raw = s.inp(f"ck[{r}]")
            x, y = s.inp(f"xsq{r & 3}"), s.inp(f"ylim{r >> 2}")
            t = s.out(f"ck[{r}]")
            s.vcc = True
            s.emit(f"v_cmp_le_i32 vcc, {x}, {y}")
            s.emit(f"v_and_or_b32 {t}, {raw}, -16, {15 - r}")
            s.emit(f"v_cndmask_b32 {t}, {s.inp('v_empty')}, {t}, vcc")
        return (3, f)

    def comparator(i, j):         # descending: ck[i] >= ck[j]
        def f(s):
            a, b = s.inp(f"ck[{i}]"), s.inp(f"ck[{j}]")
            hi, lo = s.out(f"ck[{i}]"), s.out(f"ck[{j}]")
            s.emit(f"v_max_i32 {hi}, {a}, {b}")
            s.emit(f"v_min_i32 {lo}, {a}, {b}")
        return (2, f)

    def listmax(i):
        def f(s):
            s.vcc = True
            c, l, b, base = s.inp(f"ck[{i}]"), s.inp(f"lk[{i}]"), s.inp(f"lb[{i}]"), s.inp("v_base")
            nl, nb = s.out(f"lk[{i}]"), s.out(f"lb[{i}]")
            s.emit(f"v_cmp_gt_i32 vcc, {c}, {l}")
            s.emit(f"v_max_i32 {nl}, {c}, {l}")
            s.emit(f"v_cndmask_b32 {nb}, {b}, {base}, vcc")
        return (3, f)

    def vmerge(i, j):             # ascending: lk[i] <= lk[j], payload lb carried
        def f(s):
            s.vcc = True
            a, b = s.inp(f"lk[{i}]"), s.inp(f"lk[{j}]")
            pa, pb = s.inp(f"lb[{i}]"), s.inp(f"lb[{j}]")
            lo, hi, pi, pj = s.out(f"lk[{i}]"), s.out(f"lk[{j}]"), s.out(f"lb[{i}]"), s.out(f"lb[{j}]")
            s.emit(f"v_cmp_gt_i32 vcc, {a}, {b}")
            s.emit(f"v_min_i32 {lo}, {a}, {b}")
            s.emit(f"v_max_i32 {hi}, {a}, {b}")
            s.emit(f"v_cndmask_b32 {pi}, {pa}, {pb}, vcc")
            s.emit(f"v_cndmask_b32 {pj}, {pb}, {pa}, vcc")
        return (5, f)

    for a in range(4):
        out.append(geom_x(a))
    # row limits just ahead of the row's keys (short live ranges)
    for r in range(16):
        if r % 4 == 0:
            out.append(geom_y(r // 4))
        out.append(key(r))
    out += [comparator(i, j) for i, j in sel]
    out += [listmax(i) for i in range(K)]
    out += [vmerge(i, j) for i, j in vm]
    return out


def chain(K, ind="      "):
    ms = macros(K)
    total = sum(n for n, _ in ms)
    slots = [[] for _ in range(48)]
    done = 0
    si = 0
    for n, f in ms:                     # macro goes to the slot its midpoint falls into (near-equal op counts per slot)
        mid = done + n / 2
        si = min(47, int(mid * 48 / total))
        slots[si].append(f)
        done += n
    s = f"{ind}// K = {K}: {total} vector operations in 48 slices (generated by tools/gen_pair_v5_chain.py -- do not edit)\n"
    s += ind + "int xsq0 = 0, xsq1 = 0, xsq2 = 0, xsq3 = 0, ylim0 = 0, ylim1 = 0, ylim2 = 0, ylim3 = 0;\n"
    counts = []
    for j in range(16):
        s += ind + f"// ---- K-16 step {j}\n"
        if j + 2 < 16:                  # fragments of step j + 2 (ring of three: an LDS read has ~190 cycles to land)
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
            sl = Slice()
            for f in slots[slot]:
                f(sl)
            counts.append(sl.nops)
            if sl.lines:
                s += ind + "if constexpr (do_sel) {\n" + sl.render(ind + "  ") + ind + "}\n"
    s += ind + f"// vector operations per slice: {counts}\n"
    return s


def flush(K, ind="    "):
    """the same slices without the MFMAs (last pending tile, ablations)"""
    ms = macros(K)
    s = ind + "int xsq0 = 0, xsq1 = 0, xsq2 = 0, xsq3 = 0, ylim0 = 0, ylim1 = 0, ylim2 = 0, ylim3 = 0;\n"
    for i in range(0, len(ms), 3):
        sl = Slice()
        for n, f in ms[i:i + 3]:
            f(sl)
        s += sl.render(ind)
    return s


def main():
    txt = "// GENERATED by tools/gen_pair_v5_chain.py -- do not edit.  Included twice by pair_topk_v5.hip (FGVC_V5_PART = 1: the chain, 2: the flush).\n"
    for K in (5, 10):
        txt += f"#if FGVC_V5_K == {K} && FGVC_V5_PART == 1\n" + chain(K) + "#endif\n"
        txt += f"#if FGVC_V5_K == {K} && FGVC_V5_PART == 2\n" + flush(K) + "#endif\n"
    open(OUT, "w").write(txt)
    print("wrote", OUT, len(txt.splitlines()), "lines")


if __name__ == "__main__":
    main()
