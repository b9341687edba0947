#!/usr/bin/env python3
"""Writes fgvc_amd/csrc/conv64p_sched_{plain,res}.inc: the instruction ORDER of conv64p_kernel's tile loop (csrc/conv64.hip).

A wave of that kernel is the only one on its SIMD (its 288 weight registers fill the file) and issues one instruction per ~4 cycles; a
matrix instruction keeps the pipe busy for 32 (f16 32x32x16) or 64 cycles (fp8 32x32x64), so ~5 resp. ~13 other instructions issued
right behind it cost nothing and everything else costs its full issue time.  The tile loop therefore is a fixed list of 108 matrix
instructions with the rest of the work -- operand reads, the next patch's DMA, the residual's loads, and the EPILOGUE OF THE PREVIOUS
TILE -- dealt into the gaps.  This script does the dealing (greedy, in stream order, under the capacities below) and counts the
vector-memory instructions between a load and the wait for it, so that every s_waitcnt vmcnt(N) in the loop is exact.

Vocabulary of the output (macros / lambdas defined in conv64.hip before the #include):
  GB(q)                       top of operand group q: wait for its operands (read during group q - 1)
  MF(kind, b, t, c, part, buf) one matrix instruction; kind F0 / F (f16, first of its accumulator or not), XA / XV (fp8, weights in
                              accumulation / vector registers)
  SB                          end of a gap (sched_barrier)
  everything else             a step of one of the streams (see STREAMS below)
Section 1 = the loop body, section 2 = the drain (the last tile's epilogue after the loop).
"""
import os
import sys

CAP_F16, CAP_FP8 = 5, 13
LDS_GAP = 3
OPREAD_SITE = 0        # the next group's operand reads ride behind this matrix instruction of a group (see conv64.hip: 0 overwrote operands in use)


def groups():
    out = []
    for q in range(24):
        r, dx, c = q // 6, (q % 6) // 2, q % 2
        bs = [b for b in (0, 1) if 0 <= r - b <= 2]
        sites = []
        for part in range(3):
            for b in bs:
                t = (r - b) * 3 + dx
                first = (r - b == 0 and dx == 0 and c == 0 and part == 0)
                kind = ("F0" if first else "F") if part < 2 else ("XA" if t < 7 else "XV")
                sites.append(dict(kind=kind, b=b, t=t, c=c, part=part, cap=CAP_F16 if part < 2 else CAP_FP8, fp8=part == 2))
        out.append(dict(q=q, r=r, dx=dx, c=c, sites=sites))
    return out


class Step:
    def __init__(self, text, cost, vmem=0, f32=0, boundary=False, fp8=False, min_group=0, tag=None):
        self.text, self.cost, self.vmem, self.f32 = text, cost, vmem, f32
        self.boundary, self.fp8, self.min_group, self.tag = boundary, fp8, min_group, tag


def split_steps(j, fmt1):
    if fmt1:
        return [Step(f"e_sp1({j});", 4), Step(f"e_sp2({j});", 3), Step(f"e_sp3({j});", 4), Step(f"e_sp4({j});", 6), Step(f"e_sp5({j});", 4),
                Step(f"e_sp6({j});", 4), Step(f"e_sp7({j});", 4), Step(f"e_sp8({j});", 6), Step(f"e_sp9({j});", 6), Step(f"e_spw({j});", 3)]
    return [Step(f"e_sp1({j});", 6), Step(f"e_sp2({j});", 4), Step(f"e_sp3({j});", 4), Step(f"e_sp4({j});", 6), Step(f"e_spw({j});", 2)]


def epilogue_stream(res, fmt1):
    """the previous tile's epilogue, in order; `boundary` = the next step reads what this one wrote to the LDS"""
    s = [Step("e_desc_s(0);", 9), Step("e_desc_s(1);", 5), Step("e_desc_f(0);", 9), Step("e_desc_f(1);", 5)]
    if res:
        s += [Step("e_desc_r(0);", 9), Step("e_desc_r(1);", 5)]
    s.append(Step("e_bias();", 4, boundary=True))
    for b in (0, 1):
        if res:
            for g in range(4):
                s.append(Step(f"e_fma({b}, {g});", 5))
            s[-1].boundary = True
            s.append(Step(f"e_uread({b});", 4, boundary=True))
            s.append(Step(f"WAITRES({b});", 1, tag=("waitres", b)))
            for i in range(4):
                s.append(Step(f"e_res({b}, {i});", 4))
                s.append(Step(f"e_relu({i});", 4))
                s.append(Step(f"e_stf({b}, {i});", 2, vmem=0, f32=1))
                s += split_steps(i, fmt1)
            s.append(Step(f"e_ovf({b});", 6, boundary=True))
            for i in range(4):                         # this tile's residual rows: the registers are free from here
                s.append(Step(f"resload({b}, {i});", 2, vmem=1, tag=("resload", b, i)))
        else:
            for g in range(4):
                s.append(Step(f"e_fma({b}, {g});", 4))
                s.append(Step(f"e_relu({g});", 4))
                s += split_steps(g, fmt1)
            s.append(Step(f"e_ovf({b});", 6, boundary=True))
        s.append(Step(f"e_rread({b});", 4, boundary=True))
        for i in range(4):
            s.append(Step(f"e_sts({b}, {i});", 2, vmem=1))
    return s


COSTS = {}
DYNAMIC = {"e_ovf(0)": 10, "e_ovf(1)": 10}          # steps whose static instruction count contains a rarely taken path


def load_costs(forms):
    """measured instruction counts of the steps (tools/conv64p_costs.py), the maximum over the kernel forms that share a schedule"""
    import json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv64p_costs.json")
    COSTS.clear()
    if os.path.exists(path):
        data = json.load(open(path))
        for f in forms:
            for k, v in data.get(f, {}).items():
                COSTS[k] = max(COSTS.get(k, 0), v)
    COSTS.update(DYNAMIC)


def measured(step):
    key = step.text.rstrip(";")
    if key.startswith(("flip", "opread")):
        key = key.split("(")[0]
    if key in COSTS:
        step.cost = max(1, COSTS[key])
    return step


def build(res, fmt1=True):
    G = groups()
    epi = [measured(e) for e in epilogue_stream(res, fmt1)]
    dma = []
    for k in range(15):
        dma += [measured(Step(f"dma_a({k});", 6)), measured(Step(f"dma_b({k});", 4, vmem=1, fp8=True, tag=("dma", k)))]
    flips0 = [Step(f"flip({dx}, 0, {k});", 1, min_group=17) for dx in range(3) for k in range(4)]
    flips1 = [Step(f"flip({dx}, 1, {k});", 1, min_group=23) for dx in range(3) for k in range(4)]
    misc = [measured(Step("tilenext_a();", 14)), measured(Step("tilenext_b();", 12))] + flips0 + flips1
    lines, order = [], []          # order: every emitted step with vmem counts, for the wait arithmetic
    ei = di = mi = 0
    site_no, epi_ready = 0, 0      # a step behind a boundary starts at least LDS_GAP matrix instructions later (the LDS round trip)
    for g in G:
        q = g["q"]
        lines.append(f"GB({q})")
        for si, st in enumerate(g["sites"]):
            site_no += 1
            buf = "a" if q % 2 == 0 else "b"
            lines.append(f"  MF_{st['kind']}({st['b']}, {st['t']}, {st['c']}, {st['part']}, {buf});")
            left = st["cap"]
            here = []
            if si == OPREAD_SITE and q + 1 < 24:
                here.append(Step(f"opread({q + 1});", 4))
                left -= 4
            # bookkeeping first (the next tile's origin is needed by the first DMA piece)
            while mi < len(misc) and misc[mi].min_group <= q and (misc[mi].cost <= left or (left >= 4 and mi < 2)):
                here.append(misc[mi]); left -= misc[mi].cost; mi += 1
            # a DMA piece: its constants behind any matrix instruction, the instruction itself (long at issue) behind an fp8 one
            if di < len(dma) and mi >= 2 and left >= 4 and (st["fp8"] or not dma[di].fp8):
                here.append(dma[di]); left -= dma[di].cost; di += 1
                if st["fp8"] and di < len(dma) and dma[di].fp8:
                    here.append(dma[di]); left -= dma[di].cost; di += 1
            while ei < len(epi):
                e = epi[ei]
                if site_no < epi_ready:
                    break
                if e.cost > left + 1 and not (st['fp8'] and left >= st['cap'] - 4 and e.cost > CAP_F16 + 1):
                    break      # (a step too long for any gap goes behind an fp8 instruction that has little else)
                here.append(e); left -= e.cost; ei += 1
                if e.boundary:
                    epi_ready = site_no + LDS_GAP
                    break
            for e in here:
                lines.append(f"    MARK(\"{e.text.rstrip(';')}\") {e.text}")
                order.append(e)
            lines.append("  MARK(\"end\") SB")
    assert di == len(dma) and mi == len(misc), (di, mi)
    leftover = epi[ei:]
    if leftover:                                           # what did not fit rides behind the last matrix instruction, exposed
        lines.append(f"  // {len(leftover)} steps beyond the last gap")
        for e in leftover:
            lines.append(f"    MARK(\"{e.text.rstrip(';')}\") {e.text}" + ("  LB" if e.boundary else ""))
            order.append(e)
        lines.append("  MARK(\"end\")")
    # ---- the wait arithmetic: all vector-memory instructions of the loop are issued unconditionally, in this order, every tile
    def vm_after(idx_from, idx_to_wrapped):
        """(always, f32-only) counts of vector-memory instructions issued after order[idx_from] up to (not including) the step at
        idx_to_wrapped, which lies in the NEXT tile when it is <= idx_from"""
        seq = order[idx_from + 1:] + order[:idx_to_wrapped] if idx_to_wrapped <= idx_from else order[idx_from + 1:idx_to_wrapped]
        return sum(e.vmem for e in seq), sum(e.f32 for e in seq)
    idx = {e.tag: k for k, e in enumerate(order) if e.tag}
    defs = []
    last_dma = idx[("dma", 14)]
    a = sum(e.vmem for e in order[last_dma + 1:]); f = sum(e.f32 for e in order[last_dma + 1:])
    defs.append(f"#define C64P_VM_END C64P_N({a}, {f})      // vector-memory instructions behind the last DMA piece: the wait before the tile's barrier")
    if res:
        for b in (0, 1):
            a, f = vm_after(idx[("resload", b, 3)], idx[("waitres", b)])
            defs.append(f"#define C64P_VM_RES{b} C64P_N({a}, {f})     // ... between the last load of residual row {b} and its first use in the next tile")
    total = sum(e.cost for e in order)
    return lines, defs, total, len(leftover)


def drain(res, fmt1=True):
    out = ["DRAIN_BEGIN"]
    for e in epilogue_stream(res, fmt1):
        if e.text.startswith(("resload", "WAITRES")):
            continue
        out.append(f"  {e.text}" + ("  LB" if e.boundary else ""))
    return out


def main():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fgvc_amd", "csrc")
    for name, res, fmt1, forms in (("plain", False, True, ["plain"]), ("res", True, True, ["res", "res_nof32"]), ("res_bf16", True, False, ["res_bf16"])):
        load_costs(forms)
        lines, defs, total, left = build(res, fmt1)
        path = os.path.join(root, f"conv64p_sched_{name}.inc")
        with open(path, "w") as f:
            f.write(f"// GENERATED by tools/gen_conv64p_sched.py -- do not edit.  {total} side instructions (estimated) in the gaps of 108 matrix\n"
                    f"// instructions, {left} epilogue steps beyond the last gap.\n")
            f.write("#if C64P_SECTION == 0\n" + "\n".join(defs) + "\n")
            f.write("#elif C64P_SECTION == 1\n" + "\n".join(lines) + "\n")
            f.write("#elif C64P_SECTION == 2\n" + "\n".join(drain(res, fmt1)) + "\n#endif\n")
        print(path, "side instructions", total, "left over", left, file=sys.stderr)
        for d in defs:
            print("   ", d, file=sys.stderr)


if __name__ == "__main__":
    main()
