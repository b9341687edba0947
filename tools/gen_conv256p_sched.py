#!/usr/bin/env python3
"""Writes fgvc_amd/csrc/conv256p_loop.inc: the main loop of conv256p_kernel (csrc/conv_split.hip) as ONE assembly statement.

conv256p_kernel = the 256 -> 256 channel 3 x 3 convolution of the f16 + FP6 arithmetic with ONE wave per SIMD: a wave owns 128 output
channels x 4 pixel rows x 32 pixels (16 accumulator tiles = all 256 accumulation registers) and runs a stage -- one tap of one 32-channel
input chunk: 48 matrix instructions, 1 536 cycles of pipe -- as a fixed stream, the operand reads, the weight ring's DMA and the next
chunk's patch dealt into the matrix instructions' gaps (a wave alone on its SIMD issues one instruction per ~4 cycles; 32 cycles per
matrix instruction leave ~6 slots).  conv_split_kernel's two waves per SIMD each spent 1 600-2 150 cycles on 24 matrix instructions
(768 cycles of pipe), one after the other.

Why assembly for the whole loop: the first build left register allocation to the compiler, the asynchronously written registers named
by operand constraints (as conv64p_kernel does).  With 256 accumulation registers and 212 vector registers named it spilled
accumulators around the statements that use them (2 000 spills).  Here nothing is left to allocate: the register map below is the
kernel's, the compiler sees one statement that clobbers it.

The nine stage bodies of a chunk differ in the tap's patch rows and column shift (immediates of the operand reads), in the ring slot
(tap % 3) and in what else rides along: taps 5-7 load the next chunk's patch into registers, tap 8 writes it to the LDS (its own
operands were read during tap 7).  Every vector-memory instruction is issued in every chunk, so the s_waitcnt vmcnt(N) are exact; this
script counts them.

Register map
  a[0:255]    accumulators [channel tile a][pixel row r] x 16
  v[0:63]     pixel fragments, buffer 0: row r at 16 r: [f16 k-step 0 | k-step 1 | FP6 first 16 B | second 16 B]; v[64:127] buffer 1
  v[128:143]  weight fragments, buffer 0 (same four pieces); v[144:159] buffer 1
  v[160:211]  the next chunk's patch in flight (13 pieces of 4)
  v[212:215]  weight-fragment addresses of the stage (slot variants), v[216:218] pixel-fragment slot variants, v220 / v221 temporaries
  v222 / v223 patch lane offset and piece table in use (copies of v241 / v243; the second input's v244 / v245 from its first prefetch on)
  v232 wl  v[233:238] pl[dx][par]  v239 / v240 DMA lane offsets  v241 patch lane offset  v242 16 lane  v243 piece table      (inputs)
  s[20:21] weight base  s22 tap stride  s23 chunk stride  s[24:25] patch base  s26 chunks  s27 ring  s28 patch  s29 piece offset (inputs)
  s46 the last chunk's index when a second input follows (else -1)  s[48:49] its weights  s[50:51] its patch base  s52 its chunks  (inputs)
  s30 chunk  s[32:33] slab base  s[34:35] / s36 / s37 temporaries  s[38:39] next patch base  s40 ring slot of the DMA
"""
import os
import sys

CAP = 6
NA = 4                      # output-channel tiles per wave (4: 256 channels per workgroup, 2: 128)
SLOTB = 256 * 128


def areg(buf, what):
    base = 128 + 16 * buf
    return {"f0": base, "f1": base + 4, "xl": base + 8, "xh": base + 12}[what]


def breg(buf, r, what):
    base = 64 * buf + 16 * r
    return {"f0": base, "f1": base + 4, "xl": base + 8, "xh": base + 12}[what]


def v4(b):
    return f"v[{b}:{b + 3}]"


def RA(a, buf):
    return [f"ds_read_b128 {v4(areg(buf, w))}, v{212 + i} offset:{a * 4096}" for i, w in enumerate(("f0", "f1", "xl", "xh"))]


def RB(r, buf, t):
    dy, dx = t // 3, t % 3
    off = (r + dy) * 40 * 128
    pb0 = 233 + dx * 2 + ((r + dy) & 1)
    out = [f"v_xor_b32 v216, 32, v{pb0}", f"v_xor_b32 v217, 64, v{pb0}", f"v_xor_b32 v218, 0x60, v{pb0}"]
    for w, addr in (("f0", pb0), ("f1", 216), ("xl", 217), ("xh", 218)):
        out.append(f"ds_read_b128 {v4(breg(buf, r, w))}, v{addr} offset:{off}")
    return out


def M(kind, a, r, abuf, bbuf):
    acc = "a[%d:%d]" % ((a * 4 + r) * 16, (a * 4 + r) * 16 + 15)
    if kind in ("F0", "F1"):
        w = "f0" if kind == "F0" else "f1"
        return f"v_mfma_f32_32x32x16_f16 {acc}, {v4(areg(abuf, w))}, {v4(breg(bbuf, r, w))}, {acc}"
    ax, bx = areg(abuf, "xl"), breg(bbuf, r, "xl")
    return f"v_mfma_scale_f32_32x32x64_f8f6f4 {acc}, v[{ax}:{ax + 5}], v[{bx}:{bx + 5}], {acc}, v{ax + 6}, v{bx + 6} op_sel_hi:[0,0,0] cbsz:2 blgp:2"


def WB(t):
    """s[32:33] = base of the weight slab of stage q + 2, s40 = its ring slot's LDS address"""
    out = []
    if t < 7:
        out += [f"s_mul_i32 s36, s22, {t + 2}", "s_mul_i32 s37, s23, s30"]
    else:                                   # the next chunk's tap t - 7 (the last chunk: its own again, or the second input's slab t - 7)
        out += ["s_add_i32 s37, s30, 1", "s_sub_i32 s36, s26, 1", "s_min_i32 s37, s37, s36", "s_mul_i32 s37, s23, s37", f"s_mul_i32 s36, s22, {t - 7}"]
    out += ["s_add_i32 s36, s36, s37", "s_add_u32 s32, s20, s36", "s_addc_u32 s33, s21, 0", f"s_add_i32 s40, s27, {((t + 2) % 3) * SLOTB}"]
    if t >= 7:
        out += ["s_cmp_eq_u32 s30, s46", f"s_cbranch_scc0 7{t}f", "s_sub_i32 s36, s52, 1", f"s_min_i32 s36, s36, {t - 7}", "s_mul_i32 s36, s36, s23",
                "s_add_u32 s32, s48, s36", "s_addc_u32 s33, s49, 0", f"7{t}:"]
    return out


def WP(j):
    return [f"s_add_i32 s36, s29, {j * 1024}", "s_add_u32 s34, s32, s36", "s_addc_u32 s35, s33, 0", "s_add_i32 m0, s36, s40", "s_nop 0",
            f"global_load_lds_dwordx4 v{239 + (j & 1)}, s[34:35]"]


def PB():
    return ["s_add_i32 s37, s30, 1", "s_sub_i32 s36, s26, 1", "s_min_i32 s37, s37, s36", "s_lshl_b32 s37, s37, 7", "s_add_u32 s38, s24, s37", "s_addc_u32 s39, s25, 0",
            "s_cmp_eq_u32 s30, s46", "s_cbranch_scc0 75f", "s_mov_b64 s[38:39], s[50:51]", "v_mov_b32 v222, v244", "v_mov_b32 v223, v245", "75:"]


def PL(k):
    return [f"v_readlane_b32 s36, v223, {k}", f"v_readlane_b32 s37, v223, {32 + k}", "s_add_u32 s34, s38, s36", "s_addc_u32 s35, s39, 0",
            "v_xor_b32 v220, s37, v222", f"global_load_dwordx4 {v4(160 + 4 * k)}, v220, s[34:35]"]


def PW(k, n):
    return [f"s_waitcnt vmcnt({n})", f"v_readlane_b32 s36, v223, {16 + k}", "s_add_i32 s36, s36, s28", "v_add_u32 v221, s36, v242",
            f"ds_write_b128 v221, {v4(160 + 4 * k)}"]


def generate(na, fname):
    global NA, SLOTB
    NA, SLOTB = na, na * 64 * 128
    NS = NA * 12                                            # matrix instructions per stage
    PPW = NA * 2                                            # weight DMA pieces per wave and stage
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fgvc_amd", "csrc")
    bodies, vm_seq = [], []
    for t in range(9):
        cur, nxt = t & 1, (t + 1) & 1
        sites = [(kind, a, r) for a in range(NA) for kind in ("F0", "F1", "X") for r in range(4)]
        side = []                                           # (earliest site, instructions, vector-memory tag)
        for a in range(NA - 1):
            side.append((a * 12, RA(a + 1, (a + 1) & 1), None))
        if t < 8:
            for r in range(4):
                side.append((r * (NS // 4) + 1, RB(r, nxt, t + 1), None))
        side.append((2, WB(t), None))
        for j in range(PPW):
            side.append((4 + j * ((NS - 8) // PPW), WP(j), ("w", t)))
        if t in (5, 6, 7):
            ks = {5: range(0, 5), 6: range(5, 9), 7: range(9, 13)}[t]
            if t == 5:
                side.append((NS // 3, PB(), None))
            for i, k in enumerate(ks):
                side.append((NS // 2 + i * (NS // 12 + 1), PL(k), ("p", k)))
        if t == 8:
            for k in range(13):
                side.append((2 + k * max(1, (NS - 6) // 13), PW(k, f"@PWN{k}@"), None))
        side.sort(key=lambda x: x[0])
        lines = [f"; ---- tap {t}", f"s_waitcnt vmcnt(@TOPN{t}@)", "s_waitcnt lgkmcnt(0)", "s_barrier",
                 f"v_add_u32 v212, {(t % 3) * SLOTB}, v232", "v_xor_b32 v213, 32, v212", "v_xor_b32 v214, 64, v212", "v_xor_b32 v215, 0x60, v212"]
        lines += RA(0, 0)
        if t == 0:
            for r in range(4):
                lines += RB(r, 0, 0)
        lines.append("s_waitcnt lgkmcnt(0)")
        si, order, carry = 0, [], 0
        for n, (kind, a, r) in enumerate(sites):
            if n % 12 == 0 and n > 0:
                lines.append("s_waitcnt lgkmcnt(0)")        # tile a's fragments (read during tile a - 1)
            lines.append(M(kind, a, r, a & 1, cur))
            left = CAP - carry
            carry = 0
            while si < len(side) and side[si][0] <= n and (len(side[si][1]) <= left + 1 or left >= CAP - 1):
                lines += side[si][1]
                if side[si][2]:
                    order.append(side[si][2])
                left -= len(side[si][1])
                si += 1
            if left < 0:
                carry = min(-left, CAP)
        while si < len(side):
            lines += side[si][1]
            if side[si][2]:
                order.append(side[si][2])
            si += 1
        bodies.append(lines)
        vm_seq.append(order)
    flat = []
    for t in range(9):
        flat += [(t, kind, x) for (kind, x) in vm_seq[t]]
    n_all = len(flat)

    def after(idx_last, upto_tap):
        cnt, i = 0, idx_last + 1
        while True:
            if i == n_all:
                i = 0
            if flat[i][0] == upto_tap and flat[i - 1][0] != upto_tap:
                return cnt
            cnt += 1
            i += 1
    subst = {}
    for t in range(9):
        src = (t - 2) % 9
        last = max(i for i, f in enumerate(flat) if f[0] == src and f[1] == "w")
        subst[f"@TOPN{t}@"] = str(after(last, t))
    for k in range(13):
        idx = next(i for i, f in enumerate(flat) if f[1] == "p" and f[2] == k)
        subst[f"@PWN{k}@"] = str(sum(1 for i in range(idx + 1, n_all) if flat[i][0] < 8))
    out = ["; zero the accumulators"] + [f"v_accvgpr_write_b32 a{i}, 0" for i in range(NA * 64)]
    out += ["s_mov_b32 s30, 0", "s_mov_b64 s[38:39], s[24:25]", "v_mov_b32 v222, v241", "v_mov_b32 v223, v243", "s_memtime s[42:43]", "1:"]
    for t in range(9):
        out += bodies[t]
    out += ["s_add_i32 s30, s30, 1", "s_cmp_lt_i32 s30, s26", "s_cbranch_scc1 1b"]
    # ---- the second input (the block's projection shortcut folded in): one stage per 32-channel chunk, its only tap the centre one.  The
    # patch changes with every stage: a second barrier behind the operand reads frees it, the next chunk's pieces are loaded early in
    # the stage and written late.  (s30 = chunk of the second input, s41 = its ring slot's offset, s[38:39] = its patch base.)
    x2 = ["s_cmp_eq_u32 s52, 0", "s_cbranch_scc1 3f", "s_mov_b32 s30, 0", "s_mov_b32 s41, 0", "2:",
          "s_waitcnt vmcnt(9)", "s_waitcnt lgkmcnt(0)", "s_barrier",
          "v_add_u32 v212, s41, v232", "v_xor_b32 v213, 32, v212", "v_xor_b32 v214, 64, v212", "v_xor_b32 v215, 0x60, v212"]
    x2 += RA(0, 0)
    for r in range(4):
        x2 += RB(r, 0, 4)
    x2 += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    sites = [(kind, a, r) for a in range(NA) for kind in ("F0", "F1", "X") for r in range(4)]
    side = [(a * 12, RA(a + 1, (a + 1) & 1)) for a in range(NA - 1)]
    # next chunk of the second input (the last: itself again): patch base, then the 13 pieces; weights of the stage two ahead
    side.append((1, ["s_add_i32 s37, s30, 1", "s_sub_i32 s36, s52, 1", "s_min_i32 s37, s37, s36", "s_lshl_b32 s37, s37, 7", "s_add_u32 s38, s50, s37",
                     "s_addc_u32 s39, s51, 0"]))
    for k in range(13):
        side.append((2 + k, PL(k)))
    side.append((16, ["s_add_i32 s37, s30, 2", "s_sub_i32 s36, s52, 1", "s_min_i32 s37, s37, s36", "s_mul_i32 s36, s37, s23", "s_add_u32 s32, s48, s36",
                      "s_addc_u32 s33, s49, 0", "s_add_i32 s40, s41, {0}".format(2 * SLOTB), "s_cmp_ge_u32 s40, {0}".format(3 * SLOTB),
                      "s_cselect_b32 s37, {0}, 0".format(3 * SLOTB), "s_sub_i32 s40, s40, s37", "s_add_i32 s40, s40, s27"]))
    for j in range(NA * 2):
        side.append((18 + j * 2, WP(j)))
    for k in range(13):
        side.append((NS - 14 + k, PW(k, 8 + 12 - k)))
    side.sort(key=lambda x: x[0])
    si = 0
    for n, (kind, a, r) in enumerate(sites):
        if n % 12 == 0 and n > 0:
            x2.append("s_waitcnt lgkmcnt(0)")
        x2.append(M(kind, a, r, a & 1, 0))
        while si < len(side) and side[si][0] <= n:
            x2 += side[si][1]
            si += 1
    while si < len(side):
        x2 += side[si][1]
        si += 1
    x2 += ["s_add_i32 s41, s41, {0}".format(SLOTB), "s_cmp_ge_u32 s41, {0}".format(3 * SLOTB), "s_cselect_b32 s37, {0}, 0".format(3 * SLOTB), "s_sub_i32 s41, s41, s37",
           "s_add_i32 s30, s30, 1", "s_cmp_lt_i32 s30, s52", "s_cbranch_scc1 2b", "3:"]
    out += x2
    out += ["s_nop 15", "s_nop 15", "s_memtime s[44:45]",
            "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_sub_u32 s42, s44, s42", "s_subb_u32 s43, s45, s43"]          # s[42:43] = cycles of the loop
    text = "\n".join(out)
    for k, v in subst.items():
        text = text.replace(k, v)
    clob = [f"v{i}" for i in range(0, 224)] + [f"a{i}" for i in range(NA * 64)] + [f"s{i}" for i in range(30, 42)] + ["s44", "s45"] + ["m0", "vcc", "scc", "memory"]
    with open(os.path.join(root, fname), "w") as f:
        f.write("// GENERATED by tools/gen_conv256p_sched.py -- do not edit.  The main loop of conv256p_kernel: one assembly statement.\n")
        f.write("asm volatile(\n")
        for line in text.split("\n"):
            f.write(f'    "{line}\\n\\t"\n')
        f.write("    : \"={s[42:43]}\"(loop_cycles)\n    : C256P_INPUTS\n    : " + ", ".join(f'"{c}"' for c in clob) + ");\n")
    n_instr = sum(1 for line in text.split("\n") if line and not line.startswith(";") and not line.endswith(":"))
    print(fname, "instructions", n_instr, "tops", [subst[f"@TOPN{t}@"] for t in range(9)], "writes", [subst[f"@PWN{k}@"] for k in range(13)], file=sys.stderr)


def main():
    generate(4, "conv256p_loop.inc")
    generate(2, "conv128p_loop.inc")


if __name__ == "__main__":
    main()
