#!/usr/bin/env python3
"""Counts, in the compiled ISA of conv64p_kernel, the instructions of every step of the generated schedule (the `; @@step` comments
that MARK() leaves) and writes tools/conv64p_costs.json: {form: {step: instructions}} (maximum over a step's occurrences).
tools/gen_conv64p_sched.py deals the steps into the matrix instructions' gaps by these costs instead of its estimates.
Usage: python tools/conv64p_costs.py   (compiles csrc/conv64.hip to assembly with hipcc; no GPU needed)"""
import json, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "fgvc_amd", "csrc", "conv64.hip")
FORMS = {"plain": "ILb0ELb0ELi1E", "res": "ILb1ELb1ELi1E", "res_nof32": "ILb1ELb0ELi1E", "res_bf16": "ILb1ELb0ELi0E"}
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "conv64.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", out], check=True,
                   stderr=subprocess.DEVNULL, cwd=td)
    text = open(out).read()
costs = {}
for form, sym in FORMS.items():
    start = text.index("_ZN4fgvc14conv64p_kernel%sEEvNS_12Conv64ParamsE:" % sym)
    body = text[start:text.index(".Lfunc_end", start)].split("\n")
    cur, n, c = None, 0, {}
    for line in body:
        t = line.strip()
        mm = re.match(r"; @@(.*)", t)
        if mm:
            if cur and cur != "end":
                key = re.sub(r"\(.*", "", cur) if cur.startswith(("flip", "opread")) else cur
                c[key] = max(c.get(key, 0), n)
            cur, n = mm.group(1).strip(), 0
            continue
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        if t.startswith("v_mfma"):
            continue
        n += 1
    costs[form] = c
    print(form, "steps", len(c), "sum", sum(c.values()), file=sys.stderr)
json.dump(costs, open(os.path.join(root, "tools", "conv64p_costs.json"), "w"), indent=1, sort_keys=True)
