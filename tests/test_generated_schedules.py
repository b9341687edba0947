"""The generated instruction orders under fgvc_amd/csrc (conv64p_sched_*.inc, conv256p_loop.inc, conv128p_loop.inc) are what their
generators write: an edit of a generator (or of tools/conv64p_costs.json) without the regenerated files -- or the reverse -- fails here.  CPU only."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _regen(tmp_path, script, names):
    work = tmp_path / "repo"
    (work / "tools").mkdir(parents=True)
    (work / "fgvc_amd" / "csrc").mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "tools", script), work / "tools" / script)
    costs = os.path.join(ROOT, "tools", "conv64p_costs.json")
    if os.path.exists(costs):
        shutil.copy(costs, work / "tools" / "conv64p_costs.json")
    subprocess.run([sys.executable, str(work / "tools" / script)], check=True, capture_output=True)
    for n in names:
        want = (work / "fgvc_amd" / "csrc" / n).read_text()
        have = open(os.path.join(ROOT, "fgvc_amd", "csrc", n)).read()
        assert want == have, f"{n} is not what tools/{script} writes: regenerate it"


def test_conv64p_schedules_are_current(tmp_path):
    _regen(tmp_path, "gen_conv64p_sched.py", ["conv64p_sched_plain.inc", "conv64p_sched_res.inc", "conv64p_sched_res_bf16.inc"])


def test_pair_v8_statements_are_current(tmp_path):
    """pair_v8.inc (the one-statement tiles of the opt-in pair kernel): tools/gen_pair_v8.py reads the comparator lists of csrc/sortnet.hpp
    through tools/gen_pair_v5_chain.py, verifies its register allocation on 300 random tiles per K, and must write the file byte for byte
    on every run (its allocator once iterated a set of tuples: the statement changed from run to run)"""
    work = tmp_path / "repo"
    (work / "tools").mkdir(parents=True)
    (work / "fgvc_amd" / "csrc").mkdir(parents=True)
    for f in ("gen_pair_v8.py", "gen_pair_v5_chain.py"):
        shutil.copy(os.path.join(ROOT, "tools", f), work / "tools" / f)
    shutil.copy(os.path.join(ROOT, "fgvc_amd", "csrc", "sortnet.hpp"), work / "fgvc_amd" / "csrc" / "sortnet.hpp")
    have = open(os.path.join(ROOT, "fgvc_amd", "csrc", "pair_v8.inc")).read()
    for seed in ("0", "12345"):
        subprocess.run([sys.executable, str(work / "tools" / "gen_pair_v8.py")], check=True, capture_output=True, env={**os.environ, "PYTHONHASHSEED": seed})
        assert (work / "fgvc_amd" / "csrc" / "pair_v8.inc").read_text() == have, "pair_v8.inc is not what tools/gen_pair_v8.py writes: regenerate it"


def test_conv256p_loops_are_current(tmp_path):
    _regen(tmp_path, "gen_conv256p_sched.py", ["conv256p_loop.inc", "conv128p_loop.inc"])


def test_vmcnt_waits_fit_the_instruction(tmp_path):
    """s_waitcnt vmcnt(N) encodes N in 6 bits on gfx9"""
    import re
    for n in ("conv64p_sched_plain.inc", "conv64p_sched_res.inc", "conv64p_sched_res_bf16.inc", "conv256p_loop.inc", "conv128p_loop.inc"):
        text = open(os.path.join(ROOT, "fgvc_amd", "csrc", n)).read()
        for m in re.finditer(r"vmcnt\((\d+)\)", text):
            assert int(m.group(1)) < 64, (n, m.group(0))
        for m in re.finditer(r"C64P_N\((\d+), (\d+)\)", text):
            assert int(m.group(1)) + int(m.group(2)) < 64, (n, m.group(0))
