"""The code-object notes of every kernel in fgvc_amd/lib/libfgvc_hip.so: registers, spills, scratch, LDS -- what the compiler actually
allocated, read from the built library with the LLVM tools of the ROCm image (no GPU needed).

    python tools/kernel_notes.py [substring]        # one line per kernel whose (mangled) name contains the substring
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")


def kernel_notes(lib=None):
    """{mangled kernel name: {field: int}} for the gfx950 code objects embedded in `lib` (one bundle per translation unit)."""
    lib = lib or os.path.join(ROOT, "fgvc_amd", "lib", "libfgvc_hip.so")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "copy.so")], check=True, capture_output=True)
        data = open(fat, "rb").read()
        pos = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)] + [len(data)]
        for i in range(len(pos) - 1):
            b, co = os.path.join(tmp, f"b{i}.bin"), os.path.join(tmp, f"co{i}.o")
            open(b, "wb").write(data[pos[i]:pos[i + 1]])
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={b}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:|\n\s+- \.args:", notes)[1:]:
                m = re.search(r"\.name:\s+(\S+)", blk)
                if not m:
                    continue
                d = {}
                for f in FIELDS:
                    mm = re.search(r"\." + f + r":\s+(\d+)", ("\n    .agpr_count:" + blk) if f == "agpr_count" else blk)
                    if mm:
                        d[f] = int(mm.group(1))
                out[m.group(1)] = d
    return out


if __name__ == "__main__":
    sub = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, d in sorted(kernel_notes().items()):
        if sub in name:
            print(name, " ".join(f"{k.replace('_count', '').replace('_fixed_size', '')}={v}" for k, v in d.items()))
