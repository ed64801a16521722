"""VGPR / SGPR / LDS / occupancy table of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python scripts/kernel_resources.py gut_render.hip [filter-substring] [extra hipcc flags ...]
"""
import importlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("3dgrut_amd.build")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", re.sub(r"grut::\(anonymous namespace\)::", "", o)).replace("void ", "") for o in out]


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    cmd = [b._hipcc(), "-x", "hip", *b.FLAGS, *b.unit_flags(src), *extra, "-Rpass-analysis=kernel-resource-usage", "-c",
           os.path.join(b.CSRC, src), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    names = demangle([r["name"] for r in rows])
    print(f"{'kernel':70s} vgpr agpr sgpr occ    lds scratch")
    for r, n in zip(rows, names):
        if flt in n:
            print(f"{n[:70]:70s} {r.get('vgpr', -1):4d} {r.get('agpr', -1):4d} {r.get('sgpr', -1):4d} {r.get('occ', -1):3d} {r.get('lds', -1):6d} {r.get('scratch', -1):5d}")


if __name__ == "__main__":
    main()
