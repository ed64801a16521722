"""VALU issue-time model of the compositing kernels from the chip's own calibration (profiles/r02a_valu_calib.json, scripts/valu_calib.hip).

    python scripts/valu_model.py  ->  profiles/valu_model.json

For each kernel the STATIC instruction mix of its compiled code (hipcc -S, the build's own flags) is split into the classes the
calibration measured — plain VOP1/VOP2, three-source (v_fma / v_fmac / v_min3 ...), packed fp32 (v_pk_*), transcendental
(v_exp / v_rcp / v_sqrt / v_rsq / v_log), DPP forms, compares — and priced with the measured SIMD occupancy of a wave64 instruction of
that class at saturation (4 waves per SIMD, all 1024 SIMDs busy; wall clock, so the chip's real clock under a VALU-bound load is
in the number).  bench.py multiplies the dynamic instruction count of a launch (rocprofv3 SQ_INSTS_VALU, profiles/sq_insts_valu.json)
by this average: the time the launch's VALU instructions alone occupy the SIMDs, against the measured kernel duration.
"""
import importlib
import json
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("3dgrut_amd.build")
CALIB = os.path.join(ROOT, "profiles", "r02a_valu_calib.json")
KERNELS = {  # bench stage -> (source file, mangled-name pattern of the kernel the bench frame runs)
    "render_fwd": ("gut_render.hip", r"gut_render_fwd_kernelILi2ELb1ELb0E"),
    "render_bwd": ("gut_render.hip", r"gut_render_bwd_kernelILi2ELb0ELb0E"),
    "project": ("gut_kernels.hip", r"gut_project_kernel"),
    "expand": ("gut_kernels.hip", r"gut_expand_kernel"),
    "grt_trace_fwd": ("grt_kernels.hip", r"grt_trace_fwd_kernelILi4ELb0ELb1E"),   # the packet-list build (frames with one ray origin)
}
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
THREE = ("v_fma_", "v_fmac_", "v_mad_", "v_min3_", "v_max3_", "v_med3_", "v_fmaak", "v_fmamk", "v_bfe_", "v_bfi_", "v_perm_", "v_alignbit", "v_lshl_add",
         "v_add3_", "v_lshl_or", "v_and_or", "v_or3_", "v_mad_u", "v_add_lshl", "v_xad_")


def classify(op):
    if not op.startswith("v_"):
        return None
    if op.endswith("_dpp"):
        return "dpp"
    if op.startswith("v_pk_"):
        return "packed"
    if op.startswith(TRANS):
        return "transcendental"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "compare"
    if op.startswith(THREE):
        return "three_source"
    return "plain"


def main():
    calib = json.load(open(CALIB))
    rows = {r["inst"]: r for r in calib["rows"] if r["waves_per_simd"] == 4}
    per_inst = calib["insts_per_wave"] * 4
    ns = lambda name: rows[name]["wall_ms"] * 1e6 / per_inst
    cost = {"plain": ns("v_mul_f32"), "three_source": ns("v_fma_f32"),
            "packed": (ns("v_pk_fma_f32") + ns("v_pk_mul_f32") + ns("v_pk_add_f32")) / 3.0, "transcendental": ns("v_exp_f32"),
            "dpp": ns("v_add_f32_dpp(row_shr:1)"), "compare": ns("v_cmp_lt_f32")}
    out = {"calibration": os.path.relpath(CALIB, ROOT), "ns_per_wave_instruction_per_simd": cost, "kernels": {}}
    asm = {}
    for stage, (src, pat) in KERNELS.items():
        if src not in asm:
            cmd = [b._hipcc(), "-x", "hip", *b.FLAGS, *b.unit_flags(src), "--cuda-device-only", "-S", os.path.join(b.CSRC, src), "-o", "-"]
            asm[src] = subprocess.run(cmd, capture_output=True, text=True).stdout
        s = asm[src]
        names = [n for n in re.findall(r"^(_Z\w+):", s, re.M) if re.search(pat, n)]
        assert names, (stage, pat)
        i = s.index("\n" + names[0] + ":")
        body = s[i:s.index(".Lfunc_end", i)]
        ops = [l.split()[0] for l in (x.strip() for x in body.split("\n")) if l and not l.startswith((".", ";", "//", "_Z")) and not l.endswith(":")]
        mix = Counter(c for c in (classify(o) for o in ops) if c)
        total = sum(mix.values())
        avg = sum(cost[c] * k for c, k in mix.items()) / total
        out["kernels"][stage] = {"kernel": names[0], "static_valu_instructions": total, "mix": {c: k / total for c, k in sorted(mix.items())},
                                 "avg_ns_per_instruction": avg}
        print(stage, total, {c: round(k / total, 3) for c, k in sorted(mix.items())}, f"avg {avg:.3f} ns")
    json.dump(out, open(os.path.join(ROOT, "profiles", "valu_model.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
