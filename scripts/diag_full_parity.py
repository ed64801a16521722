"""On the GPU box: the staged full-size parity statistics of tests/parity_util.py for BASELINE's configurations, printed and written
as JSON (gpurun_out/full_parity.json -> profiles/rNN_full_parity.json).  The tests assert on the same numbers."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as pu  # noqa: E402

CASES = {
    "c1_100k_400": ("gut", 100_000, 400, 400, 0.01),
    "c2_1m_800": ("gut", 1_000_000, 800, 800, 0.01),
    "c4_1m_1080p": ("gut", 1_000_000, 1920, 1080, 0.01),
    "c4_3m_1080p": ("gut", 3_000_000, 1920, 1080, 0.007),
    "c3_grt_100k_400": ("grt", 100_000, 400, 400, 0.01, 1),
    "c3_grt_1m_800": ("grt", 1_000_000, 800, 800, 0.01, 149),
}

if __name__ == "__main__":
    which = sys.argv[1:] or list(CASES)
    res = {}
    for name in which:
        c = CASES[name]
        print(f"== {name}", flush=True)
        if c[0] == "gut":
            res[name] = pu.gut_full_parity(*c[1:], log=print)
        else:
            res[name] = pu.grt_full_parity(*c[1:5], ray_stride=c[5], log=print)
        sys.stdout.flush()
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(res, open(os.path.join(ROOT, "gpurun_out", "full_parity.json"), "w"), indent=1)
