#!/usr/bin/env python
"""Turn rocprofv3's rocpd sqlite outputs (gpurun_out/prof_*/…_results.db) into the text summaries kept under profiles/.

    python scripts/rocprof_summary.py stats  <stats.db>  > profiles/rNN_kernel_stats.txt
    python scripts/rocprof_summary.py pmc    <fetch.db> <write.db>  > profiles/rNN_pmc_hbm.txt
    python scripts/rocprof_summary.py traffic <fetch.db> <write.db> <workload>   (writes profiles/pmc_traffic.json)
    python scripts/rocprof_summary.py counters <pmc.db> [title]   > profiles/rNN_sq_counters.txt
    python scripts/rocprof_summary.py valu <sq.db> [<sq.db> ...]    (writes profiles/sq_insts_valu.json)
    python scripts/rocprof_summary.py timeline <stats.db>           one step's kernels in launch order with the gaps between them

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE tallies 128-B read requests at 64 B
(MI355X_MICROARCH.md, "HBM"): the corrected column doubles it.  WRITE_SIZE is taken as reported (uncalibrated there).
"""
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name: str) -> str:
    m = re.search(r"((?:gut|grt|radix|scan)_[a-z0-9_]+)(<[^>]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    return name[:70]


def stats(db):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# rocprofv3 --kernel-trace --stats summary ({os.path.basename(os.path.dirname(db))}); durations in us")
    print(f"{'kernel':60s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for n, calls, tot, avg, pct in rows:
        print(f"{short(n):60s} {calls:6d} {tot:12.1f} {avg:10.2f} {pct:6.2f}")


def counter_avgs(db, counter):
    c = sqlite3.connect(db)
    out = {}
    q = "select kernel_name, value, duration from counters_collection where counter_name = ?"
    for name, value, dur in c.execute(q, (counter,)):
        k = short(name)
        e = out.setdefault(k, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += value
        e[2] += dur
    return {k: (n, v / n, d / n) for k, (n, v, d) in out.items()}


def counters(db, title=""):
    """every counter of a --pmc run, averaged per dispatch and per kernel"""
    c = sqlite3.connect(db)
    acc = {}
    for name, counter, value, dur in c.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        k = acc.setdefault(short(name), {})
        v = k.setdefault(counter, [0.0, 0])
        v[0] += value; v[1] += 1
        d = k.setdefault("dur_us", [0.0, 0])
        d[0] += dur / 1e3; d[1] += 1
    if title:
        print(f"# {title}")
    print("# averages per dispatch")
    for kname in sorted(acc):
        print(kname)
        for cn in sorted(acc[kname]):
            tot, n = acc[kname][cn]
            print(f"   {cn:24s} {tot / max(n, 1):16.1f}")


def pmc(fetch_db, write_db):
    f = counter_avgs(fetch_db, "FETCH_SIZE")
    w = counter_avgs(write_db, "WRITE_SIZE")
    print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), averages per dispatch")
    print("# read_MB_corrected = 2 x FETCH_SIZE (gfx950 counts 128-B requests at 64 B); write as reported")
    print(f"{'kernel':60s} {'calls':>6s} {'fetch_MB':>10s} {'read_MB_corr':>13s} {'write_MB':>10s} {'hbm_MB':>10s} {'avg_us':>9s}")
    for k in sorted(f, key=lambda k: -f[k][1]):
        n, fv, dur = f[k]
        wv = w.get(k, (0, 0.0, 0.0))[1]
        if not (k.startswith(("gut_", "grt_", "radix_", "scan_"))):
            continue
        print(f"{k:60s} {n:6d} {fv / 1024:10.2f} {2 * fv / 1024:13.2f} {wv / 1024:10.2f} {(2 * fv + wv) / 1024:10.2f} {dur / 1e3:9.1f}")


STAGE_OF = {"gut_project_kernel": "project", "gut_expand_kernel": "expand", "gut_tile_ranges_kernel": "tile_ranges",
            "gut_render_fwd_kernel": "render_fwd", "gut_render_bwd_kernel": "render_bwd", "gut_project_bwd_kernel": "project_bwd",
            "grt_trace_fwd_kernel": "trace_fwd", "grt_replay_bwd_kernel": "replay_bwd"}
VALU_KEY = {"gut_project_kernel": "project", "gut_expand_kernel": "expand", "gut_render_fwd_kernel": "render_fwd", "gut_render_bwd_kernel": "render_bwd",
            "gut_project_bwd_kernel": "project_bwd", "grt_trace_fwd_kernel": "grt_trace_fwd"}


def traffic(fetch_db, write_db, workload):
    f = counter_avgs(fetch_db, "FETCH_SIZE")
    w = counter_avgs(write_db, "WRITE_SIZE")
    res, calls = {}, {}
    for k, (n, fv, _) in f.items():
        base = re.sub(r"<.*", "", k)
        if base in STAGE_OF and n > calls.get(STAGE_OF[base], 0):   # of several instantiations, the one the timed steps run
            calls[STAGE_OF[base]] = n
            res[STAGE_OF[base]] = int((2 * fv + w.get(k, (0, 0.0, 0.0))[1]) * 1024)
    path = os.environ.get("GRUT_TRAFFIC_JSON", os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    allw = json.load(open(path)) if os.path.exists(path) else {}
    allw[workload] = res
    json.dump(allw, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(res))


def valu(*dbs):
    """profiles/sq_insts_valu.json: SQ_INSTS_VALU per dispatch of the kernels bench.py prices against the VALU model (the instantiation
    with the most dispatches of each)"""
    path = os.environ.get("GRUT_VALU_JSON", os.path.join(ROOT, "profiles", "sq_insts_valu.json"))
    out = json.load(open(path)) if os.path.exists(path) else {}
    for db in dbs:
        best = {}
        for k, (n, v, _) in counter_avgs(db, "SQ_INSTS_VALU").items():
            base = re.sub(r"<.*", "", k)
            if base in VALU_KEY and n > best.get(VALU_KEY[base], (0, 0))[0]:
                best[VALU_KEY[base]] = (n, int(v))
        out.update({k: v for k, (_, v) in best.items()})
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out))


def timeline(db, first="gut_project_kernel"):
    """the kernels of the LAST step of a --kernel-trace run in launch order: start offset, duration and the idle gap in front of each"""
    c = sqlite3.connect(db)
    try:
        rows = list(c.execute("select name, start, end from kernels order by start"))
    except sqlite3.Error as e:
        print("no `kernels` view:", e)
        for (n,) in c.execute("select name from sqlite_master where type in ('table','view')"):
            print("  ", n)
        return
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    if len(starts) < 2:
        print("fewer than two steps in the trace"); return
    a, b = starts[-2], starts[-1]
    t0, prev_end = rows[a][1], rows[a][1]
    print(f"# one step ({b - a} kernels) of {os.path.basename(os.path.dirname(db))}: start offset, duration, gap before (us)")
    busy = gaps = 0.0
    for n, st, en in rows[a:b]:
        print(f"{short(n):58s} {(st - t0) / 1e3:9.1f} {(en - st) / 1e3:9.1f} {(st - prev_end) / 1e3:8.1f}")
        busy += (en - st) / 1e3; gaps += max(0.0, (st - prev_end) / 1e3)
        prev_end = max(prev_end, en)
    print(f"# step period {(rows[b][1] - t0) / 1e3:.1f} us: kernels {busy:.1f} us, gaps inside the step {gaps:.1f} us, gap to the next step {(rows[b][1] - prev_end) / 1e3:.1f} us")


if __name__ == "__main__":
    cmd = sys.argv[1]
    {"stats": stats, "pmc": pmc, "traffic": traffic, "counters": counters, "timeline": timeline, "valu": valu}[cmd](*sys.argv[2:])
