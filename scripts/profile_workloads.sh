#!/bin/bash
# profile_workloads.sh TAG — on the GPU box: one bench line (and, for the workloads with their own kernels, a rocprofv3 kernel-stats
# summary) for every bench.py workload other than the two profile_all.sh covers.  Output: gpurun_out/workloads_TAG/ (copy to
# profiles/TAG_workloads/).
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
S=$R/gpurun_out/workloads_$TAG
mkdir -p $S
B="python $R/bench.py --no-cpu-baseline --no-secondary"
for w in c1_100k_400 c2_1m_800 c4_3m_1080p c3_grt_100k_400 c5_hybrid_2m_1080p c4_nht_1m_1080p c3_grt_nht_1m_800 c3_grt_icosa_1m_800 c3_grt_custom_1m_800 c3_grt_trisurfel_1m_800 c3_grt_trihexa_1m_800 c3_grt_sphere_1m_800; do
    timeout 300 $B --workload $w > $S/bench_$w.json 2> $S/bench_$w.err
    [ -s $S/bench_$w.err ] || rm -f $S/bench_$w.err
done
timeout 300 $B --k-buffer 16 > $S/bench_c4_1m_1080p_k16.json 2> /dev/null
timeout 300 $B --workload c4_nht_1m_1080p --k-buffer 16 > $S/bench_c4_nht_1m_1080p_k16.json 2> /dev/null   # (round 6: features behind the sorted hit buffer)
for w in c2_1m_800 c4_nht_1m_1080p c3_grt_nht_1m_800 c3_grt_icosa_1m_800 c5_hybrid_2m_1080p; do
    timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/wl_${TAG}_$w -o st -- $B --workload $w > /tmp/wl_${TAG}_$w.log 2>&1
    python $R/scripts/rocprof_summary.py stats /tmp/wl_${TAG}_$w/st_results.db > $S/${w}_kernel_stats.txt
done
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/wl_${TAG}_k16 -o st -- $B --k-buffer 16 > /tmp/wl_${TAG}_k16.log 2>&1
python $R/scripts/rocprof_summary.py stats /tmp/wl_${TAG}_k16/st_results.db > $S/c4_1m_1080p_k16_kernel_stats.txt
ls -la $S
for f in $S/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split("bench_")[-1], round(d["ms_per_step"], 3), "ms", f'{d["value"]:.4g}', d["unit"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
