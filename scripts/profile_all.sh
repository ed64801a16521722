#!/bin/bash
# profile_all.sh TAG — on the GPU box: rocprofv3 kernel stats + separate PMC passes (HBM fetch / write, SQ counters) of
# bench.py for both workloads; raw outputs under gpurun_out/prof_TAG_*, summarised later by scripts/rocprof_summary.py.
# (PMC passes carry --kernel-trace only: gpurun refuses --pmc together with sys / runtime traces.)
# profile_all.sh TAG gut — the 3DGUT workload only (when the 3DGRT kernels have not changed since the last profile).
# profile_all.sh TAG stats — only the kernel-trace timing pass of the 3DGUT workload plus a plain bench line.
TAG=${1:-rXX}
ONLY=${2:-all}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=/tmp/prof_$TAG          # raw rocprofv3 databases stay on the box (gpurun_out/ is capped at 64 MiB)
S=$R/gpurun_out/summ_$TAG  # text summaries travel back
mkdir -p $O $S
GUT="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary"   # (--no-secondary: no 3DGRT / training-surrogate kernels in the averages)
GRT="python $R/bench.py --workload c3_grt_1m_800 --steps 2 --warmup 1 --no-cpu-baseline"
# the timing pass runs bench.py's default step counts (clocks settle over the first steps; 4 steps read ~6 % slow)
rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_stats -o st -- python $R/bench.py --no-cpu-baseline --no-secondary > $O/prof_${TAG}_stats.log 2>&1
[ "$ONLY" = "stats" ] && { python $R/scripts/rocprof_summary.py stats $O/prof_${TAG}_stats/st_results.db > $S/kernel_stats.txt; python $R/bench.py > $S/bench.json 2> $S/bench.err; tail -c 300 $S/bench.json; exit 0; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_${TAG}_fetch -o f -- $GUT > $O/prof_${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_${TAG}_write -o w -- $GUT > $O/prof_${TAG}_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $O/prof_${TAG}_sq -o sq -- $GUT > $O/prof_${TAG}_sq.log 2>&1
if [ "$ONLY" != "gut" ]; then
rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_grt_stats -o st -- $GRT > $O/prof_${TAG}_grt_stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $O/prof_${TAG}_grt_sq -o sq -- $GRT > $O/prof_${TAG}_grt_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_${TAG}_grt_fetch -o f -- $GRT > $O/prof_${TAG}_grt_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_${TAG}_grt_write -o w -- $GRT > $O/prof_${TAG}_grt_write.log 2>&1
fi
SUM="python $R/scripts/rocprof_summary.py"
$SUM stats $O/prof_${TAG}_stats/st_results.db > $S/kernel_stats.txt
$SUM pmc $O/prof_${TAG}_fetch/f_results.db $O/prof_${TAG}_write/w_results.db > $S/pmc_hbm.txt
$SUM counters $O/prof_${TAG}_sq/sq_results.db "rocprofv3 --kernel-trace --pmc SQ_* (one pass), bench.py c4_1m_1080p" > $S/sq_counters.txt
if [ "$ONLY" != "gut" ]; then
$SUM stats $O/prof_${TAG}_grt_stats/st_results.db > $S/grt_kernel_stats.txt
$SUM pmc $O/prof_${TAG}_grt_fetch/f_results.db $O/prof_${TAG}_grt_write/w_results.db > $S/grt_pmc_hbm.txt
$SUM counters $O/prof_${TAG}_grt_sq/sq_results.db "rocprofv3 --kernel-trace --pmc SQ_* (one pass), bench.py c3_grt_1m_800" > $S/grt_sq_counters.txt
fi
cp $R/profiles/pmc_traffic.json $S/pmc_traffic.json 2>/dev/null
cp $R/profiles/sq_insts_valu.json $S/sq_insts_valu.json 2>/dev/null
GRUT_TRAFFIC_JSON=$S/pmc_traffic.json $SUM traffic $O/prof_${TAG}_fetch/f_results.db $O/prof_${TAG}_write/w_results.db c4_1m_1080p > /dev/null
GRUT_VALU_JSON=$S/sq_insts_valu.json $SUM valu $O/prof_${TAG}_sq/sq_results.db > /dev/null
$SUM timeline $O/prof_${TAG}_stats/st_results.db > $S/timeline.txt
if [ "$ONLY" != "gut" ]; then
GRUT_TRAFFIC_JSON=$S/pmc_traffic.json $SUM traffic $O/prof_${TAG}_grt_fetch/f_results.db $O/prof_${TAG}_grt_write/w_results.db c3_grt_1m_800 > /dev/null
GRUT_VALU_JSON=$S/sq_insts_valu.json $SUM valu $O/prof_${TAG}_grt_sq/sq_results.db > /dev/null
cp $S/pmc_traffic.json $R/profiles/pmc_traffic.json; cp $S/sq_insts_valu.json $R/profiles/sq_insts_valu.json   # the bench lines below read them
fi
python $R/bench.py > $S/bench.json 2> $S/bench.err
[ "$ONLY" != "gut" ] && python $R/bench.py --workload c3_grt_1m_800 --no-cpu-baseline > $S/bench_grt.json 2> $S/bench_grt.err
ls -la $S
tail -c 400 $S/bench.json
