cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d /tmp/pq -o sq -- python $R/bench.py --workload c3_grt_1m_800 --steps 2 --warmup 1 --no-cpu-baseline > $O/prof2.log 2>&1
python $R/scripts/rocprof_summary.py counters /tmp/pq/sq_results.db "sq" > $O/grt_sq.txt
grep -A9 "^grt_trace_fwd" $O/grt_sq.txt
rocprofv3 --kernel-trace --stats -d /tmp/pg -o st -- python $R/bench.py --workload c3_grt_1m_800 --steps 4 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1
python $R/scripts/rocprof_summary.py stats /tmp/pg/st_results.db > $O/grt_kernel_stats.txt
head -14 $O/grt_kernel_stats.txt
