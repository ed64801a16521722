step() { name=$1; shift; echo "=== $name"; ( time timeout 1500 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(tail -n 4 $O/$name.log | cut -c1-300)"; }
step grt_tests python -m pytest tests/test_grt_gpu.py -x -q
step bench_grt python bench.py --workload c3_grt_1m_800 --no-cpu-baseline
grep -o '"stages_ms": {[^}]*}' $O/bench_grt.log; grep -o '"work": {[^}]*}' $O/bench_grt.log
cd /tmp
GRT="python $R/bench.py --workload c3_grt_1m_800 --steps 3 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p19_st -o st -- $GRT > $O/st.log 2>&1
python $R/scripts/rocprof_summary.py stats /tmp/p19_st/st_results.db > $O/grt_kernel_stats.txt
head -9 $O/grt_kernel_stats.txt
cd $R
