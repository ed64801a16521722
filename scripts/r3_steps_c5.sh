step() { name=$1; shift; echo "=== $name"; ( time timeout 900 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(tail -n 3 $O/$name.log | tr '\n' ' ' | cut -c1-300)"; }
step grt_tests python -m pytest tests/test_grt_gpu.py -x -q
step grt_full python -m pytest tests/test_full_size_gpu.py -x -q -s -k "grt"
grep "re-derived" $O/grt_full.log
cd /tmp
step prof rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o st -- python $R/bench.py --workload c3_grt_1m_800 --no-cpu-baseline --steps 5 --warmup 2
cd $R
grep "grt bwd:" $O/prof.log | head -2
python scripts/rocprof_summary.py stats /tmp/prof_c5/st_results.db > $O/kernel_stats.txt 2>&1
head -8 $O/kernel_stats.txt
step bench_grt python bench.py --workload c3_grt_1m_800 --no-cpu-baseline
grep -o '"stages_ms": {[^}]*}' $O/bench_grt.log
