step() { name=$1; shift; echo "=== $name"; ( time timeout 900 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(tail -n 3 $O/$name.log | cut -c1-300)"; }
cd /tmp
step prof rocprofv3 --kernel-trace --stats -d /tmp/prof_c9 -o st -- python $R/bench.py --workload c3_grt_1m_800 --no-cpu-baseline --steps 6 --warmup 2
cd $R
python scripts/rocprof_summary.py stats /tmp/prof_c9/st_results.db > $O/kernel_stats.txt 2>&1
head -9 $O/kernel_stats.txt
