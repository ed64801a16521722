step() { name=$1; shift; echo "=== $name"; ( time timeout 900 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(tail -n 3 $O/$name.log | tr '\n' ' ' | cut -c1-300)"; }
cd /tmp
step prof rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o st -- python $R/bench.py --workload c3_grt_1m_800 --no-cpu-baseline --steps 3 --warmup 2
cd $R
grep "grt bwd:" $O/prof.log | head -3
python scripts/rocprof_summary.py stats /tmp/prof_c3/st_results.db > $O/kernel_stats.txt 2>&1 || find /tmp/prof_c3 | head
head -12 $O/kernel_stats.txt
