step() { name=$1; shift; echo "=== $name"; ( time timeout 1500 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(grep -E 'passed|failed|^E  |smoke ok' $O/$name.log | tail -n 5 | cut -c1-400)"; }
step grt_lists python -m pytest tests/test_grt_gpu.py -q -m gpu -x -k "lists or fisheye or tree_walk or particles"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/pg -o st -- python $R/bench.py --workload c3_grt_1m_800 --steps 4 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1
python $R/scripts/rocprof_summary.py stats /tmp/pg/st_results.db > $O/grt_kernel_stats.txt
head -12 $O/grt_kernel_stats.txt
grep '^{"metric' $O/prof.log | cut -c100-330
