cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/pg -o st -- python $R/bench.py --workload c5_hybrid_2m_1080p --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
python $R/scripts/rocprof_summary.py stats /tmp/pg/st_results.db > $O/c5_kernel_stats.txt
head -16 $O/c5_kernel_stats.txt
