step() { name=$1; shift; echo "=== $name"; ( time timeout 1500 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(grep -E 'passed|failed|^E  |smoke ok' $O/$name.log | tail -n 5 | cut -c1-400)"; }
step grt_lists python -m pytest tests/test_grt_gpu.py -q -m gpu -x -k "lists or fisheye or tree_walk or particles"
step bench_grt python bench.py --workload c3_grt_1m_800 --steps 10 --warmup 3
tail -1 $O/bench_grt.log | cut -c1-1500
GRUT_GRT_NO_GRID=1 python bench.py --workload c3_grt_1m_800 --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-400
