step() { name=$1; shift; echo "=== $name"; ( time timeout 1500 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(grep -E 'passed|failed|^E  |smoke ok' $O/$name.log | tail -n 5 | cut -c1-400)"; }
step grt_lists python -m pytest tests/test_grt_gpu.py -q -m gpu -x -k "lists or fisheye or tree_walk or particles"
GRUT_GRT_NO_GRID=1 python -m pytest tests/test_grt_gpu.py tests/test_hybrid_gpu.py -q -m gpu -x 2>&1 | tail -2
