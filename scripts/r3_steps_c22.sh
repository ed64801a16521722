step() { name=$1; shift; echo "=== $name"; ( time timeout 1500 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(tail -n 4 $O/$name.log | cut -c1-300)"; }
step alloc_tests python -m pytest tests/test_gut_gpu.py tests/test_grt_gpu.py -x -q -k "allocator or trim"
step all_gpu python -m pytest tests -x -q -m gpu
step bench python bench.py --no-cpu-baseline
grep -o '"ms_per_step": [0-9.]*' $O/bench.log
