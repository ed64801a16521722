step() { name=$1; shift; echo "=== $name"; ( time timeout 1500 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(grep -E "passed|failed|^E  " $O/$name.log | tail -n 4 | cut -c1-400)"; }
step gut python -m pytest tests/test_gut_gpu.py tests/test_abi.py -x -q
step bench python bench.py --no-cpu-baseline --no-secondary
grep -o '"ms_per_step": [0-9.]*' $O/bench.log
