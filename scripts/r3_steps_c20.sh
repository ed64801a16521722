step() { name=$1; shift; echo "=== $name"; ( time timeout 1500 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(tail -n 4 $O/$name.log | cut -c1-300)"; }
step gut_tests python -m pytest tests/test_gut_gpu.py tests/test_abi.py tests/test_dp_gpu.py -x -q
step bench python bench.py --no-cpu-baseline --no-secondary
grep -o '"stages_ms": {[^}]*}' $O/bench.log; grep -o '"ms_per_step": [0-9.]*' $O/bench.log
export GRUT_GUT_LEGACY_LISTS=1; step bench_legacy python bench.py --no-cpu-baseline --no-secondary; unset GRUT_GUT_LEGACY_LISTS
grep -o '"stages_ms": {[^}]*}' $O/bench_legacy.log; grep -o '"ms_per_step": [0-9.]*' $O/bench_legacy.log
step gut_full python -m pytest tests/test_full_size_gpu.py -x -q -s -k "not grt"
