step() { name=$1; shift; echo "=== $name"; ( time timeout 1500 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(grep -E "passed|failed" $O/$name.log | tail -n 2 | cut -c1-300)"; }
step grt_tests python -m pytest tests/test_grt_gpu.py tests/test_hybrid_gpu.py -x -q
step bench_grt python bench.py --workload c3_grt_1m_800 --no-cpu-baseline
grep -o '"stages_ms": {[^}]*}' $O/bench_grt.log; grep -o '"work": {[^}]*}' $O/bench_grt.log
step grt_full python -m pytest tests/test_full_size_gpu.py -x -q -s -k "grt"
