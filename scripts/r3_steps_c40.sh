step() { name=$1; shift; echo "=== $name"; ( time timeout 1500 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(grep -E 'passed|failed|^E  |smoke ok' $O/$name.log | tail -n 5 | cut -c1-400)"; }
step grt python -m pytest tests/test_grt_gpu.py tests/test_hybrid_gpu.py -q -m gpu -x
step bench_grt python bench.py --workload c3_grt_1m_800 --steps 10 --warmup 3
grep "leaf tests" $O/bench_grt.log | cut -c1-400
grep '^{"metric' $O/bench_grt.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages_ms'])"
