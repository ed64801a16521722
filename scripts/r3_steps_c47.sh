step() { name=$1; shift; echo "=== $name"; ( time timeout 1500 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(grep -E 'passed|failed|^E  |smoke ok' $O/$name.log | tail -n 5 | cut -c1-400)"; }
step grt python -m pytest tests/test_grt_gpu.py tests/test_hybrid_gpu.py -q -m gpu -x
step bench_grt python bench.py --workload c3_grt_1m_800 --steps 20 --warmup 5 --no-cpu-baseline
grep '^{"metric' $O/bench_grt.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages_ms'])"
GRUT_GRT_NO_SPECULATION=1 python bench.py --workload c3_grt_1m_800 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('no speculation', d['ms_per_step'], d['stages_ms'])"
