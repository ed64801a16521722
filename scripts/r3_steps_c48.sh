step() { name=$1; shift; echo "=== $name"; ( time timeout 1500 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(grep -E 'passed|failed|^E  |smoke ok' $O/$name.log | tail -n 5 | cut -c1-400)"; }
step hybrid python -m pytest tests/test_hybrid_gpu.py tests/test_grt_gpu.py -q -m gpu -x
step bench_c5 python bench.py --workload c5_hybrid_2m_1080p --steps 5 --warmup 2 --no-cpu-baseline
grep '^{"metric' $O/bench_c5.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('stages_ms'), d['value'])"
GRUT_GRT_NO_LISTS=1 python bench.py --workload c5_hybrid_2m_1080p --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('no lists', d['ms_per_step'])"
