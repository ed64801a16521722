python bench.py --workload c3_grt_1m_800 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -E '^\{"metric' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['ms_per_step'], d['stages_ms'])"
