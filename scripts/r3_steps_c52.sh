python bench.py --workload c3_grt_1m_800 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -E 'inserts|^\{"metric' | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['stages_ms'], d['work'])
    else: print(l.strip()[:200])"
timeout 900 python -m pytest tests/test_grt_gpu.py -q -m gpu 2>&1 | tail -5 | cut -c1-300
