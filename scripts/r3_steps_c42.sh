for m in 0.5 0.8 1.0 1.25 2.0 1e30; do
echo "== mark $m"
GRUT_GRT_LIST_MARK=$m python bench.py --workload c3_grt_1m_800 --steps 10 --warmup 3 2>&1 | grep -E 'inserts|^\{"metric' | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['stages_ms'], d['work']['packet_tests'])
    else: print(l.strip()[:160])"
done
