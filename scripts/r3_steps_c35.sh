step() { name=$1; shift; echo "=== $name"; ( time timeout 900 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(grep -E '^{' $O/$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['stages_ms'])" 2>&1 | cut -c1-300)"; }
step grt_nht python bench.py --workload c3_grt_nht_1m_800 --steps 4 --warmup 2 --no-cpu-baseline
step grt python bench.py --workload c3_grt_1m_800 --steps 4 --warmup 2 --no-cpu-baseline
