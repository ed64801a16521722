step() { name=$1; shift; echo "=== $name"; ( time timeout 2400 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(grep -E 'passed|failed|^E  |smoke ok' $O/$name.log | tail -n 5 | cut -c1-400)"; }
step all_gpu python -m pytest tests -q -m gpu
step smoke python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
step bench python bench.py
grep '^{"metric' $O/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value']); s=d['secondary']
for k,v in s.items(): print(k, v.get('ms_per_step'), v.get('stages_ms'))" | cut -c1-400
