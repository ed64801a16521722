cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in 8 16 32 64; do
echo "== lane area $a"
GRUT_GRT_LANE_AREA=$a rocprofv3 --kernel-trace --stats -d /tmp/pg$a -o st -- python $R/bench.py --workload c3_grt_1m_800 --steps 4 --warmup 2 --no-cpu-baseline > $O/prof$a.log 2>&1
python $R/scripts/rocprof_summary.py stats /tmp/pg$a/st_results.db | grep "list_count\|list_expand"
grep '^{"metric' $O/prof$a.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages_ms'], d['work']['list_entries'])"
done
