cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=$R/gpurun_out/summ_other
mkdir -p $S
for w in c2_1m_800 c4_3m_1080p c1_100k_400 c3_grt_100k_400 c4_nht_1m_1080p c3_grt_nht_1m_800 c5_hybrid_2m_1080p; do
  python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric' > $S/bench_$w.json
  python -c "
import json,sys
d=json.load(open('$S/bench_$w.json')); print('$w', round(d['ms_per_step'],3), d.get('stages_ms'))" | cut -c1-400
done
python $R/bench.py --workload c4_1m_1080p --k-buffer 16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric' > $S/bench_c4_1m_1080p_k16.json
python -c "
import json
d=json.load(open('$S/bench_c4_1m_1080p_k16.json')); print('k16', round(d['ms_per_step'],3))"
for w in c4_nht_1m_1080p c3_grt_nht_1m_800 c2_1m_800; do
  rocprofv3 --kernel-trace --stats -d /tmp/p_$w -o st -- python $R/bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > /tmp/p_$w.log 2>&1
  python $R/scripts/rocprof_summary.py stats /tmp/p_$w/st_results.db | head -14 > $S/${w}_kernel_stats.txt
done
ls $S
