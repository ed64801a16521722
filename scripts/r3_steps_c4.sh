step() { name=$1; shift; echo "=== $name"; ( time timeout 900 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(tail -n 3 $O/$name.log | tr '\n' ' ' | cut -c1-300)"; }
step grt_tests python -m pytest tests/test_grt_gpu.py -x -q
step grt_full python -m pytest tests/test_full_size_gpu.py -x -q -k "grt"
step bench_grt python bench.py --workload c3_grt_1m_800 --no-cpu-baseline
grep -o '"stages_ms": {[^}]*}' $O/bench_grt.log; grep "grt bwd:" $O/bench_grt.log | head -2
step parity_grt python scripts/diag_full_parity.py c3_grt_100k_400
grep "G_replay_unmasked\|round_shift" $O/parity_grt.log
