step() { name=$1; shift; echo "=== $name"; ( time timeout 600 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(grep -E '^{|passed|failed|^E  ' $O/$name.log | cut -c1-700)"; }
step nht_tests python -m pytest tests/test_gut_gpu.py -x -q -k "nht"
step nht python bench.py --workload c4_nht_1m_1080p --steps 5 --warmup 2
