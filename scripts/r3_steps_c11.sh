step() { name=$1; shift; echo "=== $name"; ( time timeout 1500 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(tail -n 4 $O/$name.log | cut -c1-300)"; }
step hybrid python -m pytest tests/test_hybrid_gpu.py -x -q -s
grep "rays\|config 5" $O/hybrid.log | head -20
step bench_hybrid python bench.py --workload c5_hybrid_2m_1080p --steps 5 --warmup 2
tail -c 600 $O/bench_hybrid.log
