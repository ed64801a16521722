step() { name=$1; shift; echo "=== $name"; ( time timeout 1500 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(grep -E "passed|failed|^E  " $O/$name.log | tail -n 6 | cut -c1-500)"; }
step nht python -m pytest tests/test_grt_gpu.py -x -q -k "nht"
step grt python -m pytest tests/test_grt_gpu.py tests/test_hybrid_gpu.py -x -q
