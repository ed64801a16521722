step() { name=$1; shift; echo "=== $name"; ( time timeout 900 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(tail -n 8 $O/$name.log | cut -c1-600)"; }
step grt_full python -m pytest tests/test_full_size_gpu.py -x -q -s -k "rederived"
grep "re-derived\|differs" $O/grt_full.log | head
