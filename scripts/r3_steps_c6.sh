step() { name=$1; shift; echo "=== $name"; ( time timeout 900 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(tail -n 8 $O/$name.log | cut -c1-400)"; }
step bwd1m python scripts/diag_grt_bwd_1m.py 13
