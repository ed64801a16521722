step() { name=$1; shift; echo "=== $name"; ( time timeout 2400 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(tail -n 4 $O/$name.log | cut -c1-300)"; }
step all_gpu python -m pytest tests -q -m gpu -s
