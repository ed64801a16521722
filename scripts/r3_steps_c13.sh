step() { name=$1; shift; echo "=== $name"; ( time timeout 2400 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(tail -n 4 $O/$name.log | cut -c1-300)"; }
step fp16 python -m pytest tests/test_grt_gpu.py tests/test_gut_gpu.py -x -q -k "fp16"
step all_gpu python -m pytest tests -x -q -m gpu
step smoke python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
step bench python bench.py
