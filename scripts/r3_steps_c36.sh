step() { name=$1; shift; echo "=== $name"; ( time timeout 2400 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(grep -E 'passed|failed|^E  |smoke ok' $O/$name.log | tail -n 5 | cut -c1-400)"; }
step all_gpu python -m pytest tests -q -m gpu
step smoke python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
bash scripts/profile_all.sh r03zz all > $O/profile.log 2>&1
tail -3 $O/profile.log | cut -c1-300
