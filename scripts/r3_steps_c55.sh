( time timeout 600 python -m pytest tests/test_grt_gpu.py tests/test_hybrid_gpu.py -q -m gpu ) > $O/t.log 2>&1
grep -E "passed|failed|real" $O/t.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
