( time timeout 1200 python -m pytest tests/test_grt_gpu.py tests/test_hybrid_gpu.py tests/test_full_size_gpu.py -q -m gpu ) > $O/t.log 2>&1
grep -E "passed|failed|real" $O/t.log | tail -3
