step() { name=$1; shift; echo "=== $name"; ( time timeout 600 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(grep -E "passed|failed|fault" $O/$name.log | tail -n 3 | cut -c1-300)"; }
export GRUT_POISON_SCRATCH=1
step gut_fwd python -m pytest tests/test_gut_gpu.py -x -q -k "test_forward_matches_oracle"
step gut_bwd python -m pytest tests/test_gut_gpu.py -x -q -k "test_backward_matches_oracle"
step gut_k python -m pytest tests/test_gut_gpu.py -x -q -k "kbuffer"
step grt_fwd python -m pytest tests/test_grt_gpu.py -x -q -k "hit_order"
step grt_bwd python -m pytest tests/test_grt_gpu.py -x -q -k "render_and_gradients"
step hybrid python -m pytest tests/test_hybrid_gpu.py -x -q
step optim python -m pytest tests/test_optim_gpu.py tests/test_sort_gpu.py tests/test_dp_gpu.py -x -q
