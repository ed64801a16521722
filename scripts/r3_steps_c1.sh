step() { name=$1; shift; echo "=== $name"; ( time timeout 900 "$@" ) > $O/$name.log 2>&1; echo "rc=$? $(tail -n 3 $O/$name.log | tr '\n' ' ' | cut -c1-600)"; }
step grt_tests python -m pytest tests/test_grt_gpu.py tests/test_abi.py -x -q
step bench_grt_new python bench.py --workload c3_grt_1m_800 --no-cpu-baseline
step parity_grt python scripts/diag_full_parity.py c3_grt_100k_400
step noise python scripts/diag_fp32_noise.py c2_1m_800 c4_1m_1080p
export GRUT_BENCH_BACKEND=gloo; step selflaunch python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline; unset GRUT_BENCH_BACKEND
