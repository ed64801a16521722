export GRUT_BENCH_BACKEND=gloo
for kind in factored visible sharded allreduce; do
  export GRUT_BENCH_EXCHANGE=$kind
  ( timeout 600 python bench.py --gpus 2 --workload c1_100k_400 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary ) > $O/dp_$kind.log 2>&1
  echo "$kind rc=$? $(grep '^{' $O/dp_$kind.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], round(d['ms_per_step'],3), d.get('exchange'))" 2>&1 | cut -c1-400)"
done
tail -5 $O/dp_sharded.log | cut -c1-300
