# A/B of libgrut_amd.so builds that differ in the backend's scheduling strategy only (-mllvm -amdgpu-sched-strategy=max-ilp on some files):
# variants/libgrut_<tag>.so, selected through GRUT_AMD_LIB; one box, interleaved repetitions.  Usage: bash scripts/ab_sched_variants.sh "base ilpA ilpB" [c3]
mkdir -p gpurun_out
for rep in 1 2 3; do
for v in ${1:-base ilp}; do
  if [ $v = base ]; then unset GRUT_AMD_LIB; else export GRUT_AMD_LIB=$PWD/variants/libgrut_$v.so; fi
  python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v c4', round(d['ms_per_step'],4), {k:round(x,3) for k,x in d['stages_ms'].items() if x>0.05})"
  if [ -n "$2" ]; then python bench.py --workload c3_grt_1m_800 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v c3', round(d['ms_per_step'],4), {k:round(x,3) for k,x in d['stages_ms'].items()})"; fi
done; done
