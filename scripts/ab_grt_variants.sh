# A/B of libgrut_amd.so variants over the 3DGRT workloads (one box, interleaved).  Usage: bash scripts/ab_grt_variants.sh "base gilp"
for rep in 1 2; do
for wl in c3_grt_icosa_1m_800 c3_grt_custom_1m_800 c3_grt_trihexa_1m_800 c3_grt_trisurfel_1m_800 c3_grt_nht_1m_800 c5_hybrid_2m_1080p c3_grt_sphere_1m_800; do
for v in ${1:-base gilp}; do
  if [ $v = base ]; then unset GRUT_AMD_LIB; else export GRUT_AMD_LIB=$PWD/variants/libgrut_$v.so; fi
  python bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v $wl', round(d['ms_per_step'],3), {k:round(x,3) for k,x in d.get('stages_ms',{}).items()})"
done; done; done
