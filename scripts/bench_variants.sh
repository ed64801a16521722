#!/bin/bash
# bench_variants.sh TAG NAME... — on the GPU box: bench.py (no CPU baseline) once per variant library variants/libgrut_amd_NAME.so
# ("base" = the in-tree library); one JSON line per variant under gpurun_out/TAG_variants.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/${TAG}_variants.txt
: > $OUT
for v in "$@"; do
  if [ "$v" = "base" ]; then unset GRUT_AMD_LIB; else export GRUT_AMD_LIB=$R/variants/libgrut_amd_$v.so; fi
  line=$(python $R/bench.py --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | tail -1)
  echo "$v $line" | python -c "
import sys, json
name, rest = sys.stdin.read().split(' ', 1)
d = json.loads(rest)
print(name, 'ms/step %.4f' % d['ms_per_step'], 'rays/s %.4g' % d['value'], ' '.join('%s=%.3f' % (k, v) for k, v in d['stages_ms'].items()))
" | tee -a $OUT
done
