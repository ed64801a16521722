#!/bin/bash
# profile_mfma.sh — on the GPU box: the forward sweep with and without the fp32-MFMA pair geometry (variants/libgrut_amd_mfma2_w6.so,
# scripts/build_variant.sh mfma2_w6 gut_render.hip -DGRUT_FWD_MFMA=2 -DGRUT_FWD_WAVES=6): kernel time (bench stage events) and one PMC pass
# each (VALU instructions, fp32 MFMA instructions, VALU-active and MFMA-busy cycles).  Summaries -> gpurun_out/summ_r05_mfma/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=/tmp/prof_mfma; S=$R/gpurun_out/summ_r05_mfma
mkdir -p $O $S
B="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary"
PMC="SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
rocprofv3 --kernel-trace --pmc $PMC -d $O/base -o b -- $B > $O/base.log 2>&1
GRUT_AMD_LIB=$R/variants/libgrut_amd_mfma2_w6.so rocprofv3 --kernel-trace --pmc $PMC -d $O/mfma -o m -- $B > $O/mfma.log 2>&1
python $R/scripts/rocprof_summary.py counters $O/base/b_results.db "in-tree build (packed-VALU pair geometry)" > $S/counters_base.txt 2>&1
python $R/scripts/rocprof_summary.py counters $O/mfma/m_results.db "GRUT_FWD_MFMA=2 (v = M d on v_mfma_f32_4x4x1_16b_f32)" > $S/counters_mfma.txt 2>&1
grep -h "render_fwd" $S/counters_base.txt $S/counters_mfma.txt | cut -c1-250
