cd /tmp
GRT="python $R/bench.py --workload c3_grt_1m_800 --steps 3 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p16_st -o st -- $GRT > $O/st.log 2>&1
python $R/scripts/rocprof_summary.py stats /tmp/p16_st/st_results.db > $O/grt_kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d /tmp/p16_sq -o sq -- $GRT > $O/sq.log 2>&1
python $R/scripts/rocprof_summary.py counters /tmp/p16_sq/sq_results.db "rocprofv3 --kernel-trace --pmc SQ_* (one pass), bench.py c3_grt_1m_800" > $O/grt_sq_counters.txt
head -30 $O/grt_kernel_stats.txt
cd $R
