bash scripts/profile_all.sh r03a gut > $O/profile.log 2>&1
cp -r gpurun_out/summ_r03a $O/ 2>/dev/null
head -12 gpurun_out/summ_r03a/pmc_hbm.txt
