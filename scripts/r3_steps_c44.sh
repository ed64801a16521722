bash scripts/profile_all.sh r03zz all > $O/profile.log 2>&1
tail -3 $O/profile.log | cut -c1-300
