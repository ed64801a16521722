bash scripts/profile_all.sh r03z all > $O/profile.log 2>&1
tail -5 $O/profile.log | cut -c1-400
