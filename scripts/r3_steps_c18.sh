python scripts/diag_grt_balance.py > $O/balance.log 2>&1; tail -6 $O/balance.log
