rm -f /tmp/bwd_multi_*.pt
for res in r1 r2; do
OMNIHD_POOL_BWD_PACKED=0 timeout 300 python3 scripts/lab/bwd_multi.py $res two_tables
OMNIHD_POOL_BWD_PACKED=1 timeout 300 python3 scripts/lab/bwd_multi.py $res packed
OMNIHD_POOL_BWD_PACKED=0 timeout 300 python3 scripts/lab/bwd_multi.py $res two_tables
OMNIHD_POOL_BWD_PACKED=1 timeout 300 python3 scripts/lab/bwd_multi.py $res packed
done
