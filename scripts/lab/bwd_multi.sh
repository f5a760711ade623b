rm -f /tmp/bwd_multi_*.pt
for res in r1; do
OMNIHD_POOL_BWD_MULTI=0 timeout 300 python3 scripts/lab/bwd_multi.py $res single
for m in 256 224 192; do for fx in 100 300; do OMNIHD_POOL_BWD_MULTI=$m timeout 300 python3 scripts/lab/bwd_multi.py $res multi$m $fx; done; done
done
