for res in r1 r2; do
OMNIHD_POOL_BWD_MULTI=0 timeout 300 python3 scripts/lab/bwd_multi.py $res single
for m in 1 224 192 160 128; do OMNIHD_POOL_BWD_MULTI=$m timeout 300 python3 scripts/lab/bwd_multi.py $res multi$m; done
done
