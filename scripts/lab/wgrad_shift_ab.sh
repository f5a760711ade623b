python -m pytest tests/test_conv_gpu.py tests/test_conv_split_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "== shift form"; python3 scripts/lab/wgrad_time.py 2>&1 | grep -v "^/opt" | tail -14
echo "== three copies (OMNIHD_WGRAD_SHIFT=0)"; OMNIHD_WGRAD_SHIFT=0 python3 scripts/lab/wgrad_time.py 2>&1 | grep -v "^/opt" | tail -14
