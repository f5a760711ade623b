#!/bin/bash
# The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer: builds /tmp/liboracle_san.so, preloads libasan and
# runs the CPU tests that drive the oracle (directly and through oracle/torch_shim.py) against it.
#   bash scripts/oracle_sanitize.sh            (round 1: 16 + 49 tests pass, no report)
set -e
cd "$(dirname "$0")/.."
make -s -C oracle san
cat > /tmp/oracle_san_run.py <<'PY'
import sys
sys.path[:0] = [".", "omnihd-scenes_amd"]
import oracle.cpu as OC
OC._SO = "/tmp/liboracle_san.so"
import pytest
sys.exit(pytest.main(sys.argv[1:] + ["-x", "-q", "-p", "no:cacheprovider"]))
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python /tmp/oracle_san_run.py \
  tests/test_oracle.py tests/test_model_cpu.py tests/test_pillars_cpu.py tests/test_lss_plain_cpu.py tests/test_triple_cpu.py
