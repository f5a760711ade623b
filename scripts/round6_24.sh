#!/bin/bash
# Round 6: the whole GPU suite on the tree with the ops package and the TF32-grade form
export TMPDIR=/tmp; out=gpurun_out/r6_24; mkdir -p $out
timeout 1500 python3 -m pytest tests -q -m gpu -p no:cacheprovider -x > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -8 $out/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc $?"; tail -2 $out/smoke.txt
