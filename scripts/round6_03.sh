#!/bin/bash
# Round 6: whole GPU suite after the device-plan switch (no -x: list everything that broke)
export TMPDIR=/tmp; out=gpurun_out/r6_03; mkdir -p $out
timeout 2700 python3 -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -15 $out/pytest_gpu.txt
