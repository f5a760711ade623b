#!/bin/bash
# Round-5 second GPU call: general convolution kernel parity, scatter C=320 in isolation, determinism diagnostic, DDP step profile.
export TMPDIR=/tmp; out=gpurun_out/r5b; mkdir -p $out
timeout 900 python3 -m pytest tests/test_conv_gen_gpu.py -m gpu -x -q 2>&1 | tail -30 > $out/conv_gen.txt; tail -12 $out/conv_gen.txt
timeout 300 python3 -m pytest "tests/test_radar_gpu.py::test_channels_last_scatter_with_channel_counts_whose_cells_straddle_wavefronts" -m gpu -q 2>&1 | tail -5 > $out/scatter.txt; cat $out/scatter.txt
timeout 300 python3 scripts/lab/scatter_c320.py > $out/scatter320.txt 2>&1; tail -8 $out/scatter320.txt
OMNIHD_DETERMINISTIC=1 timeout 900 python3 scripts/lab/determinism_pass.py > $out/determinism_pass.txt 2>&1; grep -v "^/opt\|Warn\|warn" $out/determinism_pass.txt | tail -45
timeout 1500 python3 -m pytest tests/test_determinism_gpu.py -m gpu -x -q 2>&1 | tail -8 > $out/determinism_test.txt; cat $out/determinism_test.txt
for mode in plain ddp; do
  rocprofv3 --output-format csv --kernel-trace --stats -d $out/prof_$mode -o step -- python3 scripts/lab/ddp1_step.py $mode > $out/ddp1_$mode.txt 2>&1
  cp $(find $out/prof_$mode -name "*kernel_stats.csv" | head -1) $out/ddp1_${mode}_kernel_stats.csv
  find $out/prof_$mode -type f -size +1M -delete
  tail -2 $out/ddp1_$mode.txt
done
