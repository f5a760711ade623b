#!/bin/bash
# Round-5 GPU call 20: per-workgroup trace of the three-taps weight gradient (who runs when, on which CU).
export TMPDIR=/tmp; out=gpurun_out/r5t; mkdir -p $out
python3 -c "import torch; p=torch.cuda.get_device_properties(0); print(p.name, p.multi_processor_count, 'CUs')" 2>/dev/null | tail -1
/opt/rocm/bin/rocminfo | grep -c "Compute Unit:" ; /opt/rocm/bin/rocminfo | grep -i "compute unit\|Shader Engines\|Shader Arrs\|Wavefront Size\|Max Waves" | head -8
rm -f /tmp/trace.txt
OMNIHD_WGRAD_NHWC_TRACE=/tmp/trace.txt WGRAD_BENCH_ONE=1,160,240,1024,1024,3,1,1 WGRAD_BENCH_LIBRARY=0 WGRAD_BENCH_CHAIN=0 timeout 300 python3 scripts/lab/wgrad_nhwc_bench.py 2>&1 | grep nhwc | cut -c1-80
python3 scripts/lab/wgrad_trace_report.py /tmp/trace.txt | tee $out/trace_1024.txt
rm -f /tmp/trace.txt
OMNIHD_WGRAD_NHWC_TRACE=/tmp/trace.txt WGRAD_BENCH_ONE=1,160,240,512,256,3,1,1 WGRAD_BENCH_LIBRARY=0 WGRAD_BENCH_CHAIN=0 timeout 300 python3 scripts/lab/wgrad_nhwc_bench.py 2>&1 | grep nhwc | cut -c1-80
python3 scripts/lab/wgrad_trace_report.py /tmp/trace.txt | tee $out/trace_512.txt
