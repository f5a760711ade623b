#!/bin/bash
# scripts/gpu.sh with retries while every GPU slot of the pod is busy (exit code 3: nothing charged).  Usage: scripts/gpu_retry.sh <timeout> <script>
set -e
make -C "$(dirname "$0")/../omnihd-scenes_amd/csrc" -j6 > /dev/null
make -C "$(dirname "$0")/../oracle" > /dev/null
for attempt in $(seq 1 30); do
  set +e
  /usr/local/graft/bin/gpurun --timeout "$1" -- "bash $2"; rc=$?
  set -e
  if [ $rc -ne 3 ]; then exit $rc; fi
  echo "[gpu_retry] no slot (attempt $attempt), waiting 60 s"; sleep 60
done
exit 3
