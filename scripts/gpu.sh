#!/bin/bash
# Build the library HERE (the built .so travels with the snapshot), then run a script on the GPU box:  scripts/gpu.sh <timeout> <script>
set -e
make -C "$(dirname "$0")/../omnihd-scenes_amd/csrc" -j6 > /dev/null
make -C "$(dirname "$0")/../oracle" > /dev/null
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "bash $2"
