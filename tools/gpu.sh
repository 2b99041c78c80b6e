#!/bin/bash
# Build the library (so that the snapshot never carries a stale .so) and run a command on the GPU box:
#   tools/gpu.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
python -m covest_amd.build > /dev/null
python -c "from oracle import covest_oracle as o; o.build()"
exec /usr/local/graft/bin/gpurun "$@"
