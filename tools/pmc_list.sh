#!/bin/bash
# the hardware counters rocprofv3 offers on this box: tools/pmc_list.sh <out file under gpurun_out/>
set -euo pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
rocprofv3 -L > "$GRAFT_REPO_ROOT/gpurun_out/$1" 2>&1 || true
