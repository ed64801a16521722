#!/bin/bash
# r3_call.sh TAG — one GPU-box session of round 3 (development aid): runs the steps listed in scripts/r3_steps_TAG.sh with logs under gpurun_out/TAG/
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
source $R/scripts/r3_steps_$TAG.sh
ls -la $O
