#!/bin/bash
# One GPU-box session: `gpurun --timeout T -- 'bash scripts/gpu_session.sh NAME "step a" "step b" ...'`.
# Every step is `label::command`; its output goes to gpurun_out/NAME/label.log, its exit code and last lines to stdout
# (gpurun returns only the tail of stdout).  Replaces round 3's 56 one-off step files.
set -u
NAME=$1; shift
export O=gpurun_out/$NAME TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p "$O"
for spec in "$@"; do
    label=${spec%%::*}; cmd=${spec#*::}
    echo "=== $label: $cmd"
    ( time timeout "${STEP_TIMEOUT:-1500}" bash -c "$cmd" ) > "$O/$label.log" 2>&1
    echo "rc=$? $(tail -n "${TAIL:-6}" "$O/$label.log" | cut -c1-400)"
done
