#!/bin/bash
# usage: tools/gpu_step.sh <log> <timeout_s> <cmd...>  -- run one GPU step under `timeout -k 10`, log to gpurun_out/<log>;
# exit 0 unless the step was killed at its limit (then no further GPU step may start: chain steps with &&).
log=$1; lim=$2; shift 2
mkdir -p gpurun_out
timeout -k 10 "$lim" "$@" > "gpurun_out/$log" 2>&1
rc=$?
echo "rc=$rc" >> "gpurun_out/$log"
echo "[$log] rc=$rc: $(tail -n 3 "gpurun_out/$log" | tr '\n' ' ' | cut -c1-300)"
[ $rc -ne 124 ] && [ $rc -ne 137 ]
