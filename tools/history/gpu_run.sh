#!/bin/bash
# usage: tools/gpu_run.sh <name> <timeout_s> '<command>'   -> gpurun_out/<name>.log (tail of command output)
name=$1; to=$2; shift 2
/usr/local/graft/bin/gpurun --timeout $to -- "mkdir -p gpurun_out; ( $* ) > gpurun_out/$name.log 2>&1; echo rc=\$? >> gpurun_out/$name.log" > gpurun_out/$name.gpurun 2>&1
tail -3 gpurun_out/$name.gpurun
