#!/bin/bash
# Run GPU steps one after another, each under its own time limit; an ordinary failure (exit 1: a failed assertion)
# lets the next step run, anything else (fault, abort, kill, time limit) ends the script there.
# usage: gpu_steps.sh "<seconds> <logname> <command...>" ...
mkdir -p gpurun_out
rc_all=0
for step in "$@"; do
    secs=${step%% *}; rest=${step#* }
    log=${rest%% *}; cmd=${rest#* }
    echo "== [$secs s] $cmd  -> gpurun_out/$log"
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$log" 2>&1
    rc=$?
    tail -n 6 "gpurun_out/$log"
    echo "== exit $rc"
    if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: exit $rc"; exit $rc; fi
    [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
