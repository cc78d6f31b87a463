#!/bin/bash
# Timeline of lbh_index_add (pinned, pre-packed 1 M-claim batches): kernel + memory-copy trace.  Run on the GPU box from the repo root.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/ingest_trace
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/ingest_trace -- python3 $R/tools/ingest_trace.py
