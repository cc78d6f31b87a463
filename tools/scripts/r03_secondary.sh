#!/bin/bash
# Round-3 records of the "next rows" on the final build (GPU box, repo root): incremental ingest latency (N1), recency
# query (N3), concurrent searches, the exchange step with a one-rank communicator, 100 M claims on one GPU.
set -e
O=gpurun_out/r03sec; mkdir -p $O
echo "== ingest latency"; timeout -k 10 300 python3 tools/ingest_latency.py > $O/r03_ingest_latency_1m.json 2> $O/ingest_latency.err
echo "== recent";         timeout -k 10 300 python3 tools/bench_recent.py > $O/r03_recent_10m.json 2> $O/recent.err
echo "== concurrent";     timeout -k 10 200 python3 tools/concurrent_search.py 1000000 2000 > $O/r03_concurrent_1m.json 2> $O/concurrent.err
echo "== sharded overhead"; timeout -k 10 300 python3 tools/sharded_overhead.py > $O/r03_sharded_overhead_1m.json 2> $O/sharded.err
echo "== 100 M"; timeout -k 10 900 python3 tools/run_100m.py > $O/r03_100m_single_gpu.json 2> $O/run_100m.err
echo all done
