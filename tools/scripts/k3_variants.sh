#!/bin/bash
# A/B of K3 register-budget variants (make -C lbm_amd/csrc k3var V="..."): 10 M x 256, per-kernel times from a kernel trace
# usage: tools/scripts/k3_variants.sh <out-prefix> <variant> ...   (variant = <ring>_<threads>_<blocks per CU> as built by `make k3var`, or "default")
pre=$1; shift
for v in "$@"; do
    lib=lbm_amd/csrc/abl/liblbhip_k3_$v.so
    [ "$v" = default ] && lib=lbm_amd/liblbhip.so
    echo "=== variant $v ($lib)"
    LBM_HIP_LIB=$PWD/$lib timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/${pre}_$v -o bq -- python tools/bench_batched.py 10000000 256 8 10 2>&1 | grep "^{" || exit 1
    python tools/trace_db.py gpurun_out/${pre}_$v/bq_results.db 7 || exit 1
done
