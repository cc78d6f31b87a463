#!/bin/bash
# Host-side AddressSanitizer build of liblbhip + the C harness tools/cabi_check.c (run on the build machine), and the run
# itself (on the GPU box):   bash tools/scripts/cabi_asan.sh build   |   bash tools/scripts/cabi_asan.sh run [claims]
# Only HOST code is instrumented (-Xarch_host -fsanitize=$SAN): GPU sanitizers are not available on this pool.
set -e
cd "$(dirname "$0")/../.."
SAN=${SAN:-address}   # (SAN=thread builds, but under ThreadSanitizer's address-space layout the HIP runtime finds no device)
B=tools/bin/$SAN
case "$1" in
build)
    mkdir -p $B
    F="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-function -Xarch_host -fsanitize=$SAN -Xarch_host -fno-omit-frame-pointer"
    for f in embed search api; do /opt/rocm/bin/hipcc $F -c lbm_amd/csrc/$f.hip -o $B/$f.o; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -Xarch_host -fsanitize=$SAN -shared -fPIC $B/embed.o $B/search.o $B/api.o -o $B/liblbhip_asan.so
    /opt/rocm/bin/hipcc -x c -std=c11 -O1 -g -Xarch_host -fsanitize=$SAN -Xarch_host -fno-omit-frame-pointer -Iinclude -c tools/cabi_check.c -o $B/cabi_check.o
    /opt/rocm/bin/hipcc -Xarch_host -fsanitize=$SAN $B/cabi_check.o -L$B -llbhip_asan -lpthread -Wl,-rpath,'$ORIGIN' -o $B/cabi_check
    ls -la $B/cabi_check $B/liblbhip_asan.so
    ;;
run)
    ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0 timeout -k 10 300 $B/cabi_check ${2:-50000}
    ;;
*) echo "usage: $0 build | run [claims]"; exit 2;;
esac
