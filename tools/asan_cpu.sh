#!/bin/bash
# HERE (no GPU needed): the host side of the library under AddressSanitizer.  The host halves of api / farm / bucket /
# host_mesher / ply are compiled with -Xarch_host -fsanitize=address (device code untouched; the GPU pool has no sanitizer),
# linked with the other objects of the normal build, put in the library's place for the run of the CPU test suite and taken
# out again.  usage: bash tools/asan_cpu.sh [pytest args]     (the C++ link test is skipped: its hosts are not instrumented)
set -e
cd "$(dirname "$0")/.."
make -C mlsgpu_amd/csrc -s
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
mkdir -p /tmp/mlsgpu_asan
F="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fvisibility=hidden -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer"
cd mlsgpu_amd/csrc
for f in api farm bucket host_mesher ply; do /opt/rocm/bin/hipcc $F -c -o /tmp/mlsgpu_asan/$f.o $f.hip 2> /dev/null; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -shared-libasan -o /tmp/mlsgpu_asan/libmlsgpu_hip.so \
    /tmp/mlsgpu_asan/api.o octree.o mls.o marching.o /tmp/mlsgpu_asan/farm.o /tmp/mlsgpu_asan/bucket.o mesher.o /tmp/mlsgpu_asan/host_mesher.o /tmp/mlsgpu_asan/ply.o
cd ../..
cp mlsgpu_amd/libmlsgpu_hip.so /tmp/mlsgpu_asan/keep.so
trap 'cp /tmp/mlsgpu_asan/keep.so mlsgpu_amd/libmlsgpu_hip.so' EXIT
cp /tmp/mlsgpu_asan/libmlsgpu_hip.so mlsgpu_amd/libmlsgpu_hip.so
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 python -m pytest tests -q -m "not gpu" --deselect tests/test_host_cpp.py "$@"
