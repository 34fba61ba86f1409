#!/bin/bash
# Host code of the library under AddressSanitizer + UndefinedBehaviorSanitizer (device code is built as always).
#   bash tools/host_sanitizers.sh build     here: build/san/liborbx.so and the C++ programs of tests/cpp against it
#   bash tools/host_sanitizers.sh run       on a GPU box: the pytest cases that drive those programs, pointed at build/san
# The programs are linked by the same clang as the library, so that both use one sanitizer runtime (linked into the
# executable; nothing is preloaded).  Python cannot load the instrumented library, the C++ programs can.
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SAN=$ROOT/build/san
CLANG=/opt/rocm/lib/llvm/bin/clang++
SANFLAGS="-fsanitize=address -fsanitize=undefined -fno-omit-frame-pointer"
case "${1:-build}" in
build)
    mkdir -p $SAN
    cd $ROOT/orb_slam_amd/csrc
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result \
        -mllvm -amdgpu-mfma-vgpr-form -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined \
        -Xarch_host -fno-omit-frame-pointer -shared -o $SAN/liborbx.so \
        orbx_api.cpp orbx_tables.cpp orbx_io.cpp orbx_kernels.hip orbx_match.hip orbx_frame.hip
    cd $ROOT/tests/cpp
    LINK="-L$SAN -lorbx -Wl,-rpath,\$ORIGIN -Wl,-rpath-link,/opt/rocm/lib"
    for p in host_extract stereo_delayed sharded_extract matcher_calls abi_sweep; do
        $CLANG -std=c++17 -O1 -g $SANFLAGS -Wall -pthread -I$ROOT/include -o $SAN/$p $p.cpp $LINK
    done
    $CLANG -std=c++11 -O1 -g $SANFLAGS -Wall -Wextra -Imini_cv -I$ROOT/include -o $SAN/adapter_frame adapter_frame.cpp $LINK
    ls -la $SAN
    ;;
run)
    cd $ROOT
    # Leaks of the HIP runtime's own start-up allocations are not this library's.  The quarantine of freed blocks is made
    # large enough never to be recycled: this runtime's allocator for page-locked memory stops with a failed check when a
    # block it handed out is recycled after the HIP runtime has unloaded, which a recycle at process exit does.
    export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:quarantine_size_mb=4096 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
    ORBX_CPP_DIR=$SAN timeout -k 10 900 python -u -m pytest tests -m gpu -x -q -k cpp --timeout 300 --timeout-method thread
    ;;
*)
    echo "usage: $0 build|run"; exit 2;;
esac
