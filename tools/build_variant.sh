#!/bin/bash
# A build of the library with edited constants, for A/B timing (tools/time_ab.py, tools/bench_with_lib.py):
#   tools/build_variant.sh <name> '<sed expression applied to every source>' ...
# writes yolo-tf_amd/lib/lib_<name>.so from a scratch copy of yolo-tf_amd/csrc.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$ROOT/yolo-tf_amd/_build/var_$NAME
rm -rf $W; mkdir -p $W
cp $ROOT/yolo-tf_amd/csrc/*.hip $ROOT/yolo-tf_amd/csrc/*.h $W/
for e in "$@"; do sed -i "$e" $W/*.hip $W/*.h; done
for f in yh_stream yh_nms yh_loss yh_abi; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
        -I$ROOT/include -c $W/$f.hip -o $W/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/yolo-tf_amd/lib/lib_$NAME.so $W/*.o
echo built lib_$NAME.so
