#!/bin/bash
# Runs HERE: builds yolo-tf_amd/lib/libyolohead_ubsan.so, the library with UndefinedBehaviorSanitizer on its HOST
# code (device code is untouched), diagnostics printed by the shared runtime, nothing preloaded.  To run the suites
# against it on the GPU box, copy it over libyolohead.so inside the call (the box's tree is a scratch copy):
#   gpurun -- 'cp yolo-tf_amd/lib/libyolohead_ubsan.so yolo-tf_amd/lib/libyolohead.so &&
#              python -m pytest tests -q -s > gpurun_out/r03/pytest_ubsan.log 2>&1; grep -c "runtime error" gpurun_out/r03/pytest_ubsan.log'
# The function-type check is off: it traps on every kernel launched through a pointer to its host stub.
set -e
cd "$(dirname "$0")/../yolo-tf_amd/csrc"
mkdir -p ../_build/ubsan
for f in yh_stream yh_nms yh_loss yh_abi; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function \
        -I../../include -Xarch_host -fsanitize=undefined,bounds -Xarch_host -fno-sanitize=alignment,function -c $f.hip -o ../_build/ubsan/$f.o
done
RT=$(dirname "$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libyolohead_ubsan.so ../_build/ubsan/*.o \
    -Xarch_host -fsanitize=undefined -shared-libsan -Wl,-rpath,$RT
