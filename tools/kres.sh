#!/bin/bash
# Register, scratch and LDS use of every kernel of one source file (device-only compile with resource remarks):
#   tools/kres.sh yh_stream [extra hipcc flags]
F=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I$ROOT/include --cuda-device-only \
    -Rpass-analysis=kernel-resource-usage "$@" -c $ROOT/yolo-tf_amd/csrc/$F.hip -o /tmp/kres_$F.o 2>&1 |
  grep -E "Function Name|SGPRs:|VGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: //' -e 's/ \[-Rpass.*//' |
  awk '/Function Name/ {if (line) print line; line=$0; next} {gsub(/^ +/, ""); line=line "  " $0} END {print line}'
