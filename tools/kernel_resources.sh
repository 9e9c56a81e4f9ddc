#!/bin/bash
# Registers, LDS and occupancy of every kernel in the library, as the compiler reports them
# (-Rpass-analysis=kernel-resource-usage; device code only, nothing is linked or run):
#   bash tools/kernel_resources.sh > profiles/<tag>_kernel_resources.txt
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
python3 -m facedetection_amd._build > /dev/null      # writes the build-id header fd_core.hip includes
for f in fd_core fd_boxes fd_image fd_conv fd_conv_split fd_conv_panel fd_conv_tailw fd_conv_fused fd_winograd fd_detector; do
    extra=""
    case $f in fd_boxes|fd_image) extra="-ffp-contract=off";; esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $extra -I $ROOT/facedetection_amd/csrc/build --cuda-device-only \
        -Rpass-analysis=kernel-resource-usage -c $ROOT/facedetection_amd/csrc/$f.hip -o $TMP/$f.co 2>&1 |
        grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" |
        sed 's/.*remark: //; s/\[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - - |
        sed 's/^ *Function Name: //; s/[ \t]\+/ /g' | while read name rest; do
            echo "$f.hip  $(echo $name | c++filt | sed 's/(.*//; s/^void //')  |$rest"
        done
done
rm -rf $TMP
