#!/bin/bash
# Compile only the weights-stationary conv kernels of conv_igemm.hip (seconds instead of minutes) and print ISA statistics.
# usage: tools/isa_probe.sh [extra hipcc flags]; output: /tmp/probe/conv_igemm-hip-amdgcn-amd-amdhsa-gfx950.s
set -e
mkdir -p /tmp/probe
T=$(cd "$(dirname "$0")" && pwd); cd "$T/../unina-yolo-dla_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DUNINA_CONV_PROBE "$@" -c conv_igemm.hip -o /tmp/probe/conv_igemm.o -save-temps=obj
python3 "$T/isastat.py" /tmp/probe/conv_igemm-hip-amdgcn-amd-amdhsa-gfx950.s _ZN5unina20conv_dual_head3x3_wsENS_10ConvParamsES0_i _ZN5unina23conv_dual_head3x3_ws_i8ENS_10ConvParamsES0_i _ZN5unina24conv_dual_head3x3_ws_s16ENS_10ConvParamsES0_i
