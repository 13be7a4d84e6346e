#!/bin/bash
# Builds variant libraries for same-box A/B runs: tools/ab_build.sh NAME "-DMACRO=VALUE ..." -> unina-yolo-dla_amd/libunina_NAME.so
# (boxes of the pool differ by +-3 % and run to run by +-1 us per kernel: only comparisons inside ONE gpurun call count).
set -e
NAME=$1; shift
cd "$(dirname "$0")/../unina-yolo-dla_amd/csrc"
mkdir -p _ab_$NAME
for u in conv_igemm c3k2_fused head_fused block_dual conv_pair stem_pool postprocess preprocess engine; do
  extra=""; case $u in postprocess|preprocess) extra="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-result $extra "$@" -c $u.hip -o _ab_$NAME/$u.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libunina_$NAME.so _ab_$NAME/*.o
echo built ../libunina_$NAME.so
