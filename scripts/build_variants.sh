#!/bin/bash
# Side-by-side builds of libesl_hip.so with different tuning macros of esl_capi.hip (e.g. -DESL_LIN_MIN_WAVES=4).
# Usage: scripts/build_variants.sh name "-DMACRO=value ..." [name flags]...
# The variants land in object-oriented-slam_amd/csrc/variants/<name>.so; select one with ESL_HIP_LIB=<path>.
set -e
cd "$(dirname "$0")/../object-oriented-slam_amd/csrc"
mkdir -p variants
make -s
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -munsafe-fp-atomics $flags -c esl_capi.hip -o variants/$name.capi.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name.so variants/$name.capi.o esl_slam.o esl_fit.o esl_init.o esl_comm.o esl_plane.o -ldl
  echo "built variants/$name.so ($flags)"
done
