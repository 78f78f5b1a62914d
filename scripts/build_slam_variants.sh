#!/bin/bash
# Side-by-side builds of libesl_hip.so with different macros of esl_slam.hip / esl_chol.hpp (the dense solver lives there).
# Usage: scripts/build_slam_variants.sh name "-DMACRO ..." [name flags]...   -> object-oriented-slam_amd/csrc/variants/<name>.so (ESL_HIP_LIB=<path>)
set -e
cd "$(dirname "$0")/../object-oriented-slam_amd/csrc"
mkdir -p variants
make -s
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -munsafe-fp-atomics $flags -c esl_slam.hip -o variants/$name.slam.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name.so esl_capi.o variants/$name.slam.o esl_fit.o esl_init.o esl_comm.o esl_plane.o -ldl
  echo "built variants/$name.so ($flags)"
done
