#!/bin/bash
# Developer script: builds dftpav_amd/variants/libdftpav_hip_<name>.so with solver_ref.hip compiled with extra flags
# (-D switches of that file, -mllvm options); select it at run time with DFTPAV_LIB=<path>.
#   scripts/build_ref_variant.sh inl -DDFTPAV_REF_TL_ATTR=__forceinline__
set -e
name=$1; shift
cd "$(dirname "$0")/../dftpav_amd/csrc"
mkdir -p ../variants /tmp/refvariant_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-pass-failed -DDFTPAV_REF_PART=1 "$@" \
  -c solver_ref.hip -o /tmp/refvariant_$name/solver_ref.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libdftpav_hip_$name.so solver.o /tmp/refvariant_$name/solver_ref.o solver_ref_wide.o corridor.o validate.o states.o shot.o fit.o frontend.o restart.o capi.o
echo built ../variants/libdftpav_hip_$name.so
