#!/bin/bash
# Developer script: builds dftpav_amd/variants/libdftpav_hip_<name>.so from a modified copy of solver.hip
# (a sed expression applied to it) and/or other solver flags; select it at run time with DFTPAV_LIB=<path>.
#   scripts/build_variant.sh inl 's/__attribute__((noinline)) double two_loop_lane/__forceinline__ double two_loop_lane/' [flags...]
set -e
name=$1; expr=$2; shift 2 || true
cd "$(dirname "$0")/../dftpav_amd/csrc"
mkdir -p ../variants /tmp/variant_$name
sed "$expr" solver.hip > /tmp/variant_$name/solver.hip
cp device_types.h traj_math.h cr_trig.h rs_math.h e4_plan.h /tmp/variant_$name/
if [ -n "$VARIANT_TRAJ_MATH" ]; then cp "$VARIANT_TRAJ_MATH" /tmp/variant_$name/traj_math.h; fi   # another traj_math.h for the solve kernel
/opt/rocm/bin/hipcc -O2 -mllvm -amdgpu-sched-strategy=max-ilp -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-pass-failed "$@" \
  -c /tmp/variant_$name/solver.hip -o /tmp/variant_$name/solver.o
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libdftpav_hip_$name.so /tmp/variant_$name/solver.o solver_ref.o solver_ref_wide.o corridor.o validate.o states.o shot.o fit.o frontend.o restart.o capi.o
echo built ../variants/libdftpav_hip_$name.so
