#!/bin/bash
# Out-of-tree build of a variant of libltk_hip.so for in-job A/Bs (LTK_LIB selects the library at run time):
#   scripts/build_variant.sh <name> [extra hipcc flags ...]   ->  ab_libs/libltk_hip_<name>.so
# e.g. scripts/build_variant.sh ablate -DLTK_ABLATE_BUILD=1
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${SRC:-$ROOT/livetalking_amd/csrc}
OBJ=$ROOT/build/$NAME
mkdir -p $OBJ $ROOT/ab_libs
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $*"
pids=()
for f in tune conv_mfma conv3_mfma conv7_mfma rowgemm misc_kernels egress_kernels nn_kernels musetalk engine; do
  extra=""
  case $f in misc_kernels|egress_kernels) extra="-ffp-contract=off";; nn_kernels) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
  if [ ! -f $OBJ/$f.o ] || [ $SRC/$f.hip -nt $OBJ/$f.o ] || [ -n "$(find $SRC -name '*.h' -newer $OBJ/$f.o)" ] || [ -n "$FORCE" ]; then
    /opt/rocm/bin/hipcc $FLAGS $extra -c $SRC/$f.hip -o $OBJ/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab_libs/libltk_hip_$NAME.so $OBJ/*.o
echo built ab_libs/libltk_hip_$NAME.so
