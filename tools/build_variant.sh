#!/bin/bash
# Development aid: a variant build of libagpl.so for same-box A/B runs (tools/kbench.py, tools/time_update.py, tools/bench_with_lib.py: AGPL_LIB_AB=<name>):
#   tools/build_variant.sh <suffix> <file.hip> "<extra -D flags>"   ->  augmentedgplikelihoods.jl_amd/libagpl_<suffix>.so
# The other objects are the regular build's (run make first).
set -e
cd "$(dirname "$0")/../augmentedgplikelihoods.jl_amd/csrc"
SUF=$1; SRC=$2; FLAGS=$3
COMMON="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function -fno-slp-vectorize"
EXTRA=""
case $SRC in agpl_synth.hip) EXTRA="-ffp-contract=off";; agpl_ops.hip) EXTRA="-ffp-contract=off -mllvm -disable-machine-licm";; esac  # (as the Makefile)
/opt/rocm/bin/hipcc $COMMON $EXTRA $FLAGS -c $SRC -o /tmp/variant_$SUF.o
OBJS=""
for f in agpl_syrk agpl_core agpl_ops agpl_mfma agpl_update agpl_synth agpl_dense agpl_split agpl_factor agpl_plan; do
  if [ "$f.hip" == "$SRC" ]; then OBJS="$OBJS /tmp/variant_$SUF.o"; else OBJS="$OBJS $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libagpl_$SUF.so $OBJS -L/opt/rocm/lib -lrocsolver -lrocblas -Wl,-rpath,/opt/rocm/lib
echo built ../libagpl_$SUF.so
