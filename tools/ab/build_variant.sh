#!/bin/bash
# A variant of the library for A/B runs (tools/ab/ab.sh, CLIORA_CHART_LIB): tools/ab/build_variant.sh <name> [-DFLAG=.. ...]
# Only api_mlp.hip is recompiled with the extra flags (the level kernels live there); the other objects are the default build's.
# -> cliora_amd/libvar_<name>.so   (travels to the GPU box with the snapshot; git-ignored)
set -e
name=$1; shift
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/cliora_amd/csrc
python -m cliora_amd.build > /dev/null
mkdir -p /tmp/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-pass-failed -Wno-unused-value "$@" -x hip -c $C/api_mlp.hip -o /tmp/var_$name/api_mlp.hip.o
objs=$(ls $C/build/*.o | grep -v api_mlp.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/cliora_amd/libvar_$name.so $objs /tmp/var_$name/api_mlp.hip.o
echo built cliora_amd/libvar_$name.so "$@"
