#!/bin/bash
# dev container: tools/probes/libfmri_<name>.so = the library with some kernel TUs compiled with extra -D flags (stamps, phase
# ablations, parameter sweeps).  usage: tools/probes/build_variant.sh <name> <tu>[,<tu>...] [-DX=1 ...]   (after fmri_hip.build)
set -e
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/thesis-fmri-reconstruction_amd/csrc
name=$1; tus=$2; shift 2
T=$(mktemp -d)
objs=$(ls $C/build/*.o)
for tu in ${tus//,/ }; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-result "$@" -c $C/$tu.hip -o $T/$tu.o
  objs=$(echo "$objs" | grep -v "/$tu.o$"); objs="$objs $T/$tu.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/probes/libfmri_$name.so $objs
rm -rf $T
