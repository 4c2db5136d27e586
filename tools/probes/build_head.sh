#!/bin/bash
# dev container: tools/probes/libfmri_head.so = the current objects with the COMMITTED (git HEAD) version of some kernel TUs,
# for a same-box A/B of an uncommitted kernel change (tools/probes/ab_variants.sh head).  usage: build_head.sh <tu>[,<tu>...]
set -e
R=$(cd $(dirname $0)/../.. && pwd)
C=thesis-fmri-reconstruction_amd/csrc
cd $R
T=$(mktemp -d)
objs=$(ls $C/build/*.o)
for tu in ${1//,/ }; do
  git show HEAD:$C/$tu.hip > $C/zz_head_$tu.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-result -c $C/zz_head_$tu.hip -o $T/$tu.o
  rm $C/zz_head_$tu.hip
  objs=$(echo "$objs" | grep -v "/$tu.o$"); objs="$objs $T/$tu.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/libfmri_head.so $objs
rm -rf $T
