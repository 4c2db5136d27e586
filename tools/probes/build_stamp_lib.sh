#!/bin/bash
# dev container: tools/probes/libfmri_stamp.so = the library with the named kernel TUs compiled with -DFMRI_STAMP
# usage: tools/probes/build_stamp_lib.sh igemm_c5w [igemm_c5 ...]     (after fmri_hip.build has produced csrc/build/*.o)
set -e
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/thesis-fmri-reconstruction_amd/csrc
T=$(mktemp -d)
objs=$(ls $C/build/*.o)
for tu in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-result -DFMRI_STAMP -c $C/$tu.hip -o $T/$tu.o
  objs=$(echo "$objs" | grep -v "/$tu.o$"); objs="$objs $T/$tu.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/probes/libfmri_stamp.so $objs
rm -rf $T
