#!/bin/bash
# A/B baseline: the library built from the kernel sources of an earlier commit (same flags as the product), as lib/libreinlife_hip_<tag>.so
#   bash tools/build_at_commit.sh <commit> <tag>      then   python tools/run_ab.py reinlife_amd/lib/libreinlife_hip_<tag>.so reinlife_amd/lib/libreinlife_hip.so
set -e
cd "$(dirname "$0")/.."
C=$1; TAG=$2; D=/tmp/rl_src_$TAG
rm -rf $D && mkdir -p $D/reinlife_amd/csrc $D/include
for f in $(git ls-tree --name-only $C reinlife_amd/csrc/); do git show $C:$f > $D/$f; done
git show $C:include/reinlife_hip.h > $D/include/reinlife_hip.h
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden --offload-compress -Wno-unused-function"
cd $D/reinlife_amd/csrc
/opt/rocm/bin/hipcc $FLAGS -c rl_world.hip -o w.o & /opt/rocm/bin/hipcc $FLAGS -DRL_RUN_UNIT=0 -c rl_run.hip -o r0.o & /opt/rocm/bin/hipcc $FLAGS -DRL_RUN_UNIT=1 -c rl_run.hip -o r1.o &
/opt/rocm/bin/hipcc $FLAGS -c rl_policy.hip -o p.o & /opt/rocm/bin/hipcc $FLAGS -c rl_capi.hip -o c.o & wait
cd - > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$D/reinlife_amd/csrc/exports.map -o reinlife_amd/lib/libreinlife_hip_$TAG.so $D/reinlife_amd/csrc/{w,r0,r1,p,c}.o
ls -la reinlife_amd/lib/libreinlife_hip_$TAG.so
