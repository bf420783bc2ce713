#!/bin/bash
# Builds ab/libgpd_hip_<NAME>.so from a git revision (default: the working tree) for A/B runs on one GPU box
# (GPD_HIP_LIB=ab/libgpd_hip_<NAME>.so python bench.py ...).   profiles/mkvariant.sh NAME [REV]
set -e
NAME=$1; REV=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/gpd_amd/csrc $T/include $ROOT/ab
if [ -n "$REV" ]; then
  for f in $(git -C $ROOT ls-tree --name-only $REV gpd_amd/csrc/ include/); do git -C $ROOT show $REV:$f > $T/$f; done
else
  cp $ROOT/gpd_amd/csrc/*.hip $ROOT/gpd_amd/csrc/*.h $ROOT/gpd_amd/csrc/*.cpp $ROOT/gpd_amd/csrc/Makefile $T/gpd_amd/csrc/; cp $ROOT/include/*.h $T/include/
fi
make -s -C $T/gpd_amd/csrc -j8 > /dev/null
# the profiling build when the revision has one (round 4 on: the measurement switches live there), else the only library
if [ -f $T/gpd_amd/libgpd_hip_prof.so ]; then cp $T/gpd_amd/libgpd_hip_prof.so $ROOT/ab/libgpd_hip_$NAME.so; else cp $T/gpd_amd/libgpd_hip.so $ROOT/ab/libgpd_hip_$NAME.so; fi
rm -rf $T
echo "ab/libgpd_hip_$NAME.so"
