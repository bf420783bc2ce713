#!/bin/bash
# Builds ab/libgpd_hip_<NAME>.so from the working tree AS IT IS (a copy in a temporary directory; the in-tree objects and library
# are not touched) — for A/B runs of an edit in progress against the committed library:  profiles/mkvariant.sh NAME
set -e
NAME=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/gpd_amd/csrc $T/include $ROOT/ab
cp $ROOT/gpd_amd/csrc/*.hip $ROOT/gpd_amd/csrc/*.h $ROOT/gpd_amd/csrc/*.cpp $ROOT/gpd_amd/csrc/Makefile $T/gpd_amd/csrc/; cp $ROOT/include/*.h $T/include/
make -s -C $T/gpd_amd/csrc -j8 EXTRA="${EXTRA:-}" > /dev/null
cp $T/gpd_amd/libgpd_hip.so $ROOT/ab/libgpd_hip_$NAME.so
if [ -n "${KEEP_OBJ:-}" ]; then cp $T/gpd_amd/csrc/$KEEP_OBJ.o $ROOT/ab/$KEEP_OBJ.$NAME.o; fi
rm -rf $T
echo "ab/libgpd_hip_$NAME.so"
