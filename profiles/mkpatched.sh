#!/bin/bash
# Builds ab/libgpd_hip_<NAME>.so from the working tree with a sed script applied to ONE source of the temporary copy — timing
# experiments (a phase removed, a constant changed) without experiment scaffolding in the shipped sources.
#   profiles/mkpatched.sh NAME FILE 'sed-script'        (release build; A/B with GPD_HIP_LIB=ab/libgpd_hip_<NAME>.so)
set -e
NAME=$1; FILE=$2; SED=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/gpd_amd/csrc $T/include $ROOT/ab
cp $ROOT/gpd_amd/csrc/*.hip $ROOT/gpd_amd/csrc/*.h $ROOT/gpd_amd/csrc/*.cpp $ROOT/gpd_amd/csrc/Makefile $T/gpd_amd/csrc/; cp $ROOT/include/*.h $T/include/
sed -i -e "$SED" $T/gpd_amd/csrc/$FILE
if cmp -s $T/gpd_amd/csrc/$FILE $ROOT/gpd_amd/csrc/$FILE; then echo "the sed script changed nothing" >&2; exit 1; fi
make -s -C $T/gpd_amd/csrc -j8 > /dev/null
cp $T/gpd_amd/libgpd_hip.so $ROOT/ab/libgpd_hip_$NAME.so
rm -rf $T
echo "ab/libgpd_hip_$NAME.so"
