#!/bin/bash
# Builds ab/libgpd_hip_<NAME>.so from the working tree with a sed script applied to ONE source of the temporary copy — timing
# experiments (a phase removed, a constant changed) without experiment scaffolding in the shipped sources.
#   profiles/mkpatched.sh NAME FILE 'sed-script'        (release build; A/B with GPD_HIP_LIB=ab/libgpd_hip_<NAME>.so)
#   profiles/mkpatched.sh NAME FILE @some.patch         (a unified diff of FILE instead of a sed script, e.g. profiles/img_exits.patch:
#                                                        the early-return points of the image kernels for profiles/img_phases.sh)
#   EXTRA=-DSOMETHING is handed on to make
set -e
NAME=$1; FILE=$2; SED=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/gpd_amd/csrc $T/include $ROOT/ab
cp $ROOT/gpd_amd/csrc/*.hip $ROOT/gpd_amd/csrc/*.h $ROOT/gpd_amd/csrc/*.cpp $ROOT/gpd_amd/csrc/Makefile $T/gpd_amd/csrc/; cp $ROOT/include/*.h $T/include/
if [ "${SED:0:1}" = "@" ]; then patch -s $T/gpd_amd/csrc/$FILE "${SED:1}"; else sed -i -e "$SED" $T/gpd_amd/csrc/$FILE; fi
if cmp -s $T/gpd_amd/csrc/$FILE $ROOT/gpd_amd/csrc/$FILE; then echo "the sed script changed nothing" >&2; exit 1; fi
make -s -C $T/gpd_amd/csrc -j8 EXTRA="${EXTRA:-}" > /dev/null
cp $T/gpd_amd/libgpd_hip.so $ROOT/ab/libgpd_hip_$NAME.so
rm -rf $T
echo "ab/libgpd_hip_$NAME.so"
