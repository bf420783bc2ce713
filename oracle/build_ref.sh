#!/bin/bash
# oracle/build_ref.sh — builds oracle/_ref/ref_fixture from the REFERENCE's own sources (read where they lie, never
# copied) and oracle/ref_fixture.cpp, then generates the reference fixtures that tests/test_ref_fixture.py
# consumes.  Needs what the reference needs: Eigen 3, PCL >= 1.9 (common io kdtree search features filters
# segmentation visualization), OpenCV >= 3.4, and the reference tree.  When any of it is missing — as in the
# build container of this repository — it says so and exits 0 without producing anything: the oracle then
# stays "parity unpinned" against the reference binary and the test reports itself skipped.
#   REF=/root/reference oracle/build_ref.sh
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
REF=${REF:-/root/reference}
OUT=$HERE/_ref
skip() { echo "oracle/build_ref.sh: SKIPPED — $1 (the oracle stays unpinned against the reference binary)"; exit 0; }
[ -d "$REF/src/gpd" ] || skip "no reference tree at $REF"
command -v pkg-config > /dev/null || skip "pkg-config not found"
pc() { pkg-config --list-all 2> /dev/null | awk '{print $1}' | grep -E "^$1(-[0-9.]+)?$" | sort -V | tail -1; }
MODS=""
for m in eigen3 opencv4 pcl_common pcl_io pcl_kdtree pcl_search pcl_features pcl_filters pcl_segmentation pcl_visualization; do
  found=$(pc $m)
  if [ -z "$found" ] && [ "$m" = opencv4 ]; then found=$(pc opencv); fi
  [ -n "$found" ] || skip "dependency $m not found by pkg-config"
  MODS="$MODS $found"
done
mkdir -p "$OUT"
SRCS=$(ls $REF/src/gpd/candidate/*.cpp $REF/src/gpd/descriptor/*.cpp $REF/src/gpd/util/*.cpp \
          $REF/src/gpd/net/classifier.cpp $REF/src/gpd/net/eigen_classifier.cpp $REF/src/gpd/net/conv_layer.cpp \
          $REF/src/gpd/net/dense_layer.cpp $REF/src/gpd/net/layer.cpp)
# the reference's own flags (CMakeLists.txt:3-4,29), single-threaded so that its racy OpenMP loops
# run in order
set -x
g++ -std=c++17 -O3 -march=native -mavx2 -mfma -fopenmp -I"$REF/include" $(pkg-config --cflags $MODS) \
    -o "$OUT/ref_fixture" "$HERE/ref_fixture.cpp" $SRCS $(pkg-config --libs $MODS) || { set +x; skip "the reference did not compile"; }
set +x
# inputs + fixtures (tests/golden/make_ref_inputs.py writes the PCD / normals / samples / parameter files)
WORK=$OUT/work
mkdir -p "$WORK"
python "$ROOT/tests/golden/make_ref_inputs.py" "$WORK" || skip "could not write the fixture inputs"
export OMP_NUM_THREADS=1
for C in 15 12 3; do
  "$OUT/ref_fixture" "$WORK/cloud.pcd" "$WORK/normals.f32" "$WORK/samples.i32" "$WORK/params/" $C "$ROOT/tests/golden/ref_fixture_c$C.bin" || skip "ref_fixture failed for $C channels"
done
echo "oracle/build_ref.sh: wrote tests/golden/ref_fixture_c{15,12,3}.bin — commit them; tests/test_ref_fixture.py now pins the oracle"
