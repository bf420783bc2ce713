#!/bin/bash
# oracle/build_ref.sh — builds oracle/_ref/ from the REFERENCE's own sources, read where they lie (never copied).
# TEST INFRASTRUCTURE.  Two tiers:
#
#  tier A (runs in the build container): the reference's translation units for the hot path — candidate/, descriptor/,
#    net/ (Eigen back-end), util/{cloud,point_list,eigen_utils,config_file}, clustering.cpp, grasp_detector.cpp — compiled
#    UNMODIFIED through the test-only third-party subsets of oracle/shim/ (Eigen, PCL, OpenCV, Boost: interface subsets
#    written for this repository, numerics restated from memory — read their headers), plus oracle/ref_glue.cpp (C entry
#    points) and oracle/ref_plot_stub.cpp, into oracle/_ref/libgpd_ref.so.  This pins the reference's IN-TREE logic;
#    third-party numerics stay unpinned.  Not compiled: util/plot.cpp (VTK), data_generator.cpp (HDF5),
#    sequential_importance_sampling.cpp, the Caffe / OpenVINO back-ends, net/layer.cpp (includes a header that does not
#    exist; the reference's own CMake does not build it either).
#    Flags: the reference's CMake ends up at -std=gnu++14 (CMAKE_CXX_STANDARD 14 is appended after CMAKE_CXX_FLAGS) with
#    -O3 -mavx2 -mfma -fopenmp.  Here: -std=gnu++14 -O2 -mavx2 -mfma -ffp-contract=off, NO -fopenmp (single-thread order
#    is the definition; `-include omp.h` keeps omp_get_wtime declared), contraction off as in the oracle.
#
#  tier B (needs the real Eigen 3 / PCL >= 1.9 / OpenCV >= 3.4 via pkg-config): the same sources against the real
#    libraries + oracle/ref_fixture.cpp -> tests/golden/ref_fixture_c{15,12,3}.bin.  Skips (exit 0) when they are absent.
#
#   REF=/root/reference oracle/build_ref.sh [A|B|all]
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
REF=${REF:-/root/reference}
OUT=${OUT:-$HERE/_ref}   # (profiles/thirdparty_sensitivity.py builds perturbed variants into oracle/_ref/perturb<N> with EXTRA=-DGPD_SHIM_PERTURB=<N>)
EXTRA=${EXTRA:-}
WHAT=${1:-all}
skip() { echo "oracle/build_ref.sh: SKIPPED — $1"; exit 0; }
[ -d "$REF/src/gpd" ] || skip "no reference tree at $REF (prebuilt oracle/_ref, if any, is used as it is)"
mkdir -p "$OUT/obj"

TUS="candidate/finger_hand candidate/antipodal candidate/hand candidate/local_frame candidate/hand_geometry candidate/hand_set
candidate/frame_estimator candidate/hand_search candidate/candidates_generator util/point_list util/eigen_utils util/config_file
util/cloud net/conv_layer net/dense_layer net/eigen_classifier net/classifier descriptor/image_strategy
descriptor/image_1_channels_strategy descriptor/image_3_channels_strategy descriptor/image_12_channels_strategy
descriptor/image_15_channels_strategy descriptor/image_generator descriptor/image_geometry clustering grasp_detector"

tier_a() {
  local CXXFLAGS="-std=gnu++14 -O2 -mavx2 -mfma -ffp-contract=off -fPIC -w -include omp.h -I$HERE/shim -I$REF/include $EXTRA"
  local objs="" pids="" fail=0
  for f in $TUS; do
    local o="$OUT/obj/$(echo $f | tr / _).o" extra=""
    [ "$f" = candidate/hand_set ] && extra="-include $HERE/shim/gpd_ref_no_jitter.h"
    objs="$objs $o"
    if [ ! -f "$o" ] || [ "$REF/src/gpd/$f.cpp" -nt "$o" ] || [ -n "$(find "$HERE/shim" -newer "$o" -type f | head -1)" ]; then
      g++ $CXXFLAGS $extra -c "$REF/src/gpd/$f.cpp" -o "$o" &
      pids="$pids $!"
    fi
  done
  for p in $pids; do wait $p || fail=1; done
  [ $fail = 0 ] || { echo "oracle/build_ref.sh: tier A FAILED to compile a reference translation unit"; exit 1; }
  g++ $CXXFLAGS -fno-access-control -c "$HERE/ref_glue.cpp" -o "$OUT/obj/ref_glue.o" \
    && g++ $CXXFLAGS -shared -o "$OUT/libgpd_ref.so" "$OUT/obj/ref_glue.o" "$HERE/ref_plot_stub.cpp" $objs -lgomp \
    || { echo "oracle/build_ref.sh: tier A FAILED to link"; exit 1; }
  echo "oracle/build_ref.sh: tier A built $OUT/libgpd_ref.so (reference sources through oracle/shim)"
}

tier_b() {
  command -v pkg-config > /dev/null || skip "tier B: pkg-config not found"
  pc() { pkg-config --list-all 2> /dev/null | awk '{print $1}' | grep -E "^$1(-[0-9.]+)?$" | sort -V | tail -1; }
  local MODS=""
  for m in eigen3 opencv4 pcl_common pcl_io pcl_kdtree pcl_search pcl_features pcl_filters pcl_segmentation pcl_visualization; do
    local found=$(pc $m)
    if [ -z "$found" ] && [ "$m" = opencv4 ]; then found=$(pc opencv); fi
    [ -n "$found" ] || skip "tier B: dependency $m not found by pkg-config (the oracle stays unpinned against the reference BINARY; tier A pins the in-tree logic)"
    MODS="$MODS $found"
  done
  local SRCS=$(ls $REF/src/gpd/candidate/*.cpp $REF/src/gpd/descriptor/*.cpp $REF/src/gpd/util/*.cpp \
            $REF/src/gpd/net/classifier.cpp $REF/src/gpd/net/eigen_classifier.cpp $REF/src/gpd/net/conv_layer.cpp \
            $REF/src/gpd/net/dense_layer.cpp)
  set -x
  g++ -std=gnu++14 -O3 -march=native -mavx2 -mfma -fopenmp -I"$REF/include" $(pkg-config --cflags $MODS) \
      -o "$OUT/ref_fixture" "$HERE/ref_fixture.cpp" $SRCS $(pkg-config --libs $MODS) || { set +x; skip "tier B: the reference did not compile"; }
  set +x
  local WORK=$OUT/work
  mkdir -p "$WORK"
  python "$ROOT/tests/golden/make_ref_inputs.py" "$WORK" || skip "tier B: could not write the fixture inputs"
  export OMP_NUM_THREADS=1
  for C in 15 12 3; do
    "$OUT/ref_fixture" "$WORK/cloud.pcd" "$WORK/normals.f32" "$WORK/samples.i32" "$WORK/params/" $C "$ROOT/tests/golden/ref_fixture_c$C.bin" || skip "tier B: ref_fixture failed for $C channels"
  done
  echo "oracle/build_ref.sh: tier B wrote tests/golden/ref_fixture_c{15,12,3}.bin — commit them; tests/test_ref_fixture.py now pins the oracle against the reference binary"
}

case "$WHAT" in
  A) tier_a ;;
  B) tier_b ;;
  *) tier_a; tier_b ;;
esac
