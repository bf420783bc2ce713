// oracle/shim/pcl/features/normal_3d_omp.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
