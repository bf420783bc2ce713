// oracle/shim/pcl/filters/normal_refinement.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
