// oracle/shim/pcl/features/normal_3d.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
