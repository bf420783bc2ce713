// oracle/shim/pcl/point_types.h — test-only stand-in, see shim_all.h
#include "shim_all.h"
