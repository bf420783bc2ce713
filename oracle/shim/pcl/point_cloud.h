// oracle/shim/pcl/point_cloud.h — test-only stand-in, see shim_all.h
#include "shim_all.h"
