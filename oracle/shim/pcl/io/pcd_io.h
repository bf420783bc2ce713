// oracle/shim/pcl/io/pcd_io.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
