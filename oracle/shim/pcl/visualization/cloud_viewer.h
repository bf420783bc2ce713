// oracle/shim/pcl/visualization/cloud_viewer.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
