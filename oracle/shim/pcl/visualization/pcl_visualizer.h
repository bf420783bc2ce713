// oracle/shim/pcl/visualization/pcl_visualizer.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
