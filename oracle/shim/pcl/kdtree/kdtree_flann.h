// oracle/shim/pcl/kdtree/kdtree_flann.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
