// oracle/shim/pcl/kdtree/kdtree.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
