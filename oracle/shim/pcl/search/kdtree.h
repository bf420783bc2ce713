// oracle/shim/pcl/search/kdtree.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
