// oracle/shim/pcl/search/impl/kdtree.hpp — test-only stand-in, see shim_all.h
#include "../../shim_all.h"
