// oracle/shim/pcl/common/transforms.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
