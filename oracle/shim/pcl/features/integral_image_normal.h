// oracle/shim/pcl/features/integral_image_normal.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
