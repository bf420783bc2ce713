// oracle/shim/pcl/segmentation/sac_segmentation.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
