// oracle/shim/pcl/filters/statistical_outlier_removal.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
