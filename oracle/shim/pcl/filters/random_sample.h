// oracle/shim/pcl/filters/random_sample.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
