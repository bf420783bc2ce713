// oracle/shim/pcl/filters/extract_indices.h — test-only stand-in, see shim_all.h
#include "../shim_all.h"
