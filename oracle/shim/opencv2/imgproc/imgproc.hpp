// oracle/shim/opencv2/imgproc/imgproc.hpp — test-only stand-in, see shim_all.hpp
#include "../shim_all.hpp"
