// oracle/shim/opencv2/core.hpp — test-only stand-in, see shim_all.hpp
#include "shim_all.hpp"
