// oracle/shim/boost/random/lagged_fibonacci.hpp — test-only stand-in, see shim_all.hpp
#include "../shim_all.hpp"
