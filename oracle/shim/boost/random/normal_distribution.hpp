// oracle/shim/boost/random/normal_distribution.hpp — test-only stand-in, see shim_all.hpp
#include "../shim_all.hpp"
