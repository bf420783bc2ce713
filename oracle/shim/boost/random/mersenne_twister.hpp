// oracle/shim/boost/random/mersenne_twister.hpp — test-only stand-in, see shim_all.hpp
#include "../shim_all.hpp"
