// oracle/shim/boost/random/uniform_real.hpp — test-only stand-in, see shim_all.hpp
#include "../shim_all.hpp"
