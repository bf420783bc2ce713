// oracle/shim/boost/random/variate_generator.hpp — test-only stand-in, see shim_all.hpp
#include "../shim_all.hpp"
