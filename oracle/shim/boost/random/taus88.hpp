// oracle/shim/boost/random/taus88.hpp — test-only stand-in, see shim_all.hpp
#include "../shim_all.hpp"
