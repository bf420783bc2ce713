// oracle/shim/boost/pool/pool.hpp — test-only stand-in, see shim_all.hpp
#include "../shim_all.hpp"
