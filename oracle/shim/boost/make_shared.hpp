// oracle/shim/boost/make_shared.hpp — test-only stand-in, see shim_all.hpp
#include "shim_all.hpp"
