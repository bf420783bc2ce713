// oracle/shim/boost/shared_ptr.hpp — test-only stand-in, see shim_all.hpp
#include "shim_all.hpp"
