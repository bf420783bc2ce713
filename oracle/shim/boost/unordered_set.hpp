// oracle/shim/boost/unordered_set.hpp — test-only stand-in, see shim_all.hpp
#include "shim_all.hpp"
