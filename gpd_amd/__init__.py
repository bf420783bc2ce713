"""gpd_amd — MI355X (gfx950) implementation of the GPD hot path behind a C-ABI.

csrc/      HIP kernels + the C-ABI (libgpd_hip.so, declared in include/gpd_hip.h)
api.py     ctypes binding used by the tests and bench.py
synth.py   synthetic benchmark clouds and LeNet parameters
"""
