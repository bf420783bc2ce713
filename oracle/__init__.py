"""CPU oracle of the GPD hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  See oracle/gpd_oracle.cpp for the definition and the "parity
unpinned" statement.
"""
from .oracle import *  # noqa: F401,F403
