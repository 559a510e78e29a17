"""CPU ORACLE — test infrastructure only (PARITY UNPINNED; see dvs_oracle.hpp header).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
from .oracle import Oracle, build, set_threads, LIB_PATH  # noqa: F401
