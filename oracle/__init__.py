"""CPU oracle of the ICP hot path — TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (lrr_amd/) never imports this.  PARITY UNPINNED (see lrr_oracle.c).
"""
from .binding import OracleEstimator, lib, build  # noqa: F401
