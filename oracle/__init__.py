"""CPU oracle for the keyed forward -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(keynet_amd/) never does.  See oracle/oracle.py.
"""
from .oracle import *  # noqa: F401,F403
