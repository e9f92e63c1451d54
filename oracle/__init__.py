"""CPU oracle for the SA-IS / BWT-table path.  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / the reported CPU baseline.  The
product (``stralg_amd``) never imports it.  Parity status: pinned (see oracle.h).
"""
from .pyoracle import *  # noqa: F401,F403
