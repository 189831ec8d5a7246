"""CPU oracle for the time-correlation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(``transport_analysis_amd``) may import, call or link anything in here: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
use it, and there only as the checker / the timed CPU baseline.
"""
