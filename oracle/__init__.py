"""TEST INFRASTRUCTURE ONLY — CPU oracle of the TRACE hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (trace_amd/) must never do so.
"""
