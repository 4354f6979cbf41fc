"""CPU oracle for the geodesic hot path -- TEST INFRASTRUCTURE ONLY (parity unpinned, see
geodesic_oracle.c).  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never by the product package."""
