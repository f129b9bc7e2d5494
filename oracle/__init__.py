"""CPU restatement of the reference's algorithms for this path.  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and the cpu_baseline leg of bench.py as the checker, never by the product path."""
