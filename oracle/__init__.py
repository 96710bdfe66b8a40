"""TEST INFRASTRUCTURE: CPU restatements of the reference path and the golden-vector generator.  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product path under clipself_amd/."""
