"""CPU oracle for the swiftwatcher segment path.  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never from swiftwatcher_amd/.  See reference_path.py and swk_oracle.c.
"""
