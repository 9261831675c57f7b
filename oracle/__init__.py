"""CPU oracle for the placement hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; nothing under usher_amd/ does.
"""
