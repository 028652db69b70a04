"""CPU oracle of the REHRSeg 3D-convolutional hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; nothing under rehrseg_amd/ does.  See oracle/README.md.
"""
