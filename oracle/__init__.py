"""CPU oracle of the cudabrot hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; nothing
under cudabrot_amd/ does.  See buddha_oracle.h for how its parity is pinned.
"""
