"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see oracle/minigrid_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
