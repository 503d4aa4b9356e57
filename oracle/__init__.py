"""CPU oracle for the ViDAR hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (vidar_amd) never does; it fails loudly when libvidar_hip.so is missing.
"""
