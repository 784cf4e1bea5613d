"""CPU parity oracle for the Sparsebit fake-quant hot path -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; sparsebit_amd/ never does (there is no CPU fallback in the product).
"""
