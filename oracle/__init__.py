"""CPU oracle for the transfer_em CycleGAN hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (transfer_em_amd/) never does.  PARITY UNPINNED: see README.md.
"""
