"""ORACLE — test infrastructure only.

CPU restatement of the reference's algorithm for the TD-VC-GAN G+D train-step path, used as
the checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg. The product
package never imports this. See oracle/model.py for parity status and reference anchors.
"""
