"""ORACLE -- test infrastructure only.

CPU fp32 restatement of the reference's per-batch forward/backward path, used as the checker by
tests/, bench.py's `cpu_baseline` leg and __graft_entry__.smoke().  The product package
(multimodal-diagnosis-ham-spine_amd/) never imports anything from here.
"""
