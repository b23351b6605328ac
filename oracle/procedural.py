"""ORACLE -- test infrastructure only.  Seeded procedural weights and synthetic batches.

The generators themselves live in the product package (hamspine/synthetic.py: they only make INPUTS -- weights and
batches -- and compute nothing of the path), so that bench.py's product leg never imports oracle/.  The oracle re-exports
them here so both sides of every parity check are fed by the same function.
"""
import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multimodal-diagnosis-ham-spine_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from hamspine.synthetic import (load_procedural, procedural_state_dict, procedural_tensor,  # noqa: E402,F401
                                synthetic_batch)
