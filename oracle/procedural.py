"""ORACLE -- test infrastructure only.  Seeded procedural weights and synthetic batches.

The generators themselves live in the product package (hamspine/synthetic.py: they only make INPUTS -- weights and
batches -- and compute nothing of the path), so that bench.py's product leg never imports oracle/.  The oracle re-exports
them here so both sides of every parity check are fed by the same function.
"""
import importlib.util
import os

# loaded by file path, NOT through sys.path: oracle/gen_golden.py imports the REFERENCE's modules, whose top-level names
# (model, encoder, modules, mibf_net, ConNexT) the product package shadows once its directory is on sys.path
_SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multimodal-diagnosis-ham-spine_amd", "hamspine",
                    "synthetic.py")
_spec = importlib.util.spec_from_file_location("_hamspine_synthetic", _SRC)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
load_procedural, procedural_state_dict = _mod.load_procedural, _mod.procedural_state_dict
procedural_tensor, synthetic_batch = _mod.procedural_tensor, _mod.synthetic_batch
