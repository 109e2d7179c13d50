"""Gives tests access to helpers of tests/golden/make_golden.py (generator-side MT19937-64)."""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location(
    "make_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
MT19937_64 = _mod.MT19937_64
synth_protein = _mod.synth_protein
