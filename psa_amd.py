"""Importable alias of the `parallel-sequence-alignment_amd` package (its name has a hyphen)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
_pkg = importlib.import_module("parallel-sequence-alignment_amd")
sys.modules[__name__] = _pkg
