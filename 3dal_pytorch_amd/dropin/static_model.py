"""Top-level `static_model` module for the reference's drivers: put this directory on sys.path ahead of
the reference's tools/ and `from static_model import ...` resolves to the MI355X implementation."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _root not in sys.path:
    sys.path.insert(0, _root)
_impl = importlib.import_module("3dal_pytorch_amd.static_model")
globals().update({k: getattr(_impl, k) for k in dir(_impl) if not k.startswith("__")})
