"""Binds the drop-in `models` package to the kernel package whether it is imported as
`healthivert-gan_amd.models` (in-tree) or as top-level `models` (PYTHONPATH drop-in)."""
import importlib
import os
import sys

_pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_root = os.path.dirname(_pkg_dir)
if _root not in sys.path:
    sys.path.insert(0, _root)
hv = importlib.import_module(os.path.basename(_pkg_dir))
engine, ops, lib, ddp, optim = hv.engine, hv.ops, hv.lib, hv.ddp, hv.optim
