"""Import alias: ``import hvgan`` == the package in ``healthivert-gan_amd/`` (hyphenated dir name)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("healthivert-gan_amd")
