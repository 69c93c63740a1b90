"""MI355X-native hot path of HealthiVert-GAN (generator / discriminator train + inference loop).

The directory name contains a hyphen, so import it with
``importlib.import_module("healthivert-gan_amd")`` or through the ``hvgan`` alias module at the
repo root.  The HIP library is loaded lazily (``lib.get()``) and every operator raises if it is
missing -- there is no CPU fallback in the product path.
"""
__version__ = "0.1.0"
from . import synth  # noqa: E402,F401  (numpy only)
from . import lib  # noqa: E402,F401  (ctypes binding; loads libhvgan.so lazily)
from . import ops  # noqa: E402,F401
from . import engine, optim, ddp, profiler, evaluation, eval_metrics  # noqa: E402,F401
from . import models  # noqa: E402,F401  (drop-in mirror of the reference's `models` package)
