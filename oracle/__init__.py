"""CPU oracle for the HealthiVert-GAN hot path (TEST INFRASTRUCTURE ONLY).

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / timed CPU baseline.  The product path
(``healthivert-gan_amd``) never imports this package and fails loudly when the
HIP library is missing.
"""
