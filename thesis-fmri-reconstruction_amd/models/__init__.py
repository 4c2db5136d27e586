"""Overlay package: ``models.vae_gan`` is the engine's drop-in module (reference models/vae_gan.py); any other
``models.*`` module is looked up in the ``models`` directory of the project that follows on ``sys.path``
(the reference's ``models/`` has no ``__init__.py``, so its directory is appended by hand)."""
import os
import sys
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
for _p in sys.path:
    _d = os.path.join(_p or os.curdir, "models")
    if os.path.isdir(_d) and os.path.abspath(_d) not in [os.path.abspath(q) for q in __path__]:
        __path__.append(_d)
