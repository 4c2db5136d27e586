"""Overlay package: ``train.train_utils`` is the engine's (PearsonCorrelation / StructuralSimilarity on the
MI355X, everything else forwarded to the shadowed module -- see train_utils.__getattr__); other ``train.*``
modules are found in the ``train`` package of the project that follows this directory on ``sys.path``.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
