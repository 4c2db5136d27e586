"""Overlay package: ``configs.models_config`` comes from here, every other ``configs.*`` module
(``gan_config``, ``data_config``, ``wae_config``, ``inference_config`` ... -- imported by the reference
scripts at e.g. train/train_vgan_stage1.py:21-22, train/train_wae_stage3.py:21-22,
inference/inference_gan.py:19-20) from the ``configs`` package of the project that follows this directory
on ``sys.path``.  Nothing of that project is copied: its directory is appended to this package's search path.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
