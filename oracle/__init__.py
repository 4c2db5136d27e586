"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference hot path.

Nothing under ``oracle/`` is part of the shipped engine.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it,
and only as the checker (never as the thing measured as the product, never as a fallback).
"""
