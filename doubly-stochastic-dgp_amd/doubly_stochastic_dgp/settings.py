"""[UPSTREAM] gpflow.settings subset used on the path: float_type is float64 (layers.py:68), numerics.jitter_level 1e-6."""
import contextlib

import numpy as np

float_type = np.float64
jitter = 1e-6


@contextlib.contextmanager
def temp_jitter(value):
    """analogue of `settings.temp_settings(custom_config)` with `numerics.jitter_level = value` (tests/test_dgp.py:7-11)"""
    global jitter
    old = jitter
    jitter = float(value)
    try:
        yield
    finally:
        jitter = old
