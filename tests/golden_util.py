import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def unhex(lst, shape=None):
    a = np.array([float.fromhex(s) for s in lst], dtype=np.float64)
    return a if shape is None else a.reshape(shape)


def bits_equal(a, b):
    """Bit-for-bit equality of two float64 arrays (NaN payloads and signed zeros included)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def max_ulp(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).view(np.int64).astype(np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64).view(np.int64).astype(np.float64)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def max_rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = np.maximum(np.abs(b), 1.0)   # poses: positions O(1..1e3), unit quaternions
    return float(np.max(np.abs(a - b) / scale)) if a.size else 0.0
