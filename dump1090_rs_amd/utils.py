"""Mirror of the reference's `utils` module (src/utils.rs)."""
from __future__ import annotations

import time

import numpy as np

from .context import MagnitudeBuffer, default_context


def read_test_data(filepath: str) -> np.ndarray:
    """src/utils.rs:23-40: 0x20000 samples; file pairs are [im][re] little-endian
    i16; returns (0x20000, 2) int16 rows in memory order [re, im]."""
    raw = np.fromfile(filepath, dtype="<i2", count=2 * 0x20000)
    if raw.size != 2 * 0x20000:
        raise IOError(f"{filepath}: expected {4 * 0x20000} bytes")  # the reference unwrap()s
    return np.ascontiguousarray(raw.reshape(-1, 2)[:, ::-1])


def save_test_data(data, directory: str = ".") -> str:
    """src/utils.rs:8-20: write [im][re] pairs to test_<unix ms>.iq."""
    a = np.asarray(data, dtype=np.int16).reshape(-1, 2)
    name = f"{directory}/test_{int(time.time() * 1000)}.iq"
    np.ascontiguousarray(a[:, ::-1]).astype("<i2").tofile(name)
    return name


def to_mag(data) -> MagnitudeBuffer:
    """src/utils.rs:43-58, computed by the HIP library on the default context."""
    return default_context().to_mag(data)
