"""Mirror of the reference's `demod_2400` module (src/demod_2400.rs)."""
from __future__ import annotations

from .context import MagnitudeBuffer, ModeSMessage, default_context  # noqa: F401


def demodulate2400(mag: MagnitudeBuffer):
    """src/demod_2400.rs:115-212 (always Ok upstream, so the list is returned bare)."""
    return default_context().demodulate2400(mag)
