"""dump1090_rs_amd -- MI355X (gfx950) drop-in for the demod_2400 hot path of
rsadsb/dump1090_rs.

The compute lives in libadsb_hip.so (hand-written HIP kernels + C ABI,
include/adsb_hip.h).  This package is the thin host-side mirror of the reference's
library surface for the path -- same module and item names as the Rust crate
(src/lib.rs:12-26): `utils.to_mag`, `utils.read_test_data`,
`demod_2400.demodulate2400`, `demod_2400.ModeSMessage.buffer`,
`icao_filter.icao_flush`, `MagnitudeBuffer`, the MODES_* constants -- plus
`Context` for explicit per-GPU streams and device-resident IQ.  There is no CPU
fallback anywhere in it.
"""
from .context import (  # noqa: F401
    MODES_LONG_MSG_BYTES,
    MODES_MAG_BUF_SAMPLES,
    MODES_SHORT_MSG_BYTES,
    TRAILING_SAMPLES,
    Context,
    MagnitudeBuffer,
    ModeSMessage,
    default_context,
)
from . import demod_2400, icao_filter, utils  # noqa: F401

__all__ = [
    "MODES_MAG_BUF_SAMPLES", "MODES_LONG_MSG_BYTES", "MODES_SHORT_MSG_BYTES", "TRAILING_SAMPLES",
    "MagnitudeBuffer", "ModeSMessage", "Context", "default_context",
    "utils", "demod_2400", "icao_filter",
]
