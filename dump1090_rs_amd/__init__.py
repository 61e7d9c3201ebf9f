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
from . import utils  # noqa: F401


def _crate_module(name: str, doc: str, **items):
    """The reference's `demod_2400` and `icao_filter` modules have one public function each on this path
    (src/demod_2400.rs:115, src/icao_filter.rs:11): the same names here, as modules of this package, without a
    file apiece."""
    import sys
    import types
    m = types.ModuleType(f"{__name__}.{name}", doc)
    m.__dict__.update(items)
    sys.modules[m.__name__] = m
    return m


def _demodulate2400(mag: MagnitudeBuffer):
    """src/demod_2400.rs:115-212 (always Ok upstream, so the list is returned bare)."""
    return default_context().demodulate2400(mag)


def _icao_flush() -> None:
    """src/icao_filter.rs:11-17, for the process-wide default context."""
    default_context().icao_flush()


_demodulate2400.__name__ = _demodulate2400.__qualname__ = "demodulate2400"
_icao_flush.__name__ = _icao_flush.__qualname__ = "icao_flush"
demod_2400 = _crate_module("demod_2400", "Mirror of the reference's `demod_2400` module (src/demod_2400.rs).",
                           demodulate2400=_demodulate2400, ModeSMessage=ModeSMessage, MagnitudeBuffer=MagnitudeBuffer)
icao_filter = _crate_module("icao_filter", "Mirror of the reference's `icao_filter` module's public entry (src/icao_filter.rs:11).",
                            icao_flush=_icao_flush)

__all__ = [
    "MODES_MAG_BUF_SAMPLES", "MODES_LONG_MSG_BYTES", "MODES_SHORT_MSG_BYTES", "TRAILING_SAMPLES",
    "MagnitudeBuffer", "ModeSMessage", "Context", "default_context",
    "utils", "demod_2400", "icao_filter",
]
