"""Mirror of the reference's `icao_filter` module's public entry (src/icao_filter.rs:11)."""
from __future__ import annotations

from .context import default_context


def icao_flush() -> None:
    """src/icao_filter.rs:11-17, for the process-wide default context."""
    default_context().icao_flush()
