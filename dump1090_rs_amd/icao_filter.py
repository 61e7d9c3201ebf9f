"""Mirror of the reference's `icao_filter` module's public entry (src/icao_filter.rs:11)."""
from .context import default_context


def icao_flush() -> None:
    default_context().icao_flush()
