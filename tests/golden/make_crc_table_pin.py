"""Put all 256 entries of the reference's CRC_TABLE (rsadsb/dump1090_rs src/crc.rs:3-260) into
reference_frames.json as data ("crc_table": 256 six-digit hex strings), next to the ten spot pins.

    python tests/golden/make_crc_table_pin.py [/root/reference]

Only the numeric constants are taken (a known-answer vector the reference holds for this path); run
in the build container, where the reference checkout exists -- the GPU box only sees the JSON."""
import json
import re
import sys
from pathlib import Path

GOLDEN = Path(__file__).resolve().parent


def main():
    ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    text = (ref / "src" / "crc.rs").read_text()
    body = text[text.index("pub const CRC_TABLE: [u32; 256] = ["):]
    body = body[: body.index("];")]
    vals = [int(v.replace("_", ""), 16) for v in re.findall(r"0x([0-9a-fA-F_]+)", body)]
    assert len(vals) == 256 and vals[0] == 0 and vals[1] == 0xFFF409 and vals[255] == 0xFA0480
    f = GOLDEN / "reference_frames.json"
    d = json.loads(f.read_text())
    d["crc_table"] = [f"{v:06x}" for v in vals]
    d["crc_table_source"] = "rsadsb/dump1090_rs v0.8.1 src/crc.rs:3-260 (CRC_TABLE, all 256 entries; make_crc_table_pin.py)"
    f.write_text(json.dumps(d, indent=1) + "\n")
    print("pinned", len(vals), "entries")


if __name__ == "__main__":
    main()
