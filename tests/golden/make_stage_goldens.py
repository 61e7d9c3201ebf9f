"""Freeze stage-level values of the CPU oracle over the three reference captures into
stage_goldens.json (run from the repo root: python tests/golden/make_stage_goldens.py).

The oracle is pinned against the reference's own known-answer vectors first
(tests/test_oracle_golden.py: reference tests/test.rs:19-59); these values are then provenance
from the oracle, there so that a failing stage -- magnitudes, preamble match, 3.5 dB gate, quiet
gate, slicer + CRC -- can be located, on the GPU (tests/test_gpu_parity.py compares the device's
magnitudes, its candidate list and its address/parity trial list with them) and in the oracle
itself (tests/test_oracle_golden.py regenerates and compares)."""
import hashlib
import json
import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import binding  # noqa: E402

GOLDEN = ROOT / "tests" / "golden"


def digest_u64(values) -> str:
    return hashlib.sha256(np.asarray(values, dtype="<u8").tobytes()).hexdigest()


def stage_values(iq) -> dict:
    st = binding.stage_lists(iq)
    trials = np.array([(c << 45 | j << 28 | tp << 24 | res) for c, j, tp, df, res in st["trials"]], dtype=np.uint64)
    dfs = np.bincount([df for *_, df, _ in st["trials"]], minlength=32)
    return {
        "mag_crc32": [zlib.crc32(m.astype("<u2").tobytes()) for m in st["mags"]],
        "mag_max": [int(m.max()) for m in st["mags"]],
        "n_preamble": len(st["preamble"]), "preamble_sha256": digest_u64(st["preamble"]),
        "n_snr": len(st["snr"]), "snr_sha256": digest_u64(st["snr"]),
        "n_cand": len(st["cand"]), "cand_sha256": digest_u64(st["cand"]),
        "n_trials": len(st["trials"]), "trial_residual_sha256": digest_u64(trials),
        "trial_df_histogram": [int(x) for x in dfs],
        "n_ap": len(st["ap"]), "ap_sha256": digest_u64(st["ap"]),
    }


def main():
    ref = json.loads((GOLDEN / "reference_frames.json").read_text())
    out = {"comment": "stage-level values of oracle/ over the reference captures; see make_stage_goldens.py",
           "encoding": {"positions": "buffer << 32 | j, ascending, little-endian u64, sha256",
                        "trials": "buffer << 45 | j << 28 | try_phase << 24 | CRC residual over the message's own length"},
           "fixtures": {}}
    for fx in ref["fixtures"]:
        raw = np.fromfile(GOLDEN / fx["file"], dtype="<i2").reshape(-1, 2)
        out["fixtures"][fx["file"]] = stage_values(np.ascontiguousarray(raw[:, ::-1]))
    (GOLDEN / "stage_goldens.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps({k: {q: v[q] for q in ("n_preamble", "n_snr", "n_cand", "n_trials", "n_ap")}
                      for k, v in out["fixtures"].items()}, indent=1))


if __name__ == "__main__":
    main()
