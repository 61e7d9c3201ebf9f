"""ctypes binding of oracle/liboracle.so (dump1090_oracle.h).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "liboracle.so"
MAG_DATA_LEN = 326 + 131072


class OrcMsg(C.Structure):
    _fields_ = [("msg", C.c_uint8 * 14), ("len", C.c_uint8), ("try_phase", C.c_uint8),
                ("score", C.c_int32), ("j", C.c_uint32), ("chunk", C.c_uint64),
                ("signal_level", C.c_double)]


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("preamble_pass", "snr_pass", "quiet_pass", "trials", "frames")]


class OrcMagBuf(C.Structure):
    _fields_ = [("data", C.c_uint16 * MAG_DATA_LEN), ("length", C.c_size_t),
                ("first_sample_timestamp_12mhz", C.c_size_t)]


class OrcFilter(C.Structure):
    _fields_ = [("a", C.c_uint32 * 4096), ("b", C.c_uint32 * 4096)]


def build(force: bool = False) -> Path:
    src = [HERE / "dump1090_oracle.c", HERE / "dump1090_oracle_mt.c", HERE / "dump1090_oracle.h", HERE / "Makefile"]
    if force or not LIB_PATH.exists() or any(p.stat().st_mtime > LIB_PATH.stat().st_mtime for p in src):
        subprocess.run(["make", "-C", str(HERE), "-B" if force else "-s", "liboracle.so"], check=True)
    return LIB_PATH


_lib = None


def build_native() -> Optional[Path]:
    """The same sources with -march=native, into a scratch directory: for TIMING the CPU baseline on
    the box it runs on (SURVEY 8d asks for -O3 -march=native; the tree's liboracle.so is x86-64-v3 so
    that one build runs on every box and stays the checker).  None when it cannot be built."""
    import tempfile
    out = Path(tempfile.gettempdir()) / f"liboracle_native_{__import__('os').getpid()}.so"
    cmd = ["gcc", "-O3", "-march=native", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fno-math-errno", "-fPIC",
           "-pthread", "-shared", "-o", str(out), str(HERE / "dump1090_oracle.c"), str(HERE / "dump1090_oracle_mt.c"), "-lm"]
    try:
        subprocess.run(cmd, check=True, capture_output=True, timeout=120)
    except (OSError, subprocess.SubprocessError):
        return None
    return out


def load(path: Path) -> C.CDLL:
    """A second instance of the oracle library (build_native), prototypes set like lib()'s."""
    return _bind(C.CDLL(str(path)))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        _lib = _bind(C.CDLL(str(LIB_PATH)))
    return _lib


def _bind(L: C.CDLL) -> C.CDLL:
    vp, sz = C.c_void_p, C.c_size_t
    L.orc_icao_flush.argtypes = [vp]
    L.orc_icao_hash.argtypes = [C.c_uint32]
    L.orc_icao_hash.restype = C.c_uint32
    L.orc_icao_filter_add.argtypes = [vp, C.c_uint32]
    L.orc_icao_filter_test.argtypes = [vp, C.c_uint32]
    L.orc_icao_filter_test.restype = C.c_int
    L.orc_crc_table_entry.argtypes = [C.c_uint]
    L.orc_crc_table_entry.restype = C.c_uint32
    L.orc_modes_checksum.argtypes = [vp, sz]
    L.orc_modes_checksum.restype = C.c_uint32
    L.orc_getbits.argtypes = [vp, sz, sz]
    L.orc_getbits.restype = sz
    L.orc_score_modes_message.argtypes = [vp, vp, sz, C.POINTER(C.c_int), C.POINTER(C.c_int32)]
    L.orc_score_modes_message.restype = C.c_int
    L.orc_to_mag.argtypes = [vp, sz, vp]
    L.orc_to_mag.restype = C.c_int
    L.orc_mag_sample.argtypes = [C.c_int16, C.c_int16]
    L.orc_mag_sample.restype = C.c_uint16
    L.orc_check_preamble.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.orc_check_preamble.restype = C.c_int
    L.orc_slice_phase.argtypes = [vp, sz, C.c_int, vp]
    L.orc_slice_phase.restype = None
    L.orc_demodulate2400.argtypes = [vp, vp, C.c_uint64, vp, sz, vp]
    L.orc_demodulate2400.restype = sz
    L.orc_demod_iq.argtypes = [vp, vp, sz, vp, sz, vp]
    L.orc_demod_iq.restype = sz
    L.orc_demod_iq_carry.argtypes = [vp, vp, sz, vp, sz, vp, vp]
    L.orc_demod_iq_carry.restype = sz
    L.orc_demod_iq_mt.argtypes = [vp, vp, sz, vp, sz, vp, C.c_int]
    L.orc_demod_iq_mt.restype = sz
    L.orc_read_test_data.argtypes = [C.c_char_p, vp, sz]
    L.orc_read_test_data.restype = C.c_long
    L.orc_mag_x_digest.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.orc_mag_x_digest.restype = C.c_uint64
    L.orc_all_trials.argtypes = [vp, C.c_uint64, vp, sz]
    L.orc_all_trials.restype = sz
    return L


def as_iq(iq) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(iq), dtype=np.int16)
    return a.reshape(-1, 2)


class Oracle:
    """One stream of the reference algorithm on the CPU (own ICAO filter)."""

    def __init__(self, L: Optional[C.CDLL] = None):
        self.L = L if L is not None else lib()
        self.filter = OrcFilter()

    def icao_flush(self) -> None:
        self.L.orc_icao_flush(C.byref(self.filter))

    def to_mag(self, iq) -> Tuple[np.ndarray, int]:
        a = as_iq(iq)
        mb = OrcMagBuf()
        if self.L.orc_to_mag(a.ctypes.data, a.shape[0], C.byref(mb)) != 0:
            raise IndexError("to_mag: more than 131072 samples")
        return np.ctypeslib.as_array(mb.data).copy(), int(mb.length)

    def demodulate2400(self, data: np.ndarray, length: int, cap: int = 65536):
        mb = OrcMagBuf()
        C.memmove(mb.data, np.ascontiguousarray(data, dtype=np.uint16).ctypes.data, MAG_DATA_LEN * 2)
        mb.length = length
        out = (OrcMsg * cap)()
        st = OrcStats()
        n = self.L.orc_demodulate2400(C.byref(self.filter), C.byref(mb), 0, out, cap, C.byref(st))
        assert n <= cap
        return [unpack(m) for m in out[:n]], st

    def demod_iq(self, iq, cap: Optional[int] = None, threads: int = 1, timing: Optional[list] = None):
        """threads > 1: workers per buffer + ordered replay (dump1090_oracle_mt.c), same result.
        `timing`: a list that gets the seconds spent inside the C call appended (the output array --
        40 MB for a million messages -- is allocated and zeroed outside of that)."""
        import time
        a = as_iq(iq)
        cap = cap or max(4096, a.shape[0] // 64)
        if getattr(self, "_out_cap", 0) != cap:
            self._out, self._out_cap = (OrcMsg * cap)(), cap
        out = self._out
        st = OrcStats()
        t0 = time.perf_counter()
        if threads > 1:
            n = self.L.orc_demod_iq_mt(C.byref(self.filter), a.ctypes.data, a.shape[0], out, cap,
                                       C.byref(st), threads)
        else:
            n = self.L.orc_demod_iq(C.byref(self.filter), a.ctypes.data, a.shape[0], out, cap, C.byref(st))
        if timing is not None:
            timing.append(time.perf_counter() - t0)
        assert n <= cap, "oracle output overflowed its buffer"
        return [unpack(m) for m in out[:n]], st


def demod_iq_carry(orc: "Oracle", iq, carry: np.ndarray, cap: Optional[int] = None):
    """The carry-over extension (NOT reference behaviour, see dump1090_oracle.h): `carry` is a
    (326, 2) int16 array holding the stream state, updated in place."""
    a = as_iq(iq)
    assert carry.dtype == np.int16 and carry.shape == (326, 2) and carry.flags.c_contiguous
    cap = cap or max(4096, a.shape[0] // 64)
    out = (OrcMsg * cap)()
    st = OrcStats()
    n = orc.L.orc_demod_iq_carry(C.byref(orc.filter), a.ctypes.data, a.shape[0], out, cap, C.byref(st),
                                 carry.ctypes.data)
    assert n <= cap
    return [unpack(m) for m in out[:n]], st


TRIAL_DTYPE = np.dtype([("power", "<u8"), ("chunk", "<u4"), ("j_tp", "<u4"), ("msg", "u1", (14,)), ("pad", "<u2")])
AP_DFS = frozenset([0, 4, 5, 16, 20, 21] + list(range(24, 32)))   # src/mode_s/mod.rs:56-72,110-135


def all_trials(iq_chunk, chunk: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """(magnitudes, the 5 trial messages of every gate-passing j in (j, try_phase) order) of one buffer."""
    L = lib()
    a = as_iq(iq_chunk)
    mb = OrcMagBuf()
    if L.orc_to_mag(a.ctypes.data, a.shape[0], C.byref(mb)) != 0:
        raise IndexError("more than 131072 samples")
    cap = 5 * 131072
    buf = np.zeros(cap, dtype=TRIAL_DTYPE)
    k = L.orc_all_trials(C.byref(mb), chunk, buf.ctypes.data, cap)
    return np.ctypeslib.as_array(mb.data).copy(), buf[:k].copy()


def stage_lists(iq) -> dict:
    """Stage-level values of the reference algorithm over an IQ stream, buffer by buffer -- what
    tests/golden/stage_goldens.json freezes and adsb_selftest_stage_lists is compared with:
      mags        list of u16 arrays (one per buffer)
      preamble    positions where check_preamble returns Some      (buffer << 32 | j)
      snr         ... that also pass the 3.5 dB gate (demod_2400.rs:129)
      cand        ... and the quiet gate (:135-146): the positions that get sliced
      trials      (buffer, j, try_phase, DF, residual over the message's own length) of all 5 trials of each
      ap          the address/parity ones as buffer << 45 | j << 28 | try_phase << 24 | residual"""
    L = lib()
    a = as_iq(iq)
    out = {"mags": [], "preamble": [], "snr": [], "cand": [], "trials": [], "ap": []}
    hi, sig, noise = C.c_int32(), C.c_uint32(), C.c_uint32()
    for chunk, off in enumerate(range(0, a.shape[0], 131072)):
        part = np.ascontiguousarray(a[off:off + 131072])
        data, tr = all_trials(part, chunk)
        out["mags"].append(data)
        d = np.ascontiguousarray(data, dtype=np.uint16)
        base = d.ctypes.data
        # the quick test of check_preamble (demod_2400.rs:221) first, vectorised: only those j can match
        n = part.shape[0]
        quick = np.nonzero((d[0:n] < d[1:n + 1]) & (d[12:n + 12] > d[13:n + 13]))[0]
        for j in quick:
            if L.orc_check_preamble(base + 2 * int(j), C.byref(hi), C.byref(sig), C.byref(noise)):
                out["preamble"].append(chunk << 32 | int(j))
                if 2 * sig.value >= 3 * noise.value:
                    out["snr"].append(chunk << 32 | int(j))
        js = tr["j_tp"] & 0xFFFFFF
        out["cand"].extend((chunk << 32 | int(j)) for j in np.unique(js))
        for r in tr:
            msg = bytes(r["msg"])
            df = msg[0] >> 3
            bits = 112 if df & 0x10 else 56
            res = L.orc_modes_checksum(msg, bits)
            j, tp = int(r["j_tp"]) & 0xFFFFFF, int(r["j_tp"]) >> 24
            out["trials"].append((chunk, j, tp, df, res))
            if df in AP_DFS:
                out["ap"].append(chunk << 45 | j << 28 | tp << 24 | res)
    out["ap"].sort()
    return out


def unpack(m: OrcMsg) -> dict:
    return {"msg": bytes(m.msg), "len": int(m.len), "try_phase": int(m.try_phase), "score": int(m.score),
            "j": int(m.j), "chunk": int(m.chunk), "signal_level": float(m.signal_level),
            "buffer": bytes(m.msg[: m.len])}


def read_test_data(path: str) -> np.ndarray:
    buf = np.zeros((0x20000, 2), dtype=np.int16)
    n = lib().orc_read_test_data(str(path).encode(), buf.ctypes.data, 0x20000)
    if n < 0:
        raise IOError(path)
    return buf[:n]
