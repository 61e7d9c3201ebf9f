"""Context: one GPU stream of the demod_2400 path (wraps adsb_ctx of include/adsb_hip.h)."""
from __future__ import annotations

import ctypes as C
import struct
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib
from ._lib import AdsbError, AdsbMsg, AdsbStats

# reference src/lib.rs:22-26
MODES_MAG_BUF_SAMPLES = 131_072
TRAILING_SAMPLES = 326
MODES_LONG_MSG_BYTES = 14
MODES_SHORT_MSG_BYTES = 7
MAG_DATA_LEN = TRAILING_SAMPLES + MODES_MAG_BUF_SAMPLES


@dataclass
class MagnitudeBuffer:
    """reference src/lib.rs:30-51: 326 zero lead-in samples, then `length` magnitudes."""
    data: np.ndarray = field(default_factory=lambda: np.zeros(MAG_DATA_LEN, dtype=np.uint16))
    length: int = 0
    first_sample_timestamp_12mhz: int = 0

    def push(self, x: int) -> None:  # src/lib.rs:47-50
        if self.length >= MODES_MAG_BUF_SAMPLES:
            raise IndexError("MagnitudeBuffer is full")  # the reference panics (index OOB)
        self.data[TRAILING_SAMPLES + self.length] = x
        self.length += 1


@dataclass(frozen=True)
class ModeSMessage:
    """reference src/demod_2400.rs:92-112.  Only buffer() is public upstream; the
    other fields are what its Debug output shows, plus (chunk, j, try_phase)."""
    msg: bytes          # all 14 sliced bytes
    msglen: int         # 7 (MsgLen::Short) | 14 (MsgLen::Long)
    signal_level: float
    score: int
    j: int = 0
    try_phase: int = 0
    chunk: int = 0

    def buffer(self) -> bytes:  # src/demod_2400.rs:106-111
        return self.msg[: self.msglen]


_MSG_STRUCT = struct.Struct("<14sBBiIQd")   # adsb_msg (include/adsb_hip.h): msg, len, try_phase, score, j, chunk, signal_level
assert _MSG_STRUCT.size == C.sizeof(AdsbMsg)


def _as_iq(iq) -> np.ndarray:
    """Accept (N,2) int16 [re, im] rows, flat interleaved int16, or complex arrays of ints."""
    if type(iq) is np.ndarray and iq.dtype == np.int16 and iq.ndim == 2 and iq.shape[1] == 2 and iq.flags.c_contiguous:
        return iq
    a = np.asarray(iq)
    if np.iscomplexobj(a):
        a = np.stack([a.real, a.imag], axis=-1)
    a = np.ascontiguousarray(a, dtype=np.int16)
    if a.ndim == 1:
        if a.size % 2:
            raise ValueError("interleaved IQ needs an even number of int16")
        a = a.reshape(-1, 2)
    if a.ndim != 2 or a.shape[1] != 2:
        raise ValueError("IQ must be (N, 2) int16 rows of [re, im]")
    return a


# adsb_trial as a numpy record (32 bytes)
TRIAL_DTYPE = np.dtype([("power", "<u8"), ("chunk", "<u4"), ("j_tp", "<u4"), ("msg", "u1", (14,)), ("pad", "<u2")])


def replay_records(records: np.ndarray, filter_table: Optional[np.ndarray] = None, cap: Optional[int] = None
                   ) -> List["ModeSMessage"]:
    """adsb_replay_records: the ordered host replay (scoring + best-of-5 + ICAO filter) over raw
    trial records; `filter_table` (4096 u32, table A of the filter) is read and updated."""
    L = _lib.lib()
    rec = np.ascontiguousarray(records, dtype=TRIAL_DTYPE).copy()
    table = filter_table if filter_table is not None else np.zeros(4096, dtype=np.uint32)
    assert table.dtype == np.uint32 and table.shape == (4096,) and table.flags.c_contiguous
    cap = cap or max(4096, rec.shape[0])
    out = (AdsbMsg * cap)()
    n = C.c_size_t()
    st = L.adsb_replay_records(table.ctypes.data, rec.ctypes.data, rec.shape[0], out, cap, C.byref(n))
    if st != _lib.ADSB_OK:
        raise AdsbError(st, f"adsb_replay_records: {L.adsb_strerror(st).decode()}")
    return [ModeSMessage(bytes(m.msg), int(m.len), float(m.signal_level), int(m.score), int(m.j),
                         int(m.try_phase), int(m.chunk)) for m in out[: n.value]]


class Context:
    """One adsb_ctx: device buffers, stream and the ICAO filter of one stream of IQ."""

    def __init__(self, device: int = 0, max_chunks: int = 1):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        st = self._L.adsb_create(C.byref(self._h), int(device), int(max_chunks))
        if st != _lib.ADSB_OK:
            self._h = C.c_void_p()
            raise AdsbError(st, "adsb_create", self._L.adsb_strerror(st).decode())
        self.device = device
        self.max_chunks = max_chunks

    # -- lifetime
    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._L.adsb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, st: int, what: str) -> None:
        if st != _lib.ADSB_OK:
            detail = self._L.adsb_last_error(self._h).decode() if st == _lib.ADSB_ERR_HIP else ""
            raise AdsbError(st, f"{what}: {self._L.adsb_strerror(st).decode()}", detail)

    # -- reference API
    def icao_flush(self) -> None:
        self._check(self._L.adsb_icao_flush(self._h), "adsb_icao_flush")

    def to_mag(self, iq) -> MagnitudeBuffer:
        a = _as_iq(iq)
        # (the library writes all 131398 entries, lead-in and tail zeros included: no need to zero them first)
        out = MagnitudeBuffer(data=np.empty(MAG_DATA_LEN, dtype=np.uint16))
        n = C.c_size_t()
        st = self._L.adsb_to_mag(self._h, a.__array_interface__["data"][0], a.shape[0],
                                 out.data.__array_interface__["data"][0], C.byref(n))
        if st == _lib.ADSB_ERR_TOO_LONG:
            raise IndexError("to_mag: more than 131072 samples (the reference panics here)")
        self._check(st, "adsb_to_mag")
        out.length = n.value
        return out

    def _collect(self, call, what: str, cap: int) -> List[ModeSMessage]:
        # (the output array is kept between calls: allocating and zeroing 4096 entries costs more than
        # a one-buffer pass takes)
        if getattr(self, "_out_cap", 0) != cap:
            self._out_buf, self._out_cap = (AdsbMsg * cap)(), cap
            self._out_view = memoryview(self._out_buf).cast("B")
        buf, view = self._out_buf, self._out_view
        n = C.c_size_t()
        st = call(buf, cap, C.byref(n))
        if st == _lib.ADSB_ERR_CAPACITY:
            # the pass is consumed (the filter has advanced): never repeat the call, fetch its list
            cap = n.value
            buf = (AdsbMsg * cap)()
            view = memoryview(buf).cast("B")
            st = self._L.adsb_fetch_messages(self._h, buf, cap, C.byref(n))
        self._check(st, what)
        # (adsb_msg unpacked in one go: field by field through ctypes a list of five messages cost 11 us, more
        # than a quarter of the call that produced it)
        return [ModeSMessage(m, ln, sig, score, j, tp, chunk)
                for (m, ln, tp, score, j, chunk, sig) in _MSG_STRUCT.iter_unpack(view[: _MSG_STRUCT.size * n.value])]

    def demodulate2400(self, mag: MagnitudeBuffer, cap: int = 4096) -> List[ModeSMessage]:
        data = np.ascontiguousarray(mag.data, dtype=np.uint16)
        if data.shape != (MAG_DATA_LEN,):
            raise ValueError("MagnitudeBuffer.data must hold 131398 u16")
        if mag.length > MODES_MAG_BUF_SAMPLES:
            raise IndexError("MagnitudeBuffer.length > 131072")
        return self._collect(
            lambda out, c, n: self._L.adsb_demodulate2400(self._h, data.__array_interface__["data"][0], mag.length, out, c, n),
            "adsb_demodulate2400", cap)

    # -- stream forms (to_mag + demodulate2400 per 131072-sample buffer)
    def demod_iq(self, iq, cap: Optional[int] = None) -> List[ModeSMessage]:
        a = _as_iq(iq)
        cap = cap or max(4096, a.shape[0] // 256)
        ptr = a.__array_interface__["data"][0]   # (a.ctypes.data builds an object per call: 1.8 us against 1.1)
        return self._collect(
            lambda out, c, n: self._L.adsb_demod_iq(self._h, ptr, a.shape[0], out, c, n),
            "adsb_demod_iq", cap)

    def demod_iq_device(self, device_ptr: int, n_samples: int, cap: Optional[int] = None
                        ) -> List[ModeSMessage]:
        cap = cap or max(4096, n_samples // 256)
        return self._collect(
            lambda out, c, n: self._L.adsb_demod_iq_device(self._h, C.c_void_p(device_ptr), n_samples, out, c, n),
            "adsb_demod_iq_device", cap)

    def demod_iq_device_raw(self, device_ptr: int, n_samples: int, out_buf, cap: int) -> int:
        """No Python-side unpacking: for the bench loop.  Returns the message count."""
        n = C.c_size_t()
        st = self._L.adsb_demod_iq_device(self._h, C.c_void_p(device_ptr), n_samples, out_buf, cap, C.byref(n))
        self._check(st, "adsb_demod_iq_device")
        return n.value

    # -- pipelined form: keep up to four passes in flight (ADSB_MAX_IN_FLIGHT), results in submission order
    def submit_iq_device(self, device_ptr: int, n_samples: int) -> None:
        self._check(self._L.adsb_submit_iq_device(self._h, C.c_void_p(device_ptr), n_samples),
                    "adsb_submit_iq_device")

    def collect_raw(self, out_buf, cap: int) -> int:
        n = C.c_size_t()
        self._check(self._L.adsb_collect(self._h, out_buf, cap, C.byref(n)), "adsb_collect")
        return n.value

    def collect(self, cap: int = 1 << 16) -> List[ModeSMessage]:
        return self._collect(lambda out, c, n: self._L.adsb_collect(self._h, out, c, n), "adsb_collect", cap)

    def selftest_stage_lists(self, device_ptr: int, n_samples: int):
        """adsb_selftest_stage_lists: (candidate positions as buffer << 32 | j, address/parity trials as
        buffer << 45 | j << 28 | try_phase << 24 | residual), both ascending u64 arrays."""
        cc, ac = max(4096, n_samples // 16), max(4096, n_samples // 8)
        while True:
            cand, ap = np.zeros(cc, dtype=np.uint64), np.zeros(ac, dtype=np.uint64)
            nc, na = C.c_size_t(), C.c_size_t()
            st = self._L.adsb_selftest_stage_lists(self._h, C.c_void_p(device_ptr), n_samples, cand.ctypes.data, cc,
                                                   C.byref(nc), ap.ctypes.data, ac, C.byref(na))
            if st == _lib.ADSB_ERR_CAPACITY:
                cc, ac = max(cc, nc.value), max(ac, na.value)
                continue
            self._check(st, "adsb_selftest_stage_lists")
            return cand[: nc.value].copy(), ap[: na.value].copy()

    def selftest_gate_stages(self, device_ptr: int, n_samples: int):
        """adsb_selftest_gate_stages: (positions where check_preamble returns Some, those that also pass
        the 3.5 dB test), both as buffer << 32 | j, ascending u64 arrays."""
        pc, sc = max(4096, n_samples // 8), max(4096, n_samples // 16)
        while True:
            pre, snr = np.zeros(pc, dtype=np.uint64), np.zeros(sc, dtype=np.uint64)
            npre, nsnr = C.c_size_t(), C.c_size_t()
            st = self._L.adsb_selftest_gate_stages(self._h, C.c_void_p(device_ptr), n_samples, pre.ctypes.data, pc,
                                                   C.byref(npre), snr.ctypes.data, sc, C.byref(nsnr))
            if st == _lib.ADSB_ERR_CAPACITY:
                pc, sc = max(pc, npre.value), max(sc, nsnr.value)
                continue
            self._check(st, "adsb_selftest_gate_stages")
            return pre[: npre.value].copy(), snr[: nsnr.value].copy()

    # -- sharded capture: two phases around a host-side exchange of learned addresses
    def shard_scan(self, device_ptr: int, n_samples: int) -> np.ndarray:
        """Phase 1 on this shard: the addresses its clean DF11 / DF17 frames will add (sorted u32)."""
        cap = 1 << 16
        while True:
            out = np.zeros(cap, dtype=np.uint32)
            n = C.c_size_t()
            st = self._L.adsb_shard_scan(self._h, C.c_void_p(device_ptr), n_samples, out.ctypes.data, cap,
                                         C.byref(n))
            if st == _lib.ADSB_ERR_CAPACITY:  # the shard stays parked: finish it, then retry bigger
                self._L.adsb_shard_finish(self._h, None, 0, None, 0, None)
                cap = n.value
                continue
            self._check(st, "adsb_shard_scan")
            return out[: n.value].copy()

    def shard_finish(self, extra_addrs, cap: Optional[int] = None) -> np.ndarray:
        """Phase 2: add the other shards' addresses, match, return the raw trial records
        (structured array with the layout of adsb_trial; `chunk` is local to the shard)."""
        extra = np.ascontiguousarray(extra_addrs, dtype=np.uint32)
        cap = cap or (1 << 20)
        rec = np.zeros(cap, dtype=TRIAL_DTYPE)
        n = C.c_size_t()
        st = self._L.adsb_shard_finish(self._h, extra.ctypes.data if extra.size else None, extra.size,
                                       rec.ctypes.data, cap, C.byref(n))
        if st == _lib.ADSB_ERR_CAPACITY:
            raise AdsbError(st, f"adsb_shard_finish: {n.value} records exceed cap {cap}")
        self._check(st, "adsb_shard_finish")
        return rec[: n.value].copy()

    # -- a buffer the caller keeps (main.rs:154-167 reads the SDR into one Vec for ever): pinned and mapped once,
    #    demod_iq on samples inside it reads them in place
    def host_register(self, a: np.ndarray) -> None:
        self._check(self._L.adsb_host_register(self._h, a.ctypes.data, a.nbytes), "adsb_host_register")

    def host_unregister(self, a: np.ndarray) -> None:
        self._check(self._L.adsb_host_unregister(self._h, a.ctypes.data), "adsb_host_unregister")

    # -- streaming ring: pinned host buffers, H2D overlapped with the other slot's pass
    def ring_create(self, samples_per_slot: int) -> None:
        self._check(self._L.adsb_ring_create(self._h, samples_per_slot), "adsb_ring_create")

    def ring_acquire(self) -> np.ndarray:
        """The pinned (capacity, 2) int16 [re, im] buffer of the next submission."""
        ptr, cap = C.c_void_p(), C.c_size_t()
        self._check(self._L.adsb_ring_acquire(self._h, C.byref(ptr), C.byref(cap)), "adsb_ring_acquire")
        buf = (C.c_int16 * (2 * cap.value)).from_address(ptr.value)
        return np.ctypeslib.as_array(buf).reshape(-1, 2)

    def ring_acquire_raw(self) -> int:
        """adsb_ring_acquire without wrapping the buffer in an array (a host loop whose slots are already
        filled): the buffer's address."""
        if not hasattr(self, "_ring_ptr"):
            self._ring_ptr, self._ring_cap = C.c_void_p(), C.c_size_t()
        self._check(self._L.adsb_ring_acquire(self._h, C.byref(self._ring_ptr), C.byref(self._ring_cap)), "adsb_ring_acquire")
        return self._ring_ptr.value

    def ring_submit(self, n_samples: int) -> None:
        self._check(self._L.adsb_ring_submit(self._h, n_samples), "adsb_ring_submit")

    def pending(self) -> int:
        return int(self._L.adsb_pending(self._h))

    def max_in_flight(self) -> int:
        """4, or 8 for a context created for at most 16 buffers per pass (adsb_max_in_flight)."""
        return int(self._L.adsb_max_in_flight(self._h))

    def set_stream(self, hip_stream: int) -> None:
        self._check(self._L.adsb_set_stream(self._h, C.c_void_p(hip_stream)), "adsb_set_stream")

    def set_carry_over(self, enabled: bool) -> None:
        """Opt-in, not the reference's semantics: buffer lead-ins hold the preceding samples."""
        self._check(self._L.adsb_set_carry_over(self._h, 1 if enabled else 0), "adsb_set_carry_over")

    def set_profiling(self, level: int) -> None:
        """0 = no HIP events, 1 = scan kernel + whole chain (default), 2 = every kernel."""
        self._check(self._L.adsb_set_profiling(self._h, int(level)), "adsb_set_profiling")

    def stats_raw(self) -> AdsbStats:
        """The adsb_stats struct itself (no dict building: for tight loops)."""
        if not hasattr(self, "_stats_buf"):
            self._stats_buf = AdsbStats()
        self._check(self._L.adsb_get_stats(self._h, C.byref(self._stats_buf)), "adsb_get_stats")
        return self._stats_buf

    def stats(self) -> dict:
        s = AdsbStats()
        self._check(self._L.adsb_get_stats(self._h, C.byref(s)), "adsb_get_stats")
        return {name: getattr(s, name) for name, _ in AdsbStats._fields_ if name != "reserved"}


_default: Optional[Context] = None


def default_context() -> Context:
    """The process-wide context that stands in for the reference's global filter
    statics (src/icao_filter.rs:8-9).  Device = LOCAL_RANK if set, else 0."""
    global _default
    if _default is None:
        _default = Context(device=int(os.environ.get("LOCAL_RANK", "0")), max_chunks=1)
    return _default
