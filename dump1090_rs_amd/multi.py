"""MultiContext: one capture over several GPUs from ONE process (wraps adsb_multi of include/adsb_hip.h).

The reference is one process with one loop and one process-global ICAO filter (dump1090_rs/src/main.rs:154-167,
src/icao_filter.rs:8-9); this keeps that shape -- one handle, one filter, one message list in the reference's
order -- while the capture's contiguous ranges of 131072-sample buffers are demodulated on N devices.  Everything
that makes it work (a thread per device, the two shard phases, the in-memory union of the learned addresses,
the single ordered replay) is inside libadsb_hip.so; nothing here but ctypes.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import AdsbError, AdsbMsg, AdsbMultiStats
from .context import ModeSMessage, _MSG_STRUCT, _as_iq


class MultiContext:
    def __init__(self, devices: Sequence[int], max_chunks_per_device: int):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        st = self._L.adsb_multi_create(C.byref(self._h), arr, len(self.devices), int(max_chunks_per_device))
        if st != _lib.ADSB_OK:
            self._h = C.c_void_p()
            raise AdsbError(st, "adsb_multi_create", self._L.adsb_strerror(st).decode())
        self.max_chunks_per_device = int(max_chunks_per_device)
        self._out_cap = 0
        self._held = []   # per capture in flight: the host array adsb_multi_submit_iq reads from (None for device captures)

    # -- lifetime
    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._L.adsb_multi_destroy(self._h)
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
            detail = self._L.adsb_multi_last_error(self._h).decode() if st in (_lib.ADSB_ERR_HIP, _lib.ADSB_ERR_NOMEM, _lib.ADSB_ERR_POISONED) else ""
            raise AdsbError(st, f"{what}: {self._L.adsb_strerror(st).decode()}", detail)

    # -- the reference's surface, over N devices
    def icao_flush(self) -> None:
        self._check(self._L.adsb_multi_icao_flush(self._h), "adsb_multi_icao_flush")

    def shard_ranges(self, n_samples: int) -> List[Tuple[int, int]]:
        """(first sample, sample count) of every device's contiguous share of a capture of n_samples."""
        out = []
        for k in range(len(self.devices)):
            a, n = C.c_size_t(), C.c_size_t()
            self._check(self._L.adsb_multi_shard_range(n_samples, len(self.devices), k, C.byref(a), C.byref(n)),
                        "adsb_multi_shard_range")
            out.append((a.value, n.value))
        return out

    def _take(self, call, what: str, cap: int) -> List[ModeSMessage]:
        if self._out_cap != cap:
            self._out_buf, self._out_cap = (AdsbMsg * cap)(), cap
            self._out_view = memoryview(self._out_buf).cast("B")
        buf, view = self._out_buf, self._out_view
        n = C.c_size_t()
        st = call(buf, cap, C.byref(n))
        if st == _lib.ADSB_ERR_CAPACITY:   # the capture is consumed (the filter has advanced): fetch, never repeat
            cap = n.value
            buf = (AdsbMsg * cap)()
            view = memoryview(buf).cast("B")
            st = self._L.adsb_multi_fetch_messages(self._h, buf, cap, C.byref(n))
        self._check(st, what)
        return [ModeSMessage(m, ln, sig, score, j, tp, chunk)
                for (m, ln, tp, score, j, chunk, sig) in _MSG_STRUCT.iter_unpack(view[: _MSG_STRUCT.size * n.value])]

    def demod_iq(self, iq, cap: Optional[int] = None) -> List[ModeSMessage]:
        """A host capture of any length: cut into contiguous ranges, copied to the devices, demodulated."""
        a = _as_iq(iq)
        cap = cap or max(4096, a.shape[0] // 256)
        ptr = a.__array_interface__["data"][0]
        return self._take(lambda out, c, n: self._L.adsb_multi_demod_iq(self._h, ptr, a.shape[0], out, c, n),
                          "adsb_multi_demod_iq", cap)

    def _arrays(self, device_ptrs: Sequence[int], n_samples: Sequence[int]):
        k = len(self.devices)
        if len(device_ptrs) != k or len(n_samples) != k:
            raise ValueError("one device pointer and one sample count per device")
        return (C.c_void_p * k)(*[C.c_void_p(int(p)) for p in device_ptrs]), (C.c_size_t * k)(*[int(n) for n in n_samples])

    def demod_iq_device(self, device_ptrs: Sequence[int], n_samples: Sequence[int], cap: Optional[int] = None
                        ) -> List[ModeSMessage]:
        """The shards already resident: device_ptrs[k] / n_samples[k] is device k's contiguous range."""
        ptrs, ns = self._arrays(device_ptrs, n_samples)
        cap = cap or max(4096, sum(int(n) for n in n_samples) // 256)
        return self._take(lambda out, c, n: self._L.adsb_multi_demod_iq_device(self._h, ptrs, ns, out, c, n),
                          "adsb_multi_demod_iq_device", cap)

    # -- pipelined form: up to max_in_flight() captures in flight, results in submission order
    def submit_iq_device(self, device_ptrs: Sequence[int], n_samples: Sequence[int]) -> None:
        ptrs, ns = self._arrays(device_ptrs, n_samples)
        self._check(self._L.adsb_multi_submit_iq_device(self._h, ptrs, ns), "adsb_multi_submit_iq_device")
        self._held.append(None)

    def submit_iq(self, iq) -> None:
        """adsb_multi_submit_iq: a HOST capture of at most len(devices) x max_chunks buffers, asynchronously; every
        device thread copies its range to its device in front of its scan (a DMA per device when `iq` lives in
        memory from host_alloc).  `iq` must stay alive and unchanged until the capture is collected."""
        a = _as_iq(iq)
        self._check(self._L.adsb_multi_submit_iq(self._h, a.__array_interface__["data"][0], a.shape[0]), "adsb_multi_submit_iq")
        self._held.append(a)

    def host_alloc(self, n_samples: int) -> np.ndarray:
        """(n_samples, 2) int16 in pinned host memory that every device of this MultiContext reads by DMA
        (adsb_multi_host_alloc); freed by host_free or with the MultiContext."""
        p = C.c_void_p()
        self._check(self._L.adsb_multi_host_alloc(self._h, int(n_samples) * 4, C.byref(p)), "adsb_multi_host_alloc")
        buf = (C.c_int16 * (2 * int(n_samples))).from_address(p.value)
        return np.frombuffer(buf, dtype=np.int16).reshape(-1, 2)

    def host_free(self, array: np.ndarray) -> None:
        self._check(self._L.adsb_multi_host_free(self._h, array.__array_interface__["data"][0]), "adsb_multi_host_free")

    def submit_raw(self, ptrs, ns) -> None:
        """adsb_multi_submit_iq_device on prepared ctypes arrays (a bench loop builds them once)."""
        self._check(self._L.adsb_multi_submit_iq_device(self._h, ptrs, ns), "adsb_multi_submit_iq_device")
        self._held.append(None)

    def collect(self, cap: int = 1 << 16) -> List[ModeSMessage]:
        try:
            return self._take(lambda out, c, n: self._L.adsb_multi_collect(self._h, out, c, n), "adsb_multi_collect", cap)
        finally:
            del self._held[: max(0, len(self._held) - self.pending())]

    def collect_raw(self, out_buf, cap: int) -> int:
        n = C.c_size_t()
        try:
            self._check(self._L.adsb_multi_collect(self._h, out_buf, cap, C.byref(n)), "adsb_multi_collect")
        finally:
            del self._held[: max(0, len(self._held) - self.pending())]
        return n.value

    def pending(self) -> int:
        return int(self._L.adsb_multi_pending(self._h))

    def max_in_flight(self) -> int:
        return int(self._L.adsb_multi_max_in_flight(self._h))

    def stats(self) -> dict:
        s = AdsbMultiStats()
        self._check(self._L.adsb_multi_get_stats(self._h, C.byref(s)), "adsb_multi_get_stats")
        return {name: getattr(s, name) for name, _ in AdsbMultiStats._fields_ if name != "reserved"}

    def set_wait(self, mode: int) -> None:
        """adsb_multi_set_wait: _lib.ADSB_WAIT_AUTO / _SPIN / _BLOCK (how the handle's threads wait for their devices)."""
        self._check(self._L.adsb_multi_set_wait(self._h, int(mode)), "adsb_multi_set_wait")

    def get_wait(self) -> int:
        return int(self._L.adsb_multi_get_wait(self._h))

    def set_timeout_ms(self, ms: int) -> None:
        self._check(self._L.adsb_multi_set_timeout_ms(self._h, int(ms)), "adsb_multi_set_timeout_ms")

    def selftest_fail(self, captures_from_now: int, shard: int, kind: int) -> None:
        """adsb_multi_selftest_fail: shard `shard` of the capture submitted `captures_from_now` submissions from now fails
        (kind: _lib.ADSB_FAULT_PHASE1 / _PHASE2 / _HANG / _RECORDS; 0 disarms)."""
        self._check(self._L.adsb_multi_selftest_fail(self._h, int(captures_from_now), int(shard), int(kind)), "adsb_multi_selftest_fail")

    def selftest_tune(self, fresh_cap: int = 0, parallel_min: int = 0, score_mode: int = 0) -> None:
        self._check(self._L.adsb_multi_selftest_tune(self._h, int(fresh_cap), int(parallel_min), int(score_mode)), "adsb_multi_selftest_tune")

    def selftest_counters(self) -> dict:
        out = (C.c_uint64 * 8)()
        self._check(self._L.adsb_multi_selftest_counters(self._h, out), "adsb_multi_selftest_counters")
        return {"shards_sorted_on_host": out[0], "fresh_list_fallbacks": out[1], "device_ordered_shards": out[2],
                "parallel_scored_captures": out[3], "device_scored_shards": out[4], "scored_results_used": out[5],
                "scored_results_refused": out[6], "poisoned": out[7]}

    def filter_table(self) -> np.ndarray:
        """Table A of the one ICAO filter (4096 u32, src/icao_filter.rs:8)."""
        t = np.zeros(4096, dtype=np.uint32)
        self._check(self._L.adsb_multi_filter_table(self._h, t.ctypes.data), "adsb_multi_filter_table")
        return t
