"""ctypes binding of libadsb_hip.so (include/adsb_hip.h).  No fallback: if the HIP
library is missing this raises, it never routes anywhere else."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "libadsb_hip.so"

ADSB_OK = 0
ADSB_ERR_INVALID = -1
ADSB_ERR_NO_DEVICE = -2
ADSB_ERR_HIP = -3
ADSB_ERR_TOO_LONG = -4
ADSB_ERR_CAPACITY = -5
ADSB_ERR_NOMEM = -6
ADSB_ERR_BUSY = -7
ADSB_ERR_POISONED = -8
ADSB_WAIT_AUTO, ADSB_WAIT_SPIN, ADSB_WAIT_BLOCK = 0, 1, 2
ADSB_FAULT_PHASE1, ADSB_FAULT_PHASE2, ADSB_FAULT_HANG, ADSB_FAULT_RECORDS = 1, 2, 3, 4


class AdsbMsg(C.Structure):
    """adsb_msg (include/adsb_hip.h) == ModeSMessage + provenance."""
    _fields_ = [
        ("msg", C.c_uint8 * 14),
        ("len", C.c_uint8),
        ("try_phase", C.c_uint8),
        ("score", C.c_int32),
        ("j", C.c_uint32),
        ("chunk", C.c_uint64),
        ("signal_level", C.c_double),
    ]


class AdsbStats(C.Structure):
    _fields_ = [
        ("n_samples", C.c_uint64),
        ("n_chunks", C.c_uint64),
        ("n_candidates", C.c_uint64),
        ("n_ap_entries", C.c_uint64),
        ("n_records", C.c_uint64),
        ("n_messages", C.c_uint64),
        ("ms_scan", C.c_float),
        ("ms_match", C.c_float),
        ("ms_records", C.c_float),
        ("ms_total_device", C.c_float),
        ("retries", C.c_uint32),
        ("ms_scan_exclusive", C.c_float),
    ]


class AdsbTrial(C.Structure):
    """adsb_trial (include/adsb_hip.h): one raw trial message, before scoring."""
    _fields_ = [
        ("power", C.c_uint64),
        ("chunk", C.c_uint32),
        ("j_tp", C.c_uint32),
        ("msg", C.c_uint8 * 14),
        ("pad", C.c_uint16),
    ]


class AdsbMultiStats(C.Structure):
    """adsb_multi_stats (include/adsb_hip.h): the capture an adsb_multi collected last."""
    _fields_ = [
        ("n_samples", C.c_uint64),
        ("n_chunks", C.c_uint64),
        ("n_candidates", C.c_uint64),
        ("n_ap_entries", C.c_uint64),
        ("n_records", C.c_uint64),
        ("n_messages", C.c_uint64),
        ("n_addrs_exchanged", C.c_uint64),
        ("n_devices", C.c_uint32),
        ("retries", C.c_uint32),
        ("ms_wall", C.c_float),
        ("ms_phase1_max", C.c_float),
        ("ms_phase2_max", C.c_float),
        ("ms_phase1_span", C.c_float),
        ("ms_phase2_span", C.c_float),
        ("ms_exchange", C.c_float),
        ("ms_replay", C.c_float),
        ("reserved", C.c_float),
    ]


class AdsbError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str = ""):
        self.status = status
        super().__init__(f"{what}: {detail}" if detail else what)


_lib = None


def lib() -> C.CDLL:
    """Load libadsb_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m dump1090_rs_amd.build` "
            "(there is no CPU fallback for the demod_2400 path)")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7.
    # If this library pulled in /opt/rocm's copy first, a later `import torch` would find
    # "No HIP GPUs".  Loading torch first (when it is installed) makes both share
    # torch's runtime; without torch (a C/Rust host) the RUNPATH copy is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(str(LIB_PATH))
    vp, sz = C.c_void_p, C.c_size_t
    L.adsb_create.argtypes = [C.POINTER(vp), C.c_int, sz]
    L.adsb_destroy.argtypes = [vp]
    L.adsb_destroy.restype = None
    L.adsb_set_stream.argtypes = [vp, vp]
    L.adsb_set_profiling.argtypes = [vp, C.c_int]
    L.adsb_icao_flush.argtypes = [vp]
    L.adsb_to_mag.argtypes = [vp, vp, sz, vp, C.POINTER(sz)]
    L.adsb_demodulate2400.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
    L.adsb_demod_iq.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
    L.adsb_demod_iq_device.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
    L.adsb_submit_iq_device.argtypes = [vp, vp, sz]
    L.adsb_collect.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.adsb_pending.argtypes = [vp]
    L.adsb_max_in_flight.argtypes = [vp]
    L.adsb_max_in_flight.restype = C.c_int
    L.adsb_fetch_messages.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.adsb_ring_create.argtypes = [vp, sz]
    L.adsb_ring_acquire.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.adsb_ring_submit.argtypes = [vp, sz]
    L.adsb_read_test_data.argtypes = [C.c_char_p, vp, sz, C.POINTER(sz)]
    L.adsb_get_stats.argtypes = [vp, C.POINTER(AdsbStats)]
    L.adsb_replay_records.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
    L.adsb_set_carry_over.argtypes = [vp, C.c_int]
    L.adsb_set_carry_over.restype = C.c_int
    L.adsb_format_raw.argtypes = [vp, C.c_char_p, sz]
    L.adsb_format_raw.restype = C.c_int
    L.adsb_shard_scan.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
    L.adsb_shard_finish.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
    L.adsb_selftest_mag_digest.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.adsb_strerror.argtypes = [C.c_int]
    L.adsb_strerror.restype = C.c_char_p
    L.adsb_last_error.argtypes = [vp]
    L.adsb_last_error.restype = C.c_char_p
    L.adsb_version.restype = C.c_char_p
    L.adsb_selftest_stage_lists.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), vp, sz, C.POINTER(sz)]
    L.adsb_selftest_stage_lists.restype = C.c_int
    L.adsb_selftest_gate_stages.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), vp, sz, C.POINTER(sz)]
    L.adsb_selftest_gate_stages.restype = C.c_int
    L.adsb_selftest_set_order_polls.argtypes = [vp, C.c_uint32]
    L.adsb_selftest_set_order_polls.restype = C.c_int
    L.adsb_selftest_crc_table.argtypes = [vp]
    L.adsb_selftest_crc_table.restype = C.c_int
    L.adsb_selftest_learned_union.argtypes = [vp, sz, vp, sz, vp, sz, C.POINTER(sz)]
    L.adsb_selftest_learned_union.restype = C.c_int
    ip = C.POINTER(C.c_int)
    L.adsb_multi_create.argtypes = [C.POINTER(vp), ip, C.c_int, sz]
    L.adsb_multi_destroy.argtypes = [vp]
    L.adsb_multi_destroy.restype = None
    L.adsb_multi_device_count.argtypes = [vp]
    L.adsb_multi_max_in_flight.argtypes = [vp]
    L.adsb_multi_shard_range.argtypes = [sz, C.c_int, C.c_int, C.POINTER(sz), C.POINTER(sz)]
    L.adsb_multi_icao_flush.argtypes = [vp]
    L.adsb_multi_demod_iq.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
    L.adsb_multi_demod_iq_device.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), vp, sz, C.POINTER(sz)]
    L.adsb_multi_submit_iq_device.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.adsb_multi_submit_iq.argtypes = [vp, vp, sz]
    L.adsb_multi_host_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.adsb_multi_host_free.argtypes = [vp, vp]
    L.adsb_multi_collect.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.adsb_multi_selftest_tune.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.adsb_multi_selftest_tune.restype = C.c_int
    L.adsb_multi_selftest_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.adsb_multi_selftest_counters.restype = C.c_int
    L.adsb_multi_pending.argtypes = [vp]
    L.adsb_multi_set_wait.argtypes = [vp, C.c_int]
    L.adsb_multi_get_wait.argtypes = [vp]
    L.adsb_multi_set_timeout_ms.argtypes = [vp, C.c_uint32]
    L.adsb_multi_selftest_fail.argtypes = [vp, C.c_uint32, C.c_int, C.c_int]
    L.adsb_multi_fetch_messages.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.adsb_multi_get_stats.argtypes = [vp, C.POINTER(AdsbMultiStats)]
    L.adsb_multi_filter_table.argtypes = [vp, vp]
    L.adsb_multi_last_error.argtypes = [vp]
    L.adsb_multi_last_error.restype = C.c_char_p
    for name in ("adsb_multi_create", "adsb_multi_device_count", "adsb_multi_max_in_flight", "adsb_multi_shard_range",
                 "adsb_multi_icao_flush", "adsb_multi_demod_iq", "adsb_multi_demod_iq_device", "adsb_multi_submit_iq_device",
                 "adsb_multi_submit_iq", "adsb_multi_host_alloc", "adsb_multi_host_free",
                 "adsb_multi_collect", "adsb_multi_pending", "adsb_multi_fetch_messages", "adsb_multi_get_stats",
                 "adsb_multi_filter_table", "adsb_multi_set_wait", "adsb_multi_get_wait", "adsb_multi_set_timeout_ms",
                 "adsb_multi_selftest_fail"):
        getattr(L, name).restype = C.c_int
    L.adsb_host_replays.argtypes = [vp]
    L.adsb_host_replays.restype = C.c_uint64
    L.adsb_host_register.argtypes = [vp, C.c_void_p, C.c_size_t]
    L.adsb_host_register.restype = C.c_int
    L.adsb_host_unregister.argtypes = [vp, C.c_void_p]
    L.adsb_host_unregister.restype = C.c_int
    L.adsb_host_rematches.argtypes = [vp]
    L.adsb_host_rematches.restype = C.c_uint64
    L.adsb_host_sorts.argtypes = [vp]
    L.adsb_host_sorts.restype = C.c_uint64
    for name in ("adsb_create", "adsb_set_stream", "adsb_set_profiling", "adsb_icao_flush",
                 "adsb_to_mag", "adsb_demodulate2400", "adsb_demod_iq", "adsb_demod_iq_device",
                 "adsb_read_test_data", "adsb_get_stats", "adsb_replay_records",
                 "adsb_selftest_mag_digest", "adsb_submit_iq_device", "adsb_collect", "adsb_pending",
                 "adsb_fetch_messages",
                 "adsb_ring_create", "adsb_ring_acquire", "adsb_ring_submit", "adsb_shard_scan",
                 "adsb_shard_finish"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L
