"""The C-ABI library loads and exports every symbol include/adsb_hip.h declares.
CPU only: no compute call is made here."""
import ctypes as C
import re

from tests.conftest import ROOT


def declared_functions():
    text = (ROOT / "include" / "adsb_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(adsb_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_reference_surface():
    names = declared_functions()
    for must in ("adsb_create", "adsb_destroy", "adsb_icao_flush", "adsb_to_mag", "adsb_demodulate2400",
                 "adsb_demod_iq", "adsb_demod_iq_device", "adsb_read_test_data"):
        assert must in names


def test_library_exports_every_declared_symbol(hip_lib):
    for name in declared_functions():
        assert hasattr(hip_lib, name), f"libadsb_hip.so does not export {name}"


def test_struct_layouts_match_header(hip_lib):
    from dump1090_rs_amd._lib import AdsbMsg, AdsbStats
    assert C.sizeof(AdsbMsg) == 40
    assert AdsbMsg.len.offset == 14 and AdsbMsg.score.offset == 16 and AdsbMsg.j.offset == 20
    assert AdsbMsg.chunk.offset == 24 and AdsbMsg.signal_level.offset == 32
    assert C.sizeof(AdsbStats) == 6 * 8 + 4 * 4 + 2 * 4


def test_no_cpu_backend(hip_lib):
    """device < 0 is refused outright; there is no CPU path behind the ABI."""
    h = C.c_void_p()
    assert hip_lib.adsb_create(C.byref(h), -1, 1) == -2  # ADSB_ERR_NO_DEVICE
    assert not h.value
    assert b"no CPU fallback" in hip_lib.adsb_strerror(-2)
    assert hip_lib.adsb_create(None, 0, 1) == -1


def test_product_package_never_imports_the_oracle():
    for p in (ROOT / "dump1090_rs_amd").rglob("*"):
        if p.suffix in (".py", ".cpp", ".hip", ".h", ".hpp"):
            text = p.read_text()
            assert "oracle" not in text.lower() or p.name == "synth.py", p
    assert "oracle" not in (ROOT / "dump1090_rs_amd" / "synth.py").read_text().lower()


def test_release_library_reads_no_environment_switches(hip_lib):
    """The measurement knobs (ADSB_DEBUG_STOP, ADSB_STREAM_PRIO, ...) exist only in a library built
    with -DADSB_TUNING: a stray variable cannot change what the release build computes."""
    from dump1090_rs_amd.build import LIB
    blob = LIB.read_bytes()
    if b"ADSB_HOST_TIMES" in blob:
        pytest.skip("a tuning build (-DADSB_TUNING) is installed")
    assert b"ADSB_" not in blob and b"getenv" not in blob


def test_version_string(hip_lib):
    assert hip_lib.adsb_version().startswith(b"adsb_hip")


def test_format_raw_is_the_reference_output_line(hip_lib, golden):
    """dump1090_rs/src/main.rs:172-176: "*" + hex::encode(buffer()) + ";\\n"."""
    from dump1090_rs_amd._lib import AdsbMsg
    for fx in golden["fixtures"]:
        for hexframe in fx["frames"]:
            raw = bytes.fromhex(hexframe)
            m = AdsbMsg()
            C.memmove(m.msg, raw, len(raw))
            m.len = len(raw)
            out = C.create_string_buffer(40)
            n = hip_lib.adsb_format_raw(C.byref(m), out, 40)
            assert n == 2 * len(raw) + 3 and out.value.decode() == f"*{hexframe};\n"
    m = AdsbMsg()
    m.len = 9
    assert hip_lib.adsb_format_raw(C.byref(m), C.create_string_buffer(40), 40) == -1   # not 7 or 14
    m.len = 14
    assert hip_lib.adsb_format_raw(C.byref(m), C.create_string_buffer(40), 31) == -5   # needs 32


def test_feed_tool_fails_loudly_without_a_gpu(hip_lib, golden):
    """adsb_feed (file/pipe -> "*hex;" lines) has no CPU path either."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: covered by the gpu tests")
    feed = ROOT / "dump1090_rs_amd" / "adsb_feed"
    assert feed.exists()
    r = subprocess.run([str(feed), str(ROOT / "tests" / "golden" / golden["fixtures"][0]["file"])],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and r.stdout == "" and "no CPU fallback" in r.stderr


def test_multi_entry_points_refuse_bad_arguments_without_a_device(hip_lib):
    """adsb_multi_*: argument checks and the contiguous split are host code -- no GPU needed, nothing computed."""
    from dump1090_rs_amd import sharding
    L = hip_lib
    h = C.c_void_p()
    assert L.adsb_multi_create(C.byref(h), None, 2, 1) == -1 and not h.value
    assert L.adsb_multi_create(C.byref(h), (C.c_int * 1)(0), 0, 1) == -1
    assert L.adsb_multi_create(C.byref(h), (C.c_int * 1)(0), 65, 1) == -1
    assert L.adsb_multi_create(None, (C.c_int * 1)(0), 1, 1) == -1
    assert L.adsb_multi_create(C.byref(h), (C.c_int * 1)(-1), 1, 1) == -2 and not h.value   # no CPU backend here either
    L.adsb_multi_destroy(None)
    assert L.adsb_multi_device_count(None) == 0 and L.adsb_multi_pending(None) == 0 and L.adsb_multi_max_in_flight(None) == 0
    assert L.adsb_multi_icao_flush(None) == -1 and L.adsb_multi_collect(None, None, 0, None) == -1
    assert L.adsb_multi_last_error(None) == b""
    p = C.c_void_p()
    assert L.adsb_multi_submit_iq(None, None, 0) == -1 and L.adsb_multi_host_alloc(None, 16, C.byref(p)) == -1 and not p.value
    assert L.adsb_multi_host_free(None, None) == -1
    assert L.adsb_multi_selftest_tune(None, 0, 0, 0) == -1 and L.adsb_multi_selftest_counters(None, (C.c_uint64 * 8)()) == -1
    CHUNK = 131072
    for n_samples in (0, 1, CHUNK - 1, CHUNK, CHUNK + 1, 38 * CHUNK - 4321, 4096 * CHUNK):
        for world in (1, 2, 3, 8, 64):
            total = 0
            for k in range(world):
                a, n = C.c_size_t(), C.c_size_t()
                assert L.adsb_multi_shard_range(n_samples, world, k, C.byref(a), C.byref(n)) == 0
                lo, hi = sharding.sample_range(n_samples, world, k)
                assert (a.value, a.value + n.value) == (lo, hi) and a.value == total
                assert n.value % CHUNK == 0 or a.value + n.value == n_samples      # whole buffers but for the capture's end
                total += n.value
            assert total == n_samples
    assert L.adsb_multi_shard_range(10, 0, 0, None, None) == -1 and L.adsb_multi_shard_range(10, 2, 2, None, None) == -1
