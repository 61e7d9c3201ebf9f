"""Pin the CPU oracle against every known-answer vector the reference holds for the
path (reference tests/test.rs:19-59, src/crc.rs table entries).  CPU only."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from pathlib import Path

from tests.conftest import GOLDEN, ROOT


def test_fixture_files_are_the_reference_captures(golden):
    for fx in golden["fixtures"]:
        assert hashlib.sha256((GOLDEN / fx["file"]).read_bytes()).hexdigest() == fx["sha256"]


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_reference_frames_exact_count_and_order(golden, oracle_mod, fixture_iq, idx):
    """tests/test.rs routine(): icao_flush, read_test_data, to_mag, demodulate2400.
    Upstream only zips a prefix (test.rs:14); here count and order are exact."""
    fx = golden["fixtures"][idx]
    orc = oracle_mod.Oracle()
    orc.icao_flush()
    data, length = orc.to_mag(fixture_iq[fx["file"]])
    assert length == 131072
    msgs, st = orc.demodulate2400(data, length)
    assert [m["buffer"].hex() for m in msgs] == fx["frames"]
    assert [m["j"] for m in msgs] == fx["j"]
    assert [m["try_phase"] for m in msgs] == fx["try_phase"]
    assert [m["score"] for m in msgs] == fx["score"]
    assert [st.preamble_pass, st.snr_pass, st.quiet_pass, st.trials] == fx["stats"]
    for stale in fx.get("stale_unchecked", []):
        assert stale not in [m["buffer"].hex() for m in msgs]


def test_read_test_data_component_order(oracle_mod, fixture_iq, golden):
    """src/utils.rs:29-31: first i16 of a file pair is im, second is re."""
    f = golden["fixtures"][0]["file"]
    a = oracle_mod.read_test_data(GOLDEN / f)
    assert a.shape == (131072, 2)
    raw = np.fromfile(GOLDEN / f, dtype="<i2").reshape(-1, 2)
    assert np.array_equal(a[:, 0], raw[:, 1]) and np.array_equal(a[:, 1], raw[:, 0])
    assert np.array_equal(a, fixture_iq[f])


def test_crc_table_pins(golden, oracle_mod, hip_lib):
    """All 256 entries of the reference's CRC_TABLE (src/crc.rs:3-260, tests/golden/reference_frames.json
    "crc_table"): the oracle's table, the table regenerated from the generator 0xFFF409, and the table the
    product's host replay scores with (adsb_selftest_crc_table) are that table.  (The device's GF(2) form of
    it, adsb_tables.h, is tied to the same polynomial by the GPU stage test: every address/parity trial's
    residual against this oracle.)"""
    L = oracle_mod.lib()
    assert L.orc_crc_table_entry(0) == 0
    for k, v in golden["crc_table_pins"].items():
        assert L.orc_crc_table_entry(int(k)) == int(v, 16)
    table = [int(v, 16) for v in golden["crc_table"]]
    assert len(table) == 256 and all(int(v, 16) == table[int(k)] for k, v in golden["crc_table_pins"].items())
    assert [L.orc_crc_table_entry(i) for i in range(256)] == table

    def entry(i):
        c = i << 16
        for _ in range(8):
            c = ((c << 1) ^ 0xFFF409) if c & 0x800000 else (c << 1)
        return c & 0xFFFFFF
    assert [entry(i) for i in range(256)] == table
    out = (C.c_uint32 * 256)()
    assert hip_lib.adsb_selftest_crc_table(out) == 0 and list(out) == table
    from dump1090_rs_amd import synth
    assert all(synth.crc24(bytes([i])) == table[i] for i in range(256))   # (what the synthetic frames are built with)


def test_crc_residual_of_golden_frames_is_clean(golden, oracle_mod):
    """Every DF17 frame upstream asserts carries a valid Mode-S parity: residual 0."""
    L = oracle_mod.lib()
    for fx in golden["fixtures"]:
        for h in fx["frames"]:
            b = bytes.fromhex(h)
            buf = (C.c_uint8 * 14)(*b.ljust(14, b"\0"))
            res = L.orc_modes_checksum(buf, len(b) * 8)
            if b[0] >> 3 == 17:
                assert res == 0
            elif b[0] >> 3 == 11:
                assert res & 0xFFFF80 == 0


def test_mag_pipeline_against_numpy_float32(oracle_mod):
    """utils.rs:47-55 restated with numpy float32 + exact fma through float64."""
    L = oracle_mod.lib()
    rng = np.random.default_rng(1)
    re = rng.integers(-32768, 32768, 20000).astype(np.int16)
    im = rng.integers(-32768, 32768, 20000).astype(np.int16)
    edge = np.array([-32768, -32767, -1, 0, 1, 32767], dtype=np.int16)
    re = np.concatenate([re, np.repeat(edge, 6)])
    im = np.concatenate([im, np.tile(edge, 6)])
    fi = im.astype(np.float32) / np.float32(32768)
    fq = re.astype(np.float32) / np.float32(32768)
    t = (fq * fq).astype(np.float32)
    # fi*fi is exact in float64 (24x24 bits) and so is the sum: one rounding = fma
    m2 = (fi.astype(np.float64) * fi.astype(np.float64) + t.astype(np.float64)).astype(np.float32)
    mag = np.sqrt(m2).astype(np.float32)
    o = (mag.astype(np.float64) * 65535.0 + 0.5).astype(np.float32)
    want = np.minimum(np.floor(o), 65535).astype(np.uint16)
    got = np.array([L.orc_mag_sample(int(r), int(i)) for r, i in zip(re, im)], dtype=np.uint16)
    assert np.array_equal(got, want)
    assert L.orc_mag_sample(-32768, -32768) == 65535  # saturates
    assert L.orc_mag_sample(0, 0) == 0


def test_to_mag_layout_and_length_limit(oracle_mod):
    """src/lib.rs:36-50: 326 zeros, samples from 326, zeros after; >131072 panics upstream."""
    orc = oracle_mod.Oracle()
    iq = np.full((100, 2), 1000, dtype=np.int16)
    data, length = orc.to_mag(iq)
    assert length == 100 and not data[:326].any() and not data[426:].any()
    assert (data[326:426] == oracle_mod.lib().orc_mag_sample(1000, 1000)).all()
    with pytest.raises(IndexError):
        orc.to_mag(np.zeros((131073, 2), dtype=np.int16))
    data, length = orc.to_mag(np.zeros((0, 2), dtype=np.int16))
    assert length == 0 and not data.any()


def test_icao_filter_semantics(oracle_mod):
    """src/icao_filter.rs: test(0) always true, B never written, DF18 marker never matches."""
    L = oracle_mod.lib()
    f = oracle_mod.OrcFilter()
    L.orc_icao_flush(C.byref(f))
    assert L.orc_icao_filter_test(C.byref(f), 0) == 1
    assert L.orc_icao_filter_test(C.byref(f), 0xABCDEF) == 0
    L.orc_icao_filter_add(C.byref(f), 0xABCDEF)
    assert L.orc_icao_filter_test(C.byref(f), 0xABCDEF) == 1
    L.orc_icao_filter_add(C.byref(f), 0x123456 | (1 << 25))
    assert L.orc_icao_filter_test(C.byref(f), 0x123456) == 0
    assert not any(f.b)
    # hash: Jenkins one-at-a-time over 3 bytes, 12 bits (icao_filter.rs:19-43)
    def jenkins(a):
        h = 0
        for b in (a & 0xFF, (a >> 8) & 0xFF, (a >> 16) & 0xFF):
            h = (h + b) & (2**64 - 1); h = (h + (h << 10)) & (2**64 - 1); h ^= h >> 6
        h = (h + (h << 3)) & (2**64 - 1); h ^= h >> 11; h = (h + (h << 15)) & (2**64 - 1)
        return h & 0xFFFFFFFF & 4095
    for a in (0, 1, 0xABCDEF, 0xFFFFFF, 0x4840D6):
        assert L.orc_icao_hash(a) == jenkins(a)
    # collisions probe linearly and a full table stops inserting
    L.orc_icao_flush(C.byref(f))
    for a in range(1, 5000):
        L.orc_icao_filter_add(C.byref(f), a)
    assert sum(1 for v in f.a if v) == 4096
    assert L.orc_icao_filter_test(C.byref(f), 1) == 1
    assert L.orc_icao_filter_test(C.byref(f), 0) == 1  # via table B's empty slot


def test_score_table(oracle_mod):
    """src/mode_s/mod.rs:34-139 score values on hand-made messages."""
    from dump1090_rs_amd import synth
    L = oracle_mod.lib()
    f = oracle_mod.OrcFilter()

    def score(msg: bytes):
        buf = (C.c_uint8 * 14)(*msg.ljust(14, b"\0"))
        ln, sc = C.c_int(), C.c_int32()
        some = L.orc_score_modes_message(C.byref(f), buf, 14, C.byref(ln), C.byref(sc))
        return (ln.value, sc.value) if some else None

    L.orc_icao_flush(C.byref(f))
    assert score(bytes(14)) is None                                   # all zero -> None
    df17 = synth.df17_frame(0x4840D6, 0x58C382D690C8AC)
    assert score(df17) == (14, 1400)                                  # first sight adds
    assert score(df17) == (14, 1800)
    assert score(df17[:-1] + bytes([df17[-1] ^ 1])) == (14, -2)       # bad parity
    df11 = synth.df11_frame(0x123456)
    assert score(df11) == (7, 750)
    assert score(df11) == (7, 1600)
    df11_iid = df11[:-1] + bytes([df11[-1] ^ 5])                      # IID 5
    assert score(df11_iid) == (7, 1000)
    assert score(synth.df11_frame(0x777777)[:-1] + bytes([synth.df11_frame(0x777777)[-1] ^ 5])) == (7, -1)
    df18 = bytes([0x90]) + df17[1:11]
    df18 += synth.crc24(df18).to_bytes(3, "big")
    L.orc_icao_flush(C.byref(f))
    assert score(df18) == (14, 1400)
    assert score(df18) == (14, 1400)                                  # addr|1<<25 never matches
    # address/parity: DF4 whose residual is a known address
    L.orc_icao_flush(C.byref(f))
    body = bytes([0x20, 1, 2, 3])
    ap = (synth.crc24(body) ^ 0x4840D6).to_bytes(3, "big")
    assert score(body + ap) == (7, -1)
    score(df17)
    assert score(body + ap) == (7, 1000)
    body20 = bytes([0xA0]) + bytes(range(10))
    ap20 = (synth.crc24(body20) ^ 0x4840D6).to_bytes(3, "big")
    assert score(body20 + ap20) == (14, 1000)
    assert score(body20 + bytes(3)) == (14, -2)
    assert score(bytes([0x08]) + bytes(13)) == (7, -2)                # DF1: unknown
    # residual 0 always "tests true" (icao_filter.rs:71-80)
    assert score(body + synth.crc24(body).to_bytes(3, "big")) == (7, 1000)


def test_slicer_closed_form_matches_phase_walk(oracle_mod):
    """The closed form used by the oracle vs a literal walk of the Phase state machine
    (src/demod_2400.rs:22-83, 158-182) written out here in Python."""
    L = oracle_mod.lib()
    rng = np.random.default_rng(7)
    data = rng.integers(0, 65536, 2000).astype(np.uint16)
    nxt = {0: 2, 2: 4, 4: 1, 1: 3, 3: 0}
    inc = {0: 2, 1: 2, 2: 2, 3: 3, 4: 3}

    def calc(ph, m):
        m0, m1, m2, m3 = (int(x) for x in m[:4])
        return [5*m0-3*m1-2*m2, 4*m0-m1-3*m2, 3*m0+m1-4*m2, 2*m0+3*m1-5*m2, m0+5*m1-5*m2-m3][ph]

    for j in (0, 1, 17, 500, 1600):
        for tp in range(4, 9):
            slice_loc, phase = j + 19 + tp // 5, tp % 5
            want = bytearray(14)
            for b in range(14):
                start, index, byte = phase, 0, 0
                for i in range(8):
                    if calc(phase, data[slice_loc + index: slice_loc + index + 4]) > 0:
                        byte |= 1 << (7 - i)
                    index += inc[phase]
                    phase = nxt[phase]
                want[b] = byte
                slice_loc += index
                phase = (start + 1) % 5
            got = (C.c_uint8 * 14)()
            L.orc_slice_phase(data.ctypes.data, j, tp, got)
            assert bytes(got) == bytes(want)


def test_chunking_keeps_filter_and_has_no_carry_over(oracle_mod, fixture_iq, golden):
    """main.rs:154-167: buffers are independent except for the filter."""
    files = [fx["file"] for fx in golden["fixtures"]]
    stream = np.concatenate([fixture_iq[f] for f in files])
    orc = oracle_mod.Oracle()
    orc.icao_flush()
    msgs, _ = orc.demod_iq(stream)
    # same as running the three buffers one after another on one filter
    orc2 = oracle_mod.Oracle()
    orc2.icao_flush()
    want = []
    for c, f in enumerate(files):
        d, n = orc2.to_mag(fixture_iq[f])
        m, _ = orc2.demodulate2400(d, n)
        want += [(c, x["j"], x["buffer"]) for x in m]
    assert [(m["chunk"], m["j"], m["buffer"]) for m in msgs] == want
    # the filter carried over: buffer 2 now knows addresses from buffers 0 and 1
    assert len(msgs) >= 16


def test_threaded_oracle_matches_single_thread(oracle_mod, fixture_iq):
    """orc_demod_iq_mt (workers per buffer + ordered replay) == orc_demod_iq, frame for frame:
    on the three reference captures back to back (filter carried across buffers) and on
    synthetic IQ with address/parity traffic and a ragged last buffer."""
    import numpy as np
    from dump1090_rs_amd import synth

    three = np.concatenate([fixture_iq[k] for k in sorted(fixture_iq)])
    dense = synth.make_iq(9 * 131072 + 4321, n_bursts=150, seed=11, n_icao=12, df11_every=5)
    for iq in (three, dense):
        one = oracle_mod.Oracle()
        want, st1 = one.demod_iq(iq)
        for threads in (2, 5):
            many = oracle_mod.Oracle()
            got, stn = many.demod_iq(iq, threads=threads)
            assert got == want
            assert (stn.trials, stn.frames, stn.quiet_pass) == (st1.trials, st1.frames, st1.quiet_pass)
        assert len(want) > 0


def test_stage_goldens_are_what_the_oracle_computes(golden, fixture_iq):
    """tests/golden/stage_goldens.json (magnitude checksums, preamble / 3.5 dB / quiet-gate position
    lists, per-trial CRC residuals, address/parity trials) is regenerated from the oracle and must not
    have moved; the counts are also the ones SURVEY.md Appendix B derived independently from the
    reference source before this oracle existed."""
    import json
    import importlib.util
    from tests.conftest import GOLDEN
    spec = importlib.util.spec_from_file_location("make_stage_goldens", GOLDEN / "make_stage_goldens.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    frozen = json.loads((GOLDEN / "stage_goldens.json").read_text())["fixtures"]
    survey_b = {"test_1641427457780.iq": (5872, 2769, 1449, 7245, 3152, 65535),
                "test_1641428165033.iq": (5939, 2696, 1397, 6985, 2960, 43049),
                "test_1641428106243.iq": (5896, 2701, 1342, 6710, 2943, 54533)}
    for fx in golden["fixtures"]:
        now = mod.stage_values(fixture_iq[fx["file"]])
        assert now == frozen[fx["file"]], fx["file"]
        assert (now["n_preamble"], now["n_snr"], now["n_cand"], now["n_trials"], now["n_ap"],
                now["mag_max"][0]) == survey_b[fx["file"]]


def test_oracle_under_address_and_ub_sanitizers(golden):
    """The checker itself under -fsanitize=address,undefined (oracle/Makefile: liboracle_asan.so; the GPU
    pool offers no device sanitizer, the CPU build does): the three reference captures, a ragged synthetic
    stream with bursts at buffer edges, single- and multi-threaded, carry-over mode and the all-trials dump
    -- same answers as the plain build, no report."""
    import shutil
    import subprocess
    import sys
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not Path(asan).exists() or shutil.which("make") is None:
        pytest.skip("no libasan / make in this environment")
    code = r'''
import sys, json, ctypes as C
sys.path.insert(0, %r)
import numpy as np
from oracle import binding
from dump1090_rs_amd import synth
import subprocess
subprocess.run(["make", "-s", "-C", str(binding.HERE), "liboracle_asan.so"], check=True)
L = binding.load(binding.HERE / "liboracle_asan.so")
plain = binding.lib()
golden = json.load(open(%r))
for fx in golden["fixtures"]:
    raw = np.fromfile(%r + "/" + fx["file"], dtype="<i2").reshape(-1, 2)
    iq = np.ascontiguousarray(raw[:, ::-1])
    got, _ = binding.Oracle(L).demod_iq(iq)
    assert [m["buffer"].hex() for m in got] == fx["frames"]
n = 5 * 131072 + 4321
iq = synth.make_iq(n, n_bursts=80, seed=11, n_icao=5, df11_every=3)
synth.add_bursts(iq, [synth.Burst(5 * (131072 * k - 60) + k, 20000, k, synth.df17_frame(0x4840D6, k)) for k in range(1, 5)])
a, _ = binding.Oracle(L).demod_iq(iq)
assert a == binding.Oracle(plain).demod_iq(iq)[0]
assert a == binding.Oracle(L).demod_iq(iq, threads=4)[0]
assert binding.Oracle(L).demod_iq(iq[:400])[0] == binding.Oracle(plain).demod_iq(iq[:400])[0]
assert binding.Oracle(L).demod_iq(iq[:0])[0] == []
mb = binding.OrcMagBuf()
L.orc_to_mag(np.ascontiguousarray(iq[:131072]).ctypes.data, 131072, C.byref(mb))
buf = (C.c_uint8 * (32 * 5 * 131072 // 8))()
assert L.orc_all_trials(C.byref(mb), 0, buf, 5 * 131072 // 8) > 0
print("sanitized oracle ok", len(a))
''' % (str(ROOT), str(GOLDEN / "reference_frames.json"), str(GOLDEN))
    env = dict(__import__("os").environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0 and "sanitized oracle ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
