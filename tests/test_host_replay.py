"""The host half of the product (ordered replay: score + filter + best-of-5) against the
oracle, with no device: adsb_replay_records is fed every trial the oracle slices."""
import ctypes as C

import numpy as np
import pytest

from dump1090_rs_amd import synth
from dump1090_rs_amd._lib import AdsbMsg


class Trial(C.Structure):
    _fields_ = [("power", C.c_uint64), ("chunk", C.c_uint32), ("j_tp", C.c_uint32),
                ("msg", C.c_uint8 * 14), ("pad", C.c_uint16)]


def all_trials(oracle_mod, iq):
    """Every (chunk, j, try_phase) trial of a stream, from the oracle's slicer."""
    L = oracle_mod.lib()
    out = []
    for c, off in enumerate(range(0, len(iq), 131072)):
        mb = oracle_mod.OrcMagBuf()
        L.orc_to_mag(iq[off:off + 131072].ctypes.data, min(131072, len(iq) - off), C.byref(mb))
        buf = (Trial * (5 * 131072 // 8))()
        n = L.orc_all_trials(C.byref(mb), c, buf, len(buf))
        assert n <= len(buf)
        out += [bytes(buf[i]) for i in range(n)]
    arr = (Trial * len(out)).from_buffer_copy(b"".join(out)) if out else (Trial * 0)()
    return arr


def replay(hip_lib, trials, table=None, cap=100000):
    table = table if table is not None else (C.c_uint32 * 4096)()
    out = (AdsbMsg * cap)()
    n = C.c_size_t()
    st = hip_lib.adsb_replay_records(table, trials, len(trials), out, cap, C.byref(n))
    assert st == 0
    return out[: n.value], table


def same(msgs, want):
    assert len(msgs) == len(want)
    for m, w in zip(msgs, want):
        assert bytes(m.msg) == w["msg"] and m.len == w["len"] and m.score == w["score"]
        assert m.j == w["j"] and m.chunk == w["chunk"] and m.try_phase == w["try_phase"]
        assert m.signal_level == w["signal_level"]  # bit-exact f64


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_replay_of_all_trials_reproduces_reference_frames(hip_lib, oracle_mod, fixture_iq, golden, idx):
    fx = golden["fixtures"][idx]
    trials = all_trials(oracle_mod, fixture_iq[fx["file"]])
    assert len(trials) == fx["stats"][3]
    msgs, _ = replay(hip_lib, trials)
    assert [bytes(m.msg[: m.len]).hex() for m in msgs] == fx["frames"]
    assert [m.j for m in msgs] == fx["j"]
    assert [m.score for m in msgs] == fx["score"]


def test_replay_is_order_independent_on_input_and_keeps_filter(hip_lib, oracle_mod, fixture_iq, golden):
    files = [fx["file"] for fx in golden["fixtures"]]
    stream = np.concatenate([fixture_iq[f] for f in files])
    orc = oracle_mod.Oracle()
    want, _ = orc.demod_iq(stream)
    trials = all_trials(oracle_mod, stream)
    # shuffle: the device lists arrive in no particular order
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(trials))
    shuffled = (Trial * len(trials)).from_buffer_copy(b"".join(bytes(trials[i]) for i in perm))
    msgs, table = replay(hip_lib, shuffled)
    same(msgs, want)
    # filter table equals the oracle's table A
    assert list(table) == list(orc.filter.a)
    # a second pass over the same trials on the warmed filter scores 1800 where it scored 1400
    msgs2, _ = replay(hip_lib, trials, table)
    orc_want2, _ = orc.demod_iq(stream)
    same(msgs2, orc_want2)
    assert any(a.score != b.score for a, b in zip(msgs, msgs2))


def test_replay_on_dense_synthetic(hip_lib, oracle_mod):
    iq = synth.make_iq(4 * 131072, n_bursts=60, n_icao=5, df11_every=4)
    orc = oracle_mod.Oracle()
    want, _ = orc.demod_iq(iq)
    assert len(want) >= 55
    msgs, _ = replay(hip_lib, all_trials(oracle_mod, iq))
    same(msgs, want)
    assert {1400, 1600, 1800} <= {m.score for m in msgs}


def test_replay_edge_cases(hip_lib):
    # no records
    msgs, _ = replay(hip_lib, (Trial * 0)())
    assert len(msgs) == 0
    # capacity error reports the required count
    f = synth.df17_frame(0xABCDEF, 1)
    t = (Trial * 2)()
    for i in range(2):
        t[i].j_tp = (100 + i) | (4 << 24)
        t[i].msg[:] = list(f)
    out = (AdsbMsg * 1)()
    n = C.c_size_t()
    assert hip_lib.adsb_replay_records((C.c_uint32 * 4096)(), t, 2, out, 1, C.byref(n)) == -5
    assert n.value == 2 and out[0].score == 1400
    # strictly-greater selection: equal scores keep the earlier phase; -1 never emits
    t = (Trial * 3)()
    for i, tp in enumerate((4, 5, 6)):
        t[i].j_tp = 7 | (tp << 24)
        t[i].msg[:] = list(f)
        t[i].power = 33 * 65535 * 65535
    msgs, _ = replay(hip_lib, t)
    assert len(msgs) == 1 and msgs[0].score == 1800 and msgs[0].try_phase == 5
    assert msgs[0].signal_level == 1.0
    df4 = bytes([0x20, 1, 2, 3, 9, 9, 9]).ljust(14, b"\1")
    t = (Trial * 1)()
    t[0].j_tp = 7 | (4 << 24)
    t[0].msg[:] = list(df4)
    msgs, _ = replay(hip_lib, t)
    assert len(msgs) == 0
