"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
reference's golden vectors.  Needs a real MI355X: run with -m gpu on the GPU box.

Bar: bit-exact -- frames, count, order, (chunk, j, try_phase, score), and
signal_level as an f64 bit pattern.  The magnitude stage is the only floating-point
stage and its u16 output is bit-exact too (tolerance 0)."""
import ctypes as C

import numpy as np
import pytest

from dump1090_rs_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(hip_lib):
    from dump1090_rs_amd import Context
    c = Context(device=0, max_chunks=64)
    yield c
    c.close()


def assert_same(msgs, want):
    got = [(m.chunk, m.j, m.try_phase, m.score, m.msglen, m.msg.hex(), m.signal_level) for m in msgs]
    exp = [(w["chunk"], w["j"], w["try_phase"], w["score"], w["len"], w["msg"].hex(), w["signal_level"])
           for w in want]
    assert got == exp


# ----------------------------------------------------------------------------- golden
@pytest.mark.parametrize("idx", [0, 1, 2])
def test_reference_fixture_frames_like_upstream_tests(idx, golden, fixture_iq, hip_lib):
    """Reads like reference tests/test.rs: icao_flush, read_test_data, to_mag,
    demodulate2400, compare buffer() -- but exact in count and order."""
    from dump1090_rs_amd import demod_2400, icao_filter, utils
    from tests.conftest import GOLDEN
    fx = golden["fixtures"][idx]
    icao_filter.icao_flush()
    buf = utils.read_test_data(str(GOLDEN / fx["file"]))
    assert np.array_equal(buf, fixture_iq[fx["file"]])
    outbuf = utils.to_mag(buf)
    data = demod_2400.demodulate2400(outbuf)
    assert [a.buffer().hex() for a in data] == fx["frames"]
    assert [a.j for a in data] == fx["j"]
    assert [a.try_phase for a in data] == fx["try_phase"]
    assert [a.score for a in data] == fx["score"]


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_to_mag_bit_exact_on_fixtures(idx, golden, fixture_iq, ctx, oracle_mod):
    fx = golden["fixtures"][idx]
    mb = ctx.to_mag(fixture_iq[fx["file"]])
    want, n = oracle_mod.Oracle().to_mag(fixture_iq[fx["file"]])
    assert mb.length == n == 131072
    assert mb.data.dtype == np.uint16 and np.array_equal(mb.data, want)
    assert not mb.data[:326].any()


def test_fixture_stage_counters(golden, fixture_iq, ctx):
    """candidates == quiet-gate passes, records bounded."""
    for fx in golden["fixtures"]:
        ctx.icao_flush()
        msgs = ctx.demod_iq(fixture_iq[fx["file"]])
        st = ctx.stats()
        assert st["n_candidates"] == fx["stats"][2]
        assert len(msgs) == len(fx["frames"]) == st["n_messages"]
        assert st["n_records"] < 64 and st["retries"] == 0


def test_three_fixtures_as_one_stream_filter_persists(golden, fixture_iq, ctx, oracle_mod):
    files = [fx["file"] for fx in golden["fixtures"]]
    stream = np.concatenate([fixture_iq[f] for f in files])
    orc = oracle_mod.Oracle()
    want, _ = orc.demod_iq(stream)
    ctx.icao_flush()
    assert_same(ctx.demod_iq(stream), want)
    # not flushed: second pass scores against the warmed filter, on both sides
    want2, _ = orc.demod_iq(stream)
    assert_same(ctx.demod_iq(stream), want2)
    assert [w["score"] for w in want] != [w["score"] for w in want2]


# ----------------------------------------------------------------------------- magnitude, exhaustive
def test_magnitude_tail_every_representable_x(ctx, oracle_mod, hip_lib):
    """Every f32 X in {0} U [1, 2^31] (all values im^2 + rn(re^2) can take, and more)
    through sqrt / *65535+0.5 / saturating cast: device digest == CPU digest."""
    L = oracle_mod.lib()
    lo, hi = 0x3F800000, 0x4F000000  # 1.0f .. 2^31
    step = 1 << 24
    for first in [0] + list(range(lo, hi + 1, step)):
        count = 1 if first == 0 else min(step, hi + 1 - first)
        s, x = C.c_uint64(), C.c_uint64()
        assert hip_lib.adsb_selftest_mag_digest(ctx._h, first, count, C.byref(s), C.byref(x)) == 0
        wx = C.c_uint64()
        ws = L.orc_mag_x_digest(first, count, C.byref(wx))
        assert (s.value, x.value) == (ws, wx.value), hex(first)


def test_to_mag_random_and_extreme_pairs(ctx, oracle_mod):
    rng = np.random.default_rng(11)
    iq = rng.integers(-32768, 32768, (131072, 2)).astype(np.int16)
    edge = np.array([-32768, -32767, -16384, -1, 0, 1, 255, 256, 16384, 32767], dtype=np.int16)
    pairs = np.array([(a, b) for a in edge for b in edge], dtype=np.int16)
    iq[: len(pairs)] = pairs
    mb = ctx.to_mag(iq)
    want, _ = oracle_mod.Oracle().to_mag(iq)
    assert np.array_equal(mb.data, want)
    assert mb.data.max() == 65535


# ----------------------------------------------------------------------------- edge cases
def test_empty_and_ragged_inputs(ctx, oracle_mod):
    assert ctx.demod_iq(np.zeros((0, 2), np.int16)) == []
    mb = ctx.to_mag(np.zeros((0, 2), np.int16))
    assert mb.length == 0 and not mb.data.any()
    assert ctx.demodulate2400(mb) == []
    with pytest.raises(IndexError):
        ctx.to_mag(np.zeros((131073, 2), np.int16))
    base = synth.make_iq(2 * 131072 + 5000, n_bursts=24, seed=77)
    for n in (1, 3, 18, 19, 326, 327, 4095, 4096, 4097, 4098, 100001, 131071, 131072, 131073,
              131072 + 4099, 2 * 131072 + 4999):
        iq = base[:n]
        orc = oracle_mod.Oracle()
        want, _ = orc.demod_iq(iq)
        ctx.icao_flush()
        assert_same(ctx.demod_iq(iq), want)


def test_frame_straddling_chunk_edges_matches_reference_semantics(ctx, oracle_mod):
    """No carry-over between buffers (src/utils.rs:44, lib.rs:36-44): a burst cut by a
    chunk edge decodes (or not) exactly as on the CPU."""
    offsets = [-400, -330, -326, -300, -200, -100, -20, -1, 0, 5, 36, 326]
    n = (len(offsets) + 1) * 131072
    iq = synth.noise_numpy(n, seed=9)
    # burst k sits at the start of chunk k+1, shifted by offsets[k] samples (one per edge)
    bursts = [synth.Burst(5 * (131072 * (k + 1) + off) + k % 5, 20000, 3, synth.df17_frame(0xA1B2C3, 1000 + k))
              for k, off in enumerate(offsets)]
    synth.add_bursts(iq, bursts)
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    ctx.icao_flush()
    assert_same(ctx.demod_iq(iq), want)
    assert len(want) >= 5


def test_demodulate2400_honours_caller_magnitudes(ctx, oracle_mod, fixture_iq, golden):
    """demodulate2400 takes any MagnitudeBuffer, not only to_mag output: non-zero
    lead-in samples and a short `length`."""
    from dump1090_rs_amd import MagnitudeBuffer
    f = golden["fixtures"][2]["file"]
    orc = oracle_mod.Oracle()
    data, n = orc.to_mag(fixture_iq[f])
    rng = np.random.default_rng(5)
    data[:326] = rng.integers(0, 4000, 326)
    for length in (131072, 40000, 34916, 34915, 1):
        orc.icao_flush()
        want, _ = orc.demodulate2400(data, length)
        ctx.icao_flush()
        got = ctx.demodulate2400(MagnitudeBuffer(data.copy(), length))
        assert_same(got, want)


def test_demodulate2400_frames_over_the_tile_load_seams(ctx, oracle_mod):
    """Caller-supplied magnitudes are loaded four u16 per lane and 1024 per pass of a workgroup; the
    first lane's first load of a buffer starts two entries before data[0] (out of range, zero).  A
    compiler that folds the "+ 2048 bytes" of the next pass into the instruction's immediate offset
    makes that load out of range too (the hardware adds register and immediate offsets without
    wrapping), which would zero data[1022..1025]: frames laid over every such seam must decode."""
    from dump1090_rs_amd import MagnitudeBuffer
    orc = oracle_mod.Oracle()
    for seam in (1022, 2046, 3070, 7710, 7712 + 1022):
        for back in (40, 120, 200, 260):
            start = seam - 326 - back          # sample at which the preamble starts: the frame spans the seam
            iq = synth.noise_numpy(131072, seed=seam * 7 + back)
            synth.add_bursts(iq, [synth.Burst(5 * start + back % 5, 9000 + 50 * back, back % 16,
                                               synth.df17_frame(0xABC000 + back, 0x1234567890ABC + seam))])
            data, n = orc.to_mag(iq)
            orc.icao_flush()
            want, st = orc.demodulate2400(data, n)
            assert any(abs(w["j"] - (start + 326)) <= 2 for w in want), (seam, back)   # the frame is there
            ctx.icao_flush()
            got = ctx.demodulate2400(MagnitudeBuffer(data.copy(), n))
            assert_same(got, want)
            assert ctx.stats()["n_candidates"] == st.quiet_pass


def test_saturated_and_constant_inputs(ctx, oracle_mod):
    for val in (0, 1, -32768, 32767):
        iq = np.full((131072, 2), val, dtype=np.int16)
        want, _ = oracle_mod.Oracle().demod_iq(iq)
        ctx.icao_flush()
        assert_same(ctx.demod_iq(iq), want)
        assert want == []


# ----------------------------------------------------------------------------- synthetic, multi-chunk
def test_sparse_synthetic_64_chunks(ctx, oracle_mod):
    n = 64 * 131072
    iq = synth.make_iq(n, n_bursts=64)
    want, st = oracle_mod.Oracle().demod_iq(iq)
    ctx.icao_flush()
    assert_same(ctx.demod_iq(iq), want)
    s = ctx.stats()
    assert s["n_candidates"] == st.quiet_pass and s["n_chunks"] == 64
    assert len(want) >= 60


@pytest.mark.parametrize("n_chunks,cut", [(75, 4321), (61, 0), (131, 77777)])
def test_tile_order_covers_every_tile_when_counts_do_not_divide(hip_lib, oracle_mod, n_chunks, cut):
    """Passes with at least as many tiles as the persistent grid has workgroups walk the tiles in
    XCD-aware order (eight contiguous ranges, floor(x * tiles / 8) boundaries): buffer counts that
    are not multiples of 8, a ragged last buffer, a burst in every buffer -- every frame must come
    out once, in order."""
    from dump1090_rs_amd import Context
    n = n_chunks * 131072 - cut
    iq = synth.make_iq(n, n_bursts=2 * n_chunks, seed=1000 + n_chunks)
    want, st = oracle_mod.Oracle().demod_iq(iq, threads=8)
    with Context(0, n_chunks) as c:
        c.icao_flush()
        assert_same(c.demod_iq(iq), want)
        assert c.stats()["n_candidates"] == st.quiet_pass and c.stats()["retries"] == 0
    assert len({w["chunk"] for w in want}) > n_chunks * 0.8


def test_dense_synthetic_all_message_kinds(ctx, oracle_mod):
    """~40 bursts per chunk from 7 addresses, every 3rd a DF11: 750/1000/1400/1600/1800
    transitions and address/parity matches all occur."""
    n = 16 * 131072
    iq = synth.make_iq(n, n_bursts=640, n_icao=7, df11_every=3, seed=31337)
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    ctx.icao_flush()
    assert_same(ctx.demod_iq(iq), want)
    assert {1400, 1600, 1800} <= {w["score"] for w in want}
    assert ctx.stats()["n_records"] >= len(want)


def test_address_parity_hit_from_later_learned_address_is_ordered(ctx, oracle_mod):
    """A DF4 whose parity matches an address that is only learned LATER in the same call
    must not score (bitmap is a superset in time; the host replay restores order)."""
    n = 2 * 131072
    iq = synth.noise_numpy(n, seed=4242)
    icao = 0x4840D6
    body = bytes([0x20, 0x00, 0x05, 0x30])
    ap = (synth.crc24(body) ^ icao).to_bytes(3, "big")
    df4 = body + ap
    bursts = [synth.Burst(5 * 20000, 22000, 1, df4),                       # before the address is known
              synth.Burst(5 * 60000, 22000, 2, synth.df17_frame(icao, 77)),
              synth.Burst(5 * 90000, 22000, 5, df4),                       # after: scores 1000
              synth.Burst(5 * (131072 + 5000), 22000, 9, df4)]
    synth.add_bursts(iq, bursts)
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    ctx.icao_flush()
    got = ctx.demod_iq(iq)
    assert_same(got, want)
    hits = [w for w in want if w["buffer"] == df4]
    assert hits and all(w["chunk"] * 131072 + w["j"] > 60000 for w in hits)
    assert ctx.stats()["n_records"] > len(want)  # the early DF4 was handed back and rejected


@pytest.mark.parametrize("big_chunks", [12, 96])
def test_small_pass_behind_a_pass_still_scanning_sees_its_addresses(hip_lib, oracle_mod, big_chunks):
    """Pipelined: a long pass whose LAST buffer teaches an address, and right behind it a one-buffer pass
    of address/parity frames for that address.  Small passes match on their own scan stream, beside the
    other scan stream: their match must still wait for every earlier pass's scan (the bits it sets in
    the bitmap), or the frames are dropped before the host replay ever sees them.  (Found by the
    randomised soak: one ring case in 5 000.)"""
    import torch
    from dump1090_rs_amd import Context
    icao = 0x3C6589
    body = bytes([0x20, 0x00, 0x05, 0x30])
    df4 = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
    n_big = big_chunks * 131072
    big = synth.noise_numpy(n_big, seed=31 + big_chunks)
    synth.add_bursts(big, [synth.Burst(5 * (n_big - 3000), 22000, 3, synth.df17_frame(icao, 99))])
    small = synth.noise_numpy(131072, seed=32)
    synth.add_bursts(small, [synth.Burst(5 * (4000 + 9000 * q) + q, 21000, q, df4) for q in range(6)])
    orc = oracle_mod.Oracle()
    orc.icao_flush()
    want = [orc.demod_iq(big)[0], orc.demod_iq(small)[0]]
    assert sum(w["buffer"] == df4 and w["score"] == 1000 for w in want[1]) >= 6
    d_big, d_small = torch.from_numpy(big).cuda(), torch.from_numpy(small).cuda()
    torch.cuda.synchronize()
    with Context(0, big_chunks) as c:
        for rep in range(10):
            c.icao_flush()
            c.submit_iq_device(d_big.data_ptr(), n_big)
            c.submit_iq_device(d_small.data_ptr(), 131072)
            assert_same(c.collect(), want[0])
            assert_same(c.collect(), want[1])


def test_pass_behind_a_ring_fed_small_pass_waits_for_its_scan(hip_lib, oracle_mod):
    """The other direction: a ring-fed pass of six buffers (3 MB still crossing PCIe when it is submitted)
    teaches an address; a device-resident pass of twenty buffers submitted right behind it -- its scan and
    match are over before that copy is -- holds the address/parity frames that need it."""
    import torch
    from dump1090_rs_amd import Context
    icao = 0x7C1234
    body = bytes([0x20, 0x00, 0x11, 0x30])
    df4 = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
    n_w, n_x = 6 * 131072, 20 * 131072
    w = synth.noise_numpy(n_w, seed=61)
    synth.add_bursts(w, [synth.Burst(5 * (131072 * q + 40000), 20000, q, synth.df17_frame(icao, q)) for q in range(6)])
    x = synth.noise_numpy(n_x, seed=62)
    synth.add_bursts(x, [synth.Burst(5 * (131072 * (2 * q) + 9000) + q % 5, 20000, q, df4) for q in range(10)])
    orc = oracle_mod.Oracle()
    orc.icao_flush()
    want = [orc.demod_iq(w)[0], orc.demod_iq(x)[0]]
    assert sum(f["buffer"] == df4 and f["score"] == 1000 for f in want[1]) >= 10
    d_x = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    with Context(0, 20) as c:
        c.ring_create(n_w)
        for rep in range(5):
            c.icao_flush()
            buf = c.ring_acquire()
            buf[:n_w] = w
            c.ring_submit(n_w)
            c.submit_iq_device(d_x.data_ptr(), n_x)
            assert_same(c.collect(), want[0])
            assert_same(c.collect(), want[1])


def test_flushed_small_pass_waits_for_a_ring_fed_long_pass_before_clearing_its_bitmap(hip_lib, oracle_mod):
    """A twenty-buffer pass fed through the ring (10 MB still crossing PCIe) has address/parity frames for
    an address learned before it; an icao_flush and a one-buffer device-resident pass behind it are through
    long before that copy is -- and the small pass's records kernel is the one that clears the retired
    bitmap: it has to wait for the long pass's match."""
    import torch
    from dump1090_rs_amd import Context
    icao = 0x5A5A5A
    body = bytes([0x28, 0x00, 0x07, 0x31])
    df5 = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
    first = synth.noise_numpy(131072, seed=71)
    synth.add_bursts(first, [synth.Burst(5 * 30000, 20000, 4, synth.df17_frame(icao, 5))])
    n_long = 20 * 131072
    long_ = synth.noise_numpy(n_long, seed=72)
    synth.add_bursts(long_, [synth.Burst(5 * (131072 * (2 * q) + 7000) + q % 5, 20000, q, df5) for q in range(10)])
    small = synth.noise_numpy(131072, seed=73)
    orc = oracle_mod.Oracle()
    orc.icao_flush()
    w_first, w_long = orc.demod_iq(first)[0], orc.demod_iq(long_)[0]
    orc.icao_flush()
    w_small = orc.demod_iq(small)[0]
    assert sum(f["buffer"] == df5 and f["score"] == 1000 for f in w_long) >= 10
    d_first, d_small = torch.from_numpy(first).cuda(), torch.from_numpy(small).cuda()
    torch.cuda.synchronize()
    with Context(0, 20) as c:
        c.ring_create(n_long)
        for rep in range(4):
            c.icao_flush()
            assert_same(c.demod_iq_device(d_first.data_ptr(), 131072), w_first)
            buf = c.ring_acquire()
            buf[:n_long] = long_
            c.ring_submit(n_long)
            c.icao_flush()
            c.submit_iq_device(d_small.data_ptr(), 131072)
            assert_same(c.collect(), w_long)
            assert_same(c.collect(), w_small)


def test_small_flushed_pass_does_not_clear_the_bitmap_under_a_long_pass_in_flight(hip_lib, oracle_mod):
    """A long pass in flight still has to match its address/parity trials against the addresses learned
    before it; an icao_flush and a one-buffer pass right behind it retire that bitmap, and the small pass's
    records kernel -- which runs long before the long pass's scan is over -- is the one that clears it: not
    before the long pass's match is through."""
    import torch
    from dump1090_rs_amd import Context
    icao = 0xABCDEF
    body = bytes([0x28, 0x00, 0x1A, 0x30])
    df5 = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
    first = synth.noise_numpy(131072, seed=51)
    synth.add_bursts(first, [synth.Burst(5 * 30000, 20000, 4, synth.df17_frame(icao, 5))])
    n_long = 160 * 131072
    long_ = synth.noise_numpy(n_long, seed=52)
    synth.add_bursts(long_, [synth.Burst(5 * (131072 * (3 + 13 * q) + 7000) + q % 5, 20000, q, df5) for q in range(12)])
    small = synth.noise_numpy(131072, seed=53)
    synth.add_bursts(small, [synth.Burst(5 * 50000, 20000, 7, synth.df17_frame(0x111111, 6))])
    orc = oracle_mod.Oracle()
    orc.icao_flush()
    w_first = orc.demod_iq(first)[0]
    w_long = orc.demod_iq(long_)[0]
    orc.icao_flush()
    w_small = orc.demod_iq(small)[0]
    assert sum(w["buffer"] == df5 and w["score"] == 1000 for w in w_long) >= 12
    d_first, d_long, d_small = (torch.from_numpy(x).cuda() for x in (first, long_, small))
    torch.cuda.synchronize()
    with Context(0, 160) as c:
        for rep in range(6):
            c.icao_flush()
            assert_same(c.demod_iq_device(d_first.data_ptr(), 131072), w_first)
            c.submit_iq_device(d_long.data_ptr(), n_long)
            c.icao_flush()
            c.submit_iq_device(d_small.data_ptr(), 131072)
            assert_same(c.collect(), w_long)
            assert_same(c.collect(), w_small)


def test_device_resident_entry_point_and_determinism(ctx, oracle_mod):
    import torch
    n = 32 * 131072
    t = synth.make_iq_torch(n, n_bursts=100, seed=2024, device="cuda")
    torch.cuda.synchronize()
    want, _ = oracle_mod.Oracle().demod_iq(t.cpu().numpy())
    for _ in range(3):
        ctx.icao_flush()
        assert_same(ctx.demod_iq_device(t.data_ptr(), n), want)


def test_context_isolation(hip_lib, oracle_mod, fixture_iq, golden):
    """Two contexts are two independent streams (their own filters)."""
    from dump1090_rs_amd import Context
    f = golden["fixtures"][0]["file"]
    with Context(0, 1) as a, Context(0, 1) as b:
        a.icao_flush(); b.icao_flush()
        first = a.demod_iq(fixture_iq[f])
        again = a.demod_iq(fixture_iq[f])
        fresh = b.demod_iq(fixture_iq[f])
        assert [m.score for m in first] == [m.score for m in fresh] == golden["fixtures"][0]["score"]
        assert [m.score for m in again] != [m.score for m in first]


# ----------------------------------------------------------------------------- full size (BASELINE configs 2 and 5)
@pytest.mark.parametrize("n_bursts", [64, 5000])
def test_full_256mib_buffer_bit_exact(hip_lib, oracle_mod, n_bursts):
    """512 chunks = 256 MiB of IQ, device resident; the oracle needs a couple of seconds."""
    import torch
    from dump1090_rs_amd import Context
    n = 512 * 131072
    t = synth.make_iq_torch(n, n_bursts=n_bursts, device="cuda")
    torch.cuda.synchronize()
    host = t.cpu().numpy()
    want, st = oracle_mod.Oracle().demod_iq(host, cap=1 << 20)
    with Context(0, 512) as c:
        c.icao_flush()
        c.demod_iq_device(t.data_ptr(), n, cap=1 << 20)   # (tells the context how dense this stream is)
        c.icao_flush()
        got = c.demod_iq_device(t.data_ptr(), n, cap=1 << 20)
        s = c.stats()
        host_sorts, host_replays = c._L.adsb_host_sorts(c._h), c._L.adsb_host_replays(c._h)
    assert_same(got, want)
    assert s["n_candidates"] == st.quiet_pass and s["retries"] == 0
    if n_bursts >= 5000:
        # dense: the device handed the second pass over in replay order and scored it itself (mode_s
        # scoring + best-of-5 against its copy of the filter); the host did neither
        assert (host_sorts, host_replays) == (1, 1)
    else:
        assert (host_sorts, host_replays) == (2, 2)   # sparse: a few hundred records, the host's business
    injected = {b.frame for b in synth.plan_bursts(n, n_bursts)}
    assert len(injected & {w["buffer"] for w in want}) >= 0.95 * len(injected)


def test_list_overflow_falls_back_and_stays_exact(hip_lib, oracle_mod):
    """A context sized for 1 chunk given far denser input than its lists expect still
    returns the exact answer (per-chunk fallback)."""
    from dump1090_rs_amd import Context
    n = 8 * 131072
    iq = synth.make_iq(n, n_bursts=300, n_icao=3, seed=99)
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    with Context(0, 1) as c:
        c.icao_flush()
        assert_same(c.demod_iq(iq), want)


def test_overflow_fallback_after_a_later_flush_keeps_earlier_addresses(hip_lib, oracle_mod):
    """Pass i overflows the lists while pass i+1, submitted after an icao_flush, is already in
    flight: the bitmaps have rotated and the retired one has been cleared by then, so the
    fallback has to put the addresses the filter held before pass i back into the superset --
    or the address/parity frames of pass i that rely on them vanish."""
    import torch
    from dump1090_rs_amd import Context
    n = 131072
    icao = 0x3C6589
    body = bytes([0x20, 0x00, 0x05, 0x30])
    df4 = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")      # scores only while icao is known
    host = [synth.noise_numpy(n, seed=900 + k) for k in range(3)]
    synth.add_bursts(host[0], [synth.Burst(5 * (30000 * k + 777) + k, 21000, k, synth.df17_frame(icao, k))
                               for k in range(1, 4)])
    # pass 1: a periodic stretch that overflows a one-buffer context's lists, DF4s in the clean part
    a, b = 40000, 125000
    per = np.array(ADVERSARIAL_PERIODS[1], dtype=np.int16)     # 3.4 address/parity trials per position
    host[1][a:b, 0] = np.tile(per, (b - a) // len(per) + 1)[: b - a]
    host[1][a:b, 1] = 0
    synth.add_bursts(host[1], [synth.Burst(5 * (9000 * k + 333) + k, 21000, k, df4) for k in range(1, 4)])
    orc = oracle_mod.Oracle()
    orc.icao_flush()
    want = [orc.demod_iq(host[0], cap=1 << 18)[0], orc.demod_iq(host[1], cap=1 << 18)[0]]
    orc.icao_flush()
    want.append(orc.demod_iq(host[2], cap=1 << 18)[0])
    assert sum(w["buffer"] == df4 for w in want[1]) >= 2
    bufs = [torch.from_numpy(h).cuda() for h in host]
    torch.cuda.synchronize()
    with Context(0, 1) as c:
        c.icao_flush()
        c.submit_iq_device(bufs[0].data_ptr(), n)
        assert_same(c.collect(cap=1 << 18), want[0])
        c.submit_iq_device(bufs[1].data_ptr(), n)
        c.icao_flush()
        c.submit_iq_device(bufs[2].data_ptr(), n)
        got1 = c.collect(cap=1 << 18)
        assert c.stats()["retries"] > 0                  # the scenario really went through the fallback
        assert_same(got1, want[1])
        assert_same(c.collect(cap=1 << 18), want[2])


def test_output_array_too_small_loses_nothing_and_does_not_rerun_the_pass(hip_lib, oracle_mod, fixture_iq, golden):
    """ADSB_ERR_CAPACITY: the pass is consumed and the filter has advanced (like demodulate2400
    having returned its Vec), so the list is fetched, never recomputed -- a recomputation would
    score against the advanced filter (1400 -> 1800, extra address/parity frames)."""
    import ctypes as C
    from dump1090_rs_amd import Context
    from dump1090_rs_amd._lib import AdsbMsg, ADSB_ERR_CAPACITY, ADSB_ERR_INVALID
    fx = golden["fixtures"][2]
    iq = fixture_iq[fx["file"]]
    orc = oracle_mod.Oracle()
    orc.icao_flush()
    want1, _ = orc.demod_iq(iq)
    want2, _ = orc.demod_iq(iq)        # second time over the same capture: the filter knows the addresses
    assert [w["score"] for w in want1] != [w["score"] for w in want2]
    with Context(0, 1) as c:
        L, h = c._L, c._h
        n = C.c_size_t()
        buf = (AdsbMsg * 64)()
        assert L.adsb_fetch_messages(h, buf, 64, C.byref(n)) == ADSB_ERR_INVALID   # nothing held
        c.icao_flush()
        a = np.ascontiguousarray(iq)
        assert L.adsb_demod_iq(h, a.ctypes.data, a.shape[0], buf, 2, C.byref(n)) == ADSB_ERR_CAPACITY
        assert n.value == len(want1)
        assert [bytes(m.msg)[: m.len] for m in buf[:2]] == [w["buffer"] for w in want1[:2]]
        assert L.adsb_fetch_messages(h, buf, 3, C.byref(n)) == ADSB_ERR_CAPACITY and n.value == len(want1)
        assert L.adsb_fetch_messages(h, buf, 64, C.byref(n)) == 0 and n.value == len(want1)
        assert [(m.j, m.try_phase, m.score, bytes(m.msg)[: m.len]) for m in buf[: n.value]] == \
            [(w["j"], w["try_phase"], w["score"], w["buffer"]) for w in want1]
        # the Python wrapper does the same by itself, and the filter advanced exactly once
        assert_same(c.demod_iq(iq, cap=1), want2)


def test_compiled_c_host_runs_the_reference_test_routine(hip_lib, golden):
    """tests/abi_host.c: plain C over include/adsb_hip.h (gcc, no ctypes) doing reference tests/test.rs:7-17
    on the three captures; exit status 0 = exactly the frames upstream asserts, in order."""
    import subprocess
    from tests.conftest import GOLDEN, ROOT
    exe = ROOT / "tests" / "abi_host"
    assert exe.exists(), "tests/abi_host was not built (dump1090_rs_amd.build.build_abi_host)"
    for fx in golden["fixtures"]:
        r = subprocess.run([str(exe), str(GOLDEN / fx["file"]), *fx["frames"]], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert [ln.split()[0] for ln in r.stdout.splitlines()] == fx["frames"]
    # one frame too few / a wrong frame is a failure, not a prefix match
    fx = golden["fixtures"][0]
    assert subprocess.run([str(exe), str(GOLDEN / fx["file"]), *fx["frames"][:-1]], capture_output=True).returncode == 1


def test_compiled_c_host_runs_the_live_receiver_loop(hip_lib, oracle_mod, tmp_path):
    """tests/abi_host --live: the receiver's loop (dump1090_rs/src/main.rs:154-167) in C over include/adsb_hip.h -- a ring
    slot acquired, 131072 samples copied in, submitted, the oldest pass collected when every slot is out, the ICAO
    filter never flushed: 40 buffers of one stream that keeps teaching its aircraft; the frames it writes out are the
    oracle's ONE stream over the same bytes, signal levels bit for bit."""
    import struct
    import subprocess
    from tests.conftest import ROOT
    exe = ROOT / "tests" / "abi_host"
    assert exe.exists(), "tests/abi_host was not built (dump1090_rs_amd.build.build_abi_host)"
    iq = synth.make_iq(40 * 131072, n_bursts=500, seed=9090, n_icao=30, df11_every=5)
    orc = oracle_mod.Oracle()
    orc.icao_flush()
    want, _ = orc.demod_iq(iq, cap=1 << 16)
    src, out = tmp_path / "stream.bin", tmp_path / "frames.out"
    iq.tofile(src)
    r = subprocess.run([str(exe), "--live", str(src), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("live: 40 passes, "), r.stdout + r.stderr
    got = [ln.split() for ln in out.read_text().splitlines()]
    assert [(int(g[0]), int(g[1]), int(g[2]), int(g[3]), g[4], g[5]) for g in got] == \
           [(w["chunk"], w["j"], w["try_phase"], w["score"], w["buffer"].hex(), struct.pack(">d", w["signal_level"]).hex()) for w in want]
    assert len(want) > 300


def test_device_side_scoring_follows_the_filter_across_pipelined_passes(hip_lib, oracle_mod):
    """Passes of more than 16 buffers are scored on the device against its own copy of the ICAO
    filter: 750 -> 1600, 1400 -> 1800 and address/parity frames that depend on addresses learned earlier
    in the same pass, in earlier passes in flight, and not after an icao_flush -- against the oracle,
    with the host never scoring (adsb_host_replays) until a small pass forces it to, after which the
    device copy is rebuilt and takes over again."""
    import torch
    from dump1090_rs_amd import Context
    n = 20 * 131072
    icaos = [0x4840D6, 0x3C6589, 0xA1B2C3]
    body = bytes([0x20, 0x00, 0x05, 0x30])
    df4 = lambda icao: body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
    host = [synth.make_iq(n, n_bursts=1600, seed=700 + k, n_icao=5, df11_every=3) for k in range(4)]
    # address/parity frames whose address is learned late in buffer 0 of pass 0, and only there
    synth.add_bursts(host[0], [synth.Burst(5 * (131072 * 3 + 5000) + 2, 21000, 2, synth.df17_frame(icaos[0], 7))])
    for k in (1, 2, 3):
        synth.add_bursts(host[k], [synth.Burst(5 * (131072 * (2 + q) + 900 * q + 333) + q, 21000, q, df4(icaos[0]))
                                   for q in range(1, 6)])
    bufs = [torch.from_numpy(h).cuda() for h in host]
    torch.cuda.synchronize()
    orc = oracle_mod.Oracle()
    want = []
    for k, flush in ((0, True), (1, False), (2, False), (3, True), (1, False), (0, False)):
        if flush:
            orc.icao_flush()
        want.append(orc.demod_iq(host[k])[0])
    assert sum(w["buffer"] == df4(icaos[0]) for w in want[1]) >= 3 and not any(w["buffer"] == df4(icaos[0]) for w in want[3])
    assert {1000, 1400, 1600, 1800} <= {w["score"] for ws in want for w in ws}
    with Context(0, 32) as c:
        c.icao_flush()
        c.demod_iq_device(bufs[2].data_ptr(), n)             # (tells the context how dense this stream is)
        assert c._L.adsb_host_replays(c._h) == 1
        got = []
        c.icao_flush()
        c.submit_iq_device(bufs[0].data_ptr(), n)
        c.submit_iq_device(bufs[1].data_ptr(), n)
        c.submit_iq_device(bufs[2].data_ptr(), n)
        got.append(c.collect())
        c.icao_flush()
        c.submit_iq_device(bufs[3].data_ptr(), n)
        got.append(c.collect())
        c.submit_iq_device(bufs[1].data_ptr(), n)
        got.append(c.collect())
        c.submit_iq_device(bufs[0].data_ptr(), n)
        got += [c.collect(), c.collect(), c.collect()]
        for g, w in zip(got, want):
            assert_same(g, w)
        assert c.stats()["n_records"] >= 4096
        assert c._L.adsb_host_replays(c._h) == 1 and c._L.adsb_host_sorts(c._h) == 1
        # a small pass is the host's; once the stream is dense again the passes in flight are finished
        # early, the device's copy of the filter is rebuilt from the host's, and it takes over again
        small = host[2][: 3 * 131072]
        assert_same(c.demod_iq(small), orc.demod_iq(small)[0])
        assert c._L.adsb_host_replays(c._h) == 2
        seq = [3, 2, 1, 0, 3]
        for k in seq[:3]:
            c.submit_iq_device(bufs[k].data_ptr(), n)
        outs = [c.collect()]
        for k in seq[3:]:
            c.submit_iq_device(bufs[k].data_ptr(), n)
            outs.append(c.collect())
        outs += [c.collect(), c.collect()]
        for k, g in zip(seq, outs):
            assert_same(g, orc.demod_iq(host[k])[0])
        # (the small pass does not change the context's idea of the stream's density: the first large
        # pass behind it finds the device's copy of the filter stale, rebuilds it and is scored there)
        assert c._L.adsb_host_replays(c._h) == 2


def test_shard_pass_behind_a_flush_between_device_scored_passes_resyncs_the_device_filter(hip_lib, oracle_mod):
    """icao_flush + adsb_shard_scan / adsb_shard_finish between dense, device-scored passes: the shard call
    consumes the flush (host filter emptied), so the device's copy of the filter must be disowned and
    rebuilt too -- or the passes after it are scored against the pre-flush addresses (1600 / 1800 /
    address-parity 1000 for addresses a flushed filter does not know)."""
    import torch
    from dump1090_rs_amd import Context
    from dump1090_rs_amd.context import replay_records
    n = 20 * 131072
    icao = 0x4840D6
    body = bytes([0x20, 0x00, 0x05, 0x30])
    df4 = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
    host = [synth.make_iq(n, n_bursts=1600, seed=1500 + k, n_icao=5, df11_every=3) for k in range(3)]
    synth.add_bursts(host[0], [synth.Burst(5 * (131072 * 2 + 4000) + 1, 21000, 1, synth.df17_frame(icao, 3))])
    # host[2] carries address/parity frames for an address only host[0] teaches: after the flush they must vanish
    synth.add_bursts(host[2], [synth.Burst(5 * (131072 * (1 + q) + 700 * q + 211) + q, 21000, q, df4) for q in range(1, 6)])
    shard_iq = synth.make_iq(4 * 131072, n_bursts=30, seed=1599)
    bufs = [torch.from_numpy(h).cuda() for h in host]
    sdev = torch.from_numpy(shard_iq).cuda()
    torch.cuda.synchronize()
    orc = oracle_mod.Oracle()
    orc.icao_flush()
    want = [orc.demod_iq(host[k])[0] for k in (1, 0, 2)]
    assert sum(w["buffer"] == df4 for w in want[2]) >= 3          # known before the flush ...
    orc.icao_flush()
    want += [orc.demod_iq(host[k])[0] for k in (2, 1)]
    # (a shard's records are replayed by whoever holds all shards, through a filter of their own: the
    # context's filter only sees the flush the shard call consumed)
    want_shard = oracle_mod.Oracle().demod_iq(shard_iq)[0]
    assert not any(w["buffer"] == df4 for w in want[3])            # ... and not after it
    with Context(0, 32) as c:
        c.icao_flush()
        got = [c.demod_iq_device(bufs[1].data_ptr(), n)]          # (tells the context how dense this stream is)
        c.submit_iq_device(bufs[0].data_ptr(), n)
        c.submit_iq_device(bufs[2].data_ptr(), n)
        got += [c.collect(), c.collect()]
        replays = c._L.adsb_host_replays(c._h)
        assert replays == 1                                        # the two pipelined passes were the device's
        c.icao_flush()
        c.shard_scan(sdev.data_ptr(), 4 * 131072)
        assert_same(replay_records(c.shard_finish(np.zeros(0, np.uint32))), want_shard)
        c.submit_iq_device(bufs[2].data_ptr(), n)
        c.submit_iq_device(bufs[1].data_ptr(), n)
        got += [c.collect(), c.collect()]
        for g, w in zip(got, want):
            assert_same(g, w)
        # scored on the device again, against a copy of the filter rebuilt after the flush
        assert c._L.adsb_host_replays(c._h) == replays

def test_device_side_scoring_hands_over_before_the_filter_table_fills(hip_lib, oracle_mod):
    """icao_filter_add gives up silently once its 4096-slot table is full (src/icao_filter.rs:46-62) --
    the one behaviour the parallel scoring cannot reproduce, so the device's result is only taken while
    the host's table is at least 64 entries from full; past that the host scores the passes itself, from
    the records the device kept.  A dense stream with more distinct addresses than the table holds and
    no flush: identical to the oracle through the hand-over and on with a full table (where most
    frames stay at 1400: their address is never stored)."""
    import torch
    from dump1090_rs_amd import Context
    n = 20 * 131072
    host = [synth.make_iq(n, n_bursts=1400, seed=900 + k, n_icao=20000) for k in range(6)]
    bufs = [torch.from_numpy(h).cuda() for h in host]
    torch.cuda.synchronize()
    orc = oracle_mod.Oracle()
    orc.icao_flush()
    want = [orc.demod_iq(h)[0] for h in host]
    seen = set()
    for ws in want:
        seen |= {int.from_bytes(w["msg"][1:4], "big") for w in ws if w["score"] >= 1400}
    assert len(seen) > 6000                                                     # more addresses than slots
    assert sum(w["score"] == 1400 for w in want[5]) > sum(w["score"] == 1800 for w in want[5])   # the table is full
    with Context(0, 32) as c:
        c.icao_flush()
        got = [c.demod_iq_device(bufs[0].data_ptr(), n)]      # (tells the context how dense this stream is)
        assert c._L.adsb_host_replays(c._h) == 1
        for k in range(1, 6):
            c.submit_iq_device(bufs[k].data_ptr(), n)
            if k >= 2:
                got.append(c.collect())
        got.append(c.collect())
        for g, w in zip(got, want):
            assert_same(g, w)
        # the first pass and every pass from the hand-over on are the host's; at least one in between
        # was the device's
        assert 3 <= c._L.adsb_host_replays(c._h) <= 5, c._L.adsb_host_replays(c._h)


def test_dense_stream_with_one_overfull_buffer_bucket_falls_back_and_stays_exact(hip_lib, oracle_mod):
    """On a dense stream every hit goes into its buffer's bucket of 1024 (device-side ordering).  One
    buffer packed with back-to-back frames holds more than that while no other list is anywhere near
    full: the pass must be flagged and redone buffer by buffer, identical to the oracle, and the passes
    around it must not notice."""
    import torch
    from dump1090_rs_amd import Context
    n = 20 * 131072
    host = [synth.make_iq(n, n_bursts=1500, seed=1200 + k, n_icao=40) for k in range(3)]
    packed = synth.noise_numpy(131072, seed=77)
    synth.add_bursts(packed, [synth.Burst(5 * (200 + 300 * q) + q % 5, 14000 + 10 * q, q % 16,
                                          synth.df17_frame(0xA00000 + 0x101 * (q % 40), q)) for q in range(430)])
    host[1][5 * 131072:6 * 131072] = packed
    bufs = [torch.from_numpy(h).cuda() for h in host]
    torch.cuda.synchronize()
    orc = oracle_mod.Oracle()
    orc.icao_flush()
    want = [orc.demod_iq(h)[0] for h in host]
    assert sum(w["chunk"] == 5 for w in want[1]) >= 400
    with Context(0, 32) as c:
        c.icao_flush()
        got = [c.demod_iq_device(bufs[0].data_ptr(), n)]      # (tells the context how dense this stream is)
        c.submit_iq_device(bufs[1].data_ptr(), n)
        c.submit_iq_device(bufs[2].data_ptr(), n)
        got.append(c.collect())
        assert c.stats()["retries"] >= 1                        # the overfull bucket did overflow
        got.append(c.collect())
        for g, w in zip(got, want):
            assert_same(g, w)


# ----------------------------------------------------------------------------- stages
def test_stage_lists_match_the_stage_goldens_and_the_oracle(hip_lib, oracle_mod, golden, fixture_iq):
    """Not only frames: the device's magnitudes, the positions at which check_preamble matches, those that
    pass the 3.5 dB test, those its gates let through (all five digests of tests/golden/stage_goldens.json)
    and its address/parity trials (position, try_phase, CRC residual) on the reference captures, and
    list against list with the oracle on a synthetic stream whose buffers are ragged and carry bursts."""
    import hashlib
    import json
    import zlib
    import torch
    from dump1090_rs_amd import Context
    from tests.conftest import GOLDEN
    frozen = json.loads((GOLDEN / "stage_goldens.json").read_text())["fixtures"]
    sha = lambda a: hashlib.sha256(np.asarray(a, dtype="<u8").tobytes()).hexdigest()
    with Context(0, 8) as c:
        for fx in golden["fixtures"]:
            iq = fixture_iq[fx["file"]]
            want = frozen[fx["file"]]
            assert zlib.crc32(c.to_mag(iq).data.astype("<u2").tobytes()) == want["mag_crc32"][0]
            dev = torch.from_numpy(iq).cuda()
            cand, ap = c.selftest_stage_lists(dev.data_ptr(), len(iq))
            assert (len(cand), sha(cand)) == (want["n_cand"], want["cand_sha256"])
            assert (len(ap), sha(ap)) == (want["n_ap"], want["ap_sha256"])
            # and the two stages in front of the candidates: check_preamble, then the 3.5 dB test
            pre, snr = c.selftest_gate_stages(dev.data_ptr(), len(iq))
            assert (len(pre), sha(pre)) == (want["n_preamble"], want["preamble_sha256"])
            assert (len(snr), sha(snr)) == (want["n_snr"], want["snr_sha256"])
            assert set(cand.tolist()) <= set(snr.tolist()) <= set(pre.tolist())
        n = 5 * 131072 + 7001
        iq = synth.make_iq(n, n_bursts=60, seed=4242, n_icao=7, df11_every=5)
        st = oracle_mod.stage_lists(iq)
        dev = torch.from_numpy(iq).cuda()
        cand, ap = c.selftest_stage_lists(dev.data_ptr(), n)
        assert cand.tolist() == st["cand"]
        assert ap.tolist() == st["ap"]
        pre, snr = c.selftest_gate_stages(dev.data_ptr(), n)
        assert pre.tolist() == st["preamble"] and snr.tolist() == st["snr"]
        # the context is as it was: a normal call still gives the oracle's frames
        c.icao_flush()
        assert_same(c.demod_iq(iq), oracle_mod.Oracle().demod_iq(iq)[0])


# ----------------------------------------------------------------------------- pipelined API
def test_submit_collect_matches_blocking_calls_and_orders_flushes(ctx, oracle_mod):
    """adsb_submit_iq_device / adsb_collect: passes in flight, results in submission
    order, icao_flush taking effect exactly between the passes it was called between."""
    import torch
    from dump1090_rs_amd._lib import AdsbError, ADSB_ERR_BUSY
    n = 8 * 131072
    icao = 0x4840D6
    body = bytes([0x20, 0x00, 0x05, 0x30])
    df4 = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")   # scores only once icao is known
    host = [synth.noise_numpy(n, seed=500 + k) for k in range(3)]
    synth.add_bursts(host[0], [synth.Burst(5 * (100000 * k + 777) + k, 21000, k, synth.df17_frame(icao, k))
                               for k in range(1, 9)])
    synth.add_bursts(host[1], [synth.Burst(5 * (90000 * k + 333) + k, 21000, k, df4) for k in range(1, 9)])
    synth.add_bursts(host[2], [synth.Burst(5 * (80000 * k + 555) + k, 21000, k, df4) for k in range(1, 9)])
    bufs = [torch.from_numpy(h).cuda() for h in host]
    torch.cuda.synchronize()
    # schedule: flush, A, B, flush, C, A   (B sees A's address, C starts clean, the 2nd A knows it)
    orc = oracle_mod.Oracle()
    want = []
    for k, flush in ((0, True), (1, False), (2, True), (0, False)):
        if flush:
            orc.icao_flush()
        want.append(orc.demod_iq(host[k])[0])
    # the DF4s decode only while the address is in the filter: B after A yes, C after the flush no
    assert sum(w["buffer"] == df4 for w in want[1]) >= 6 and want[2] == [] and len(want[0]) >= 8

    got = []
    ctx.icao_flush()
    ctx.submit_iq_device(bufs[0].data_ptr(), n)
    ctx.submit_iq_device(bufs[1].data_ptr(), n)
    assert ctx.pending() == 2
    spare = torch.zeros(4 * 131072, dtype=torch.int32, device="cuda")   # ADSB_MAX_IN_FLIGHT = 4
    torch.cuda.synchronize()   # (torch fills it on ITS stream: the library's streams do not wait for that one)
    ctx.submit_iq_device(spare.data_ptr(), 131072)
    ctx.submit_iq_device(spare.data_ptr(), 131072)
    with pytest.raises(AdsbError) as ei:                      # a fifth one does not fit
        ctx.submit_iq_device(bufs[2].data_ptr(), n)
    assert ei.value.status == ADSB_ERR_BUSY
    with pytest.raises(AdsbError):                            # blocking calls refuse while pending
        ctx.demod_iq_device(bufs[2].data_ptr(), n)
    got.append(ctx.collect())
    got.append(ctx.collect())
    assert ctx.collect() == [] and ctx.collect() == []        # the all-zero buffers
    ctx.icao_flush()
    ctx.submit_iq_device(bufs[2].data_ptr(), n)
    ctx.submit_iq_device(bufs[0].data_ptr(), n)
    got.append(ctx.collect())
    got.append(ctx.collect())
    assert ctx.pending() == 0
    for g, w in zip(got, want):
        assert_same(g, w)
    with pytest.raises(AdsbError):
        ctx.collect()                                         # nothing pending


def test_streaming_ring_pinned_double_buffering(hip_lib, oracle_mod):
    """BASELINE config 3: the host fills pinned ring buffers, H2D copies overlap passes;
    frames identical to one oracle stream over the same bytes."""
    from dump1090_rs_amd import Context
    per_slot, n_batches = 4 * 131072, 5
    iq = synth.make_iq(per_slot * n_batches - 7000, n_bursts=60, seed=8088)
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    got = []
    with Context(0, 4) as c:
        c.ring_create(per_slot)
        c.icao_flush()
        for b in range(n_batches):
            if c.pending() == 2:
                got += [(m.chunk + 4 * (b - 2), m) for m in c.collect()]
            part = iq[b * per_slot:(b + 1) * per_slot]
            buf = c.ring_acquire()
            assert buf.shape == (per_slot, 2)
            buf[: len(part)] = part
            c.ring_submit(len(part))
        k = n_batches - c.pending()
        while c.pending():
            got += [(m.chunk + 4 * k, m) for m in c.collect()]
            k += 1
    assert [(ch, m.j, m.try_phase, m.score, m.msg, m.signal_level) for ch, m in got] == \
        [(w["chunk"], w["j"], w["try_phase"], w["score"], w["msg"], w["signal_level"]) for w in want]


def test_sharded_capture_two_phases_equal_single_stream(hip_lib, oracle_mod):
    """one capture cut into two shards (two contexts standing in for two GPUs),
    adsb_shard_scan -> address exchange -> adsb_shard_finish -> one ordered replay, against the
    oracle over the whole capture; the DF4 of the second shard only decodes because the first
    shard's DF17 taught its address."""
    import torch
    from dump1090_rs_amd import Context, sharding
    from dump1090_rs_amd._lib import AdsbError, ADSB_ERR_BUSY
    from dump1090_rs_amd.context import replay_records
    from tests.test_sharding_gloo import _coupled_capture

    n = 6 * 131072 + 999
    iq, df4 = _coupled_capture(n)
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    dev = torch.from_numpy(iq).cuda()
    spans = [sharding.sample_range(n, 2, r) for r in range(2)]
    ctxs = [Context(max_chunks=4), Context(max_chunks=4)]
    try:
        ptr = lambda a: dev.data_ptr() + 4 * a
        learned = [c.shard_scan(ptr(a), b - a) for c, (a, b) in zip(ctxs, spans)]
        assert 0x4840D6 in learned[0].tolist() and 0x4840D6 not in learned[1].tolist()
        with pytest.raises(AdsbError) as busy:           # parked between the phases
            ctxs[0].demod_iq_device(ptr(0), 131072)
        assert busy.value.status == ADSB_ERR_BUSY
        union = np.unique(np.concatenate(learned))
        records = [c.shard_finish(union) for c in ctxs]
        merged = sharding.merge_records(records, [a // 131072 for a, _ in spans])
        got = replay_records(merged)
        assert_same(got, want)
        assert sorted(m.chunk for m in got if m.buffer() == df4) == [2, 4, 5]
        assert len(merged) < 2000  # of ~49 000 trials: only self-validating and matched address/parity ones
        # without the exchange the second shard drops its DF4s
        ctxs[1].icao_flush()
        ctxs[1].shard_scan(ptr(spans[1][0]), spans[1][1] - spans[1][0])
        alone = ctxs[1].shard_finish(np.zeros(0, np.uint32))
        assert not [m for m in replay_records(alone) if m.buffer() == df4]
        # a single shard is the ordinary path
        ctxs[0].icao_flush()
        one = sharding.demod_sharded(ctxs[0], ptr(0), 3 * 131072, 0)
        ctxs[0].icao_flush()
        assert_same(one, oracle_mod.Oracle().demod_iq(iq[: 3 * 131072])[0])
        assert_same(ctxs[0].demod_iq_device(ptr(0), 3 * 131072), oracle_mod.Oracle().demod_iq(iq[: 3 * 131072])[0])
    finally:
        for c in ctxs:
            c.close()


def test_shard_pipeline_overlaps_the_phases_of_consecutive_captures(hip_lib, oracle_mod):
    """sharding.ShardPipeline: finish + replay of capture i on a worker thread (context i % 2) while the
    main thread scans capture i + 1 on the other context.  Five different captures through it, each
    result (a flushed filter per capture) equal to the oracle's, in submission order."""
    import torch
    from dump1090_rs_amd import Context, sharding
    caps = []
    for k in range(5):
        n = (3 + k) * 131072 - 1234 * k
        iq = synth.make_iq(n, n_bursts=40 + 10 * k, seed=600 + k, n_icao=6, df11_every=4)
        caps.append((n, torch.from_numpy(iq).cuda(), oracle_mod.Oracle().demod_iq(iq)[0]))
    torch.cuda.synchronize()
    ctxs = [Context(0, 8), Context(0, 8)]
    try:
        pipe = sharding.ShardPipeline(ctxs)
        got = [pipe.submit(dev.data_ptr(), n, 0) for n, dev, _ in caps]
        assert got[0] is None and got[1] is None
        got = got[2:] + pipe.drain()
        pipe.close()
        assert len(got) == 5
        for g, (_, _, want) in zip(got, caps):
            assert_same(g, want)
        # the contexts are ordinary ones afterwards
        ctxs[0].icao_flush()
        assert_same(ctxs[0].demod_iq_device(caps[0][1].data_ptr(), caps[0][0]), caps[0][2])
    finally:
        for c in ctxs:
            c.close()


def test_shard_finish_with_a_huge_address_union_overflows_into_the_exact_fallback(hip_lib, oracle_mod):
    """Three quarters of the 24-bit address space handed to adsb_shard_finish: most of the
    shard's address/parity trials now match the superset bitmap, far more than the hit list
    holds, so the finish goes buffer by buffer through the worst-case lists (allocated on this
    first use).  The replay through the real filter still gives the single-stream result."""
    import torch
    from dump1090_rs_amd import Context
    from dump1090_rs_amd.context import replay_records

    n = 4 * 131072 - 4321
    iq = synth.make_iq(n, n_bursts=40, seed=99)
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    dev = torch.from_numpy(iq).cuda()
    rng = np.random.default_rng(3)
    union = np.flatnonzero(rng.random(1 << 24) < 0.75).astype(np.uint32)
    with Context(0, 4) as c:
        c.icao_flush()
        c.shard_scan(dev.data_ptr(), n)
        records = c.shard_finish(union)
        assert c.stats()["retries"] == 1 and len(records) > 4096 + 4 * 1024
        assert_same(replay_records(records), want)
        # the context is as good as new afterwards
        c.icao_flush()
        assert_same(c.demod_iq_device(dev.data_ptr(), n), want)


def test_feed_tool_prints_and_serves_reference_raw_lines(hip_lib, oracle_mod, golden, fixture_iq):
    """adsb_feed = the loop of dump1090_rs/src/main.rs:154-201 over a pipe: the three reference
    captures back to back on stdin (file order, im first) -> "*hex;" lines on stdout and on a
    raw TCP client, equal to the oracle's frames for the same stream (filter never flushed)."""
    import socket
    import subprocess
    import time
    from tests.conftest import ROOT, GOLDEN

    files = [GOLDEN / fx["file"] for fx in golden["fixtures"]]
    stream = np.concatenate([fixture_iq[fx["file"]] for fx in golden["fixtures"]])
    want, _ = oracle_mod.Oracle().demod_iq(stream)
    lines = [f"*{w['buffer'].hex()};" for w in want]
    assert lines[:5] == [f"*{h};" for h in golden["fixtures"][0]["frames"]]

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    feed = subprocess.Popen([str(ROOT / "dump1090_rs_amd" / "adsb_feed"), "--port", str(port), "--buffers", "2", "-"],
                            stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        client = None
        for _ in range(200):  # the listener is up before the first read of stdin
            try:
                client = socket.create_connection(("127.0.0.1", port), timeout=1)
                break
            except OSError:
                time.sleep(0.05)
        assert client is not None
        for f in files:
            feed.stdin.write(f.read_bytes())
        feed.stdin.close()
        out = feed.stdout.read().decode()
        err = feed.stderr.read().decode()
        assert feed.wait(timeout=120) == 0, err
        assert out.splitlines() == lines
        assert f"{len(stream)} samples, {len(lines)} frames in " in err
        client.settimeout(5)
        got = b""
        while True:
            part = client.recv(65536)
            if not part:
                break
            got += part
        assert got.decode() == "".join(l + "\n" for l in lines)
        client.close()
    finally:
        if feed.poll() is None:
            feed.kill()


def test_feed_tool_reads_a_capture_file_with_several_readers(hip_lib, oracle_mod, tmp_path):
    """A regular file is filled into the ring's slots by R threads that each pread() and swap their own part
    (adsb_feed.cpp: FileFill): a capture of 37 buffers and a ragged end, in file order and in memory order, with
    1 / 3 / 4 / 16 readers and slots of 1 / 5 / 64 buffers (whole slots, a last slot that the file ends in, a
    slot below the size at which the readers split) -- every run's lines equal to the oracle's stream."""
    import subprocess
    from tests.conftest import ROOT
    n = 37 * 131072 + 77777
    iq = synth.make_iq(n, n_bursts=500, seed=6061, n_icao=14, df11_every=4)
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    lines = [f"*{w['buffer'].hex()};" for w in want]
    assert len(lines) > 300
    file_order, mem_order = tmp_path / "cap.iq", tmp_path / "cap_mem.iq"
    file_order.write_bytes(np.ascontiguousarray(iq[:, ::-1]).tobytes())     # the capture format: im first
    mem_order.write_bytes(np.ascontiguousarray(iq).tobytes() + b"\x01\x02")   # (+ half a pair: dropped)
    feed = str(ROOT / "dump1090_rs_amd" / "adsb_feed")
    for path, extra in ((file_order, []), (mem_order, ["--mem-order"])):
        for readers, buffers in ((1, 5), (3, 5), (4, 64), (16, 5), (4, 1), (7, 64)):
            r = subprocess.run([feed, *extra, "--readers", str(readers), "--buffers", str(buffers), str(path)],
                               capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stderr
            assert r.stdout.splitlines() == lines, (extra, readers, buffers)
            assert f"{n} samples, {len(lines)} frames in " in r.stderr


def _start_feed(args):
    import socket
    import subprocess
    import time
    from tests.conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    feed = subprocess.Popen([str(ROOT / "dump1090_rs_amd" / "adsb_feed"), "--port", str(port), *args, "-"],
                            stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)

    def connect():
        for _ in range(200):  # the listener is up before the first read of stdin
            try:
                return socket.create_connection(("127.0.0.1", port), timeout=1)
            except OSError:
                time.sleep(0.05)
        raise AssertionError("adsb_feed is not listening")
    return feed, connect


def test_feed_tool_trickling_input_is_not_held_back_for_a_full_slot(hip_lib, oracle_mod):
    """A live pipe delivers a buffer every 55 ms; with 64-buffer slots the first frames would wait 3.5 s.
    adsb_feed submits the whole buffers it holds once the input has been idle for --latency-ms, and the
    buffer it has begun moves on to the next slot: one and a half buffers written, a pause -- the first
    buffer's frames are out before the rest is written -- then the rest; all of it equal to the oracle's
    stream cut at the same 131072-sample boundaries."""
    import os
    import select
    import time
    n = 4 * 131072 + 5000
    iq = synth.make_iq(n, n_bursts=120, seed=321, n_icao=9, df11_every=4)
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    lines = [f"*{w['buffer'].hex()};" for w in want]
    first = [f"*{w['buffer'].hex()};" for w in want if w["chunk"] == 0]
    assert len(first) >= 10 and len(lines) > len(first)
    raw = np.ascontiguousarray(iq[:, ::-1]).tobytes()       # the capture format: im first
    feed, _ = _start_feed(["--buffers", "64", "--latency-ms", "150"])
    try:
        cut = 6 * 131072                                      # one and a half buffers (4 bytes a sample)
        feed.stdin.write(raw[:cut])
        feed.stdin.flush()
        got = b""
        deadline = time.time() + 20
        while got.count(b"\n") < len(first) and time.time() < deadline:
            if select.select([feed.stdout], [], [], 0.2)[0]:
                got += os.read(feed.stdout.fileno(), 65536)
        assert got.decode().splitlines() == first, "the first buffer's frames did not come out while the input paused"
        feed.stdin.write(raw[cut:])
        feed.stdin.close()
        got += feed.stdout.read()
        err = feed.stderr.read().decode()
        assert feed.wait(timeout=120) == 0, err
        assert got.decode().splitlines() == lines
        assert f"{n} samples, {len(lines)} frames in " in err and " 0 short passes" not in err
    finally:
        if feed.poll() is None:
            feed.kill()


def test_feed_tool_steady_input_is_cut_into_short_passes_by_a_deadline(hip_lib, oracle_mod):
    """A live pipe is never idle: a writer that delivers an eighth of a buffer every 12 ms (faster than
    a 2.4 MSPS receiver, same shape) for ten buffers.  With 64-buffer slots nothing would come out before
    the input ends unless the slot is cut by a deadline -- --latency-ms after its first whole buffer was
    complete -- rather than by an idle timer (the advisor's round-3 finding): the first buffer's frames
    must be out while the writer is still writing, a client that connects meanwhile is accepted while the
    feed waits for input, and the whole stream equals the oracle's."""
    import os
    import select
    import threading
    import time
    n_buf = 10
    n = n_buf * 131072
    iq = synth.make_iq(n, n_bursts=300, seed=4242, n_icao=9, df11_every=4)
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    lines = [f"*{w['buffer'].hex()};" for w in want]
    first = [f"*{w['buffer'].hex()};" for w in want if w["chunk"] == 0]
    assert len(first) >= 10
    raw = np.ascontiguousarray(iq[:, ::-1]).tobytes()
    feed, connect = _start_feed(["--buffers", "64", "--latency-ms", "100"])
    done_at = {}

    def writer():
        piece = 131072 * 4 // 8
        for off in range(0, len(raw), piece):
            feed.stdin.write(raw[off:off + piece])
            feed.stdin.flush()
            time.sleep(0.012)
        done_at["t"] = time.time()
        time.sleep(0.5)        # (the client below connects and is accepted before the input ends)
        feed.stdin.close()
    try:
        th = threading.Thread(target=writer)
        th.start()
        got, first_seen = b"", None
        deadline = time.time() + 60
        while time.time() < deadline:
            if select.select([feed.stdout], [], [], 0.05)[0]:
                part = os.read(feed.stdout.fileno(), 1 << 16)
                if not part:
                    break
                got += part
                if first_seen is None and got.count(b"\n") >= len(first):
                    first_seen = time.time()
                    late = connect()          # accepted while the feed is busy reading
        th.join(timeout=30)
        err = feed.stderr.read().decode()
        assert feed.wait(timeout=120) == 0, err
        assert got.decode().splitlines() == lines
        assert first_seen is not None and first_seen < done_at["t"] - 0.3, "no output until the input was nearly over"
        assert f"{n} samples, {len(lines)} frames in " in err and " 0 short passes" not in err
        late.settimeout(5)
        tail = b""
        while True:
            part = late.recv(1 << 16)
            if not part:
                break
            tail += part
        late.close()
        # the late client got the frames of the passes that were finished after it connected: a suffix
        assert tail and ("\n".join(lines) + "\n").endswith(tail.decode())
    finally:
        if feed.poll() is None:
            feed.kill()


def test_feed_tool_drops_a_client_that_stops_reading_and_keeps_every_frame(hip_lib, oracle_mod):
    """One raw-TCP client never reads: once its socket buffer is full it is dropped (the reference drops
    a client whose write fails, main.rs:184-200) and neither the demodulation nor the other client waits
    for it.  Every pass yields more frames than the output array starts with (--out-cap 8):
    adsb_fetch_messages hands out the whole list, nothing is dropped from the reading client's stream."""
    import socket
    import threading
    n = 40 * 131072
    iq = synth.make_iq(n, n_bursts=16000, seed=77, n_icao=50)
    packed = synth.noise_numpy(131072, seed=78)                # one buffer of back-to-back short frames
    synth.add_bursts(packed, [synth.Burst(5 * (150 + 160 * q) + q % 5, 15000, q % 16, synth.df11_frame(0x480000 + q % 900))
                              for q in range(800)])
    iq[7 * 131072:8 * 131072] = packed
    want, _ = oracle_mod.Oracle().demod_iq(iq, cap=1 << 20)
    lines = [f"*{w['buffer'].hex()};" for w in want]
    text = "".join(l + "\n" for l in lines)
    rep = 6
    assert rep * len(text) > 3 << 20                            # more than the socket buffers of a client that never reads hold
    raw = np.ascontiguousarray(iq[:, ::-1]).tobytes()
    feed, connect = _start_feed(["--buffers", "1", "--quiet", "--out-cap", "8"])
    try:
        stalled = connect()
        stalled.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 4096)   # and it never calls recv
        reader = connect()
        got = []

        def pump():
            reader.settimeout(60)
            while True:
                part = reader.recv(1 << 16)
                if not part:
                    break
                got.append(part)
        th = threading.Thread(target=pump)
        th.start()
        # the stalled client keeps its connection open while ~4 MB more than any socket buffer go by
        for _ in range(rep):
            feed.stdin.write(raw)
        feed.stdin.close()
        err = feed.stderr.read().decode()
        assert feed.wait(timeout=300) == 0, err
        th.join(timeout=60)
        orc = oracle_mod.Oracle()
        all_lines = []
        for _ in range(rep):                                    # the same capture six times over, one filter
            all_lines += [f"*{w['buffer'].hex()};" for w in orc.demod_iq(iq, cap=1 << 20)[0]]
        assert b"".join(got).decode().splitlines() == all_lines
        assert " 1 clients dropped" in err and f"{rep * n} samples, {len(all_lines)} frames in " in err
        stalled.close()
        reader.close()
    finally:
        if feed.poll() is None:
            feed.kill()


def test_carry_over_mode_recovers_frames_across_buffer_edges(hip_lib, oracle_mod):
    """Opt-in extension, checked against the oracle's restatement of the same
    extension: lead-ins hold the preceding 326 samples, within a call, across calls (blocking,
    pipelined and ring), and the default mode is untouched."""
    import torch
    from dump1090_rs_amd import Context
    from oracle.binding import demod_iq_carry

    n = 5 * 131072 + 7777
    iq = synth.make_iq(n, n_bursts=60, seed=99, n_icao=10, df11_every=3)
    fr = synth.df17_frame(0xABCDEF, 12345)
    edges = [synth.Burst(5 * (131072 - 100), 20000, 3, fr), synth.Burst(5 * (2 * 131072 - 250), 20000, 5, fr),
             synth.Burst(5 * (3 * 131072 - 30), 20000, 7, fr), synth.Burst(5 * (4 * 131072 + 40000), 20000, 1, fr)]
    synth.add_bursts(iq, edges)
    plain, _ = oracle_mod.Oracle().demod_iq(iq)
    carry = np.zeros((326, 2), np.int16)
    want, _ = demod_iq_carry(oracle_mod.Oracle(), iq, carry)
    assert len([w for w in want if w["buffer"] == fr]) == 4 and len([w for w in plain if w["buffer"] == fr]) == 1

    c = Context(max_chunks=8)
    try:
        assert_same(c.demod_iq(iq), plain)                       # default: the reference's semantics
        c.set_carry_over(True)
        c.icao_flush()
        got = c.demod_iq(iq)
        assert_same(got, want)
        assert sorted((m.chunk, m.j) for m in got if m.buffer() == fr)[:3] == [(1, 225), (2, 75), (3, 295)]

        # the same stream in three calls cut at awkward places (one shorter than the carry)
        cuts = [0, 131072 + 500, 131072 + 700, n]
        orc, carry = oracle_mod.Oracle(), np.zeros((326, 2), np.int16)
        c.set_carry_over(True)                                   # restarts the stream
        c.icao_flush()
        for a, b in zip(cuts[:-1], cuts[1:]):
            w, _ = demod_iq_carry(orc, iq[a:b], carry)
            assert_same(c.demod_iq(iq[a:b]), w)

        # pipelined device-resident passes: pass i+1 starts from the end of pass i's input
        dev = torch.from_numpy(iq).cuda()
        orc, carry = oracle_mod.Oracle(), np.zeros((326, 2), np.int16)
        c.set_carry_over(True)
        c.icao_flush()
        cuts = [0, 2 * 131072, 3 * 131072 + 131000, n]
        wants = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            wants.append(demod_iq_carry(orc, iq[a:b], carry)[0])
        c.submit_iq_device(dev.data_ptr() + 4 * cuts[0], cuts[1] - cuts[0])
        c.submit_iq_device(dev.data_ptr() + 4 * cuts[1], cuts[2] - cuts[1])
        assert_same(c.collect(), wants[0])
        c.submit_iq_device(dev.data_ptr() + 4 * cuts[2], cuts[3] - cuts[2])
        assert_same(c.collect(), wants[1])
        assert_same(c.collect(), wants[2])

        c.set_carry_over(False)
        c.icao_flush()
        assert_same(c.demod_iq(iq), plain)
    finally:
        c.close()


ADVERSARIAL_PERIODS = [
    [18143, 6637, 18778, 14788, 3662, 8402, 2882, 16543],                            # 25 % of all j pass every gate
    [17914, 17500, 12559, 14370, 1698, 4482, 12099, 13742, 8850, 18749, 13607],      # 18 %
    [833, 13831, 15021, 13358, 19412, 4686, 19425, 17026, 15059, 11206, 19308, 3905, 13173, 6101, 4342, 10650, 7787],
    [4962, 12056, 7113, 852, 14360, 11334, 17932, 13963, 124, 14783, 17800, 508, 13806, 1459],
]


def test_adversarial_periodic_input_keeps_every_capacity_path_exact(hip_lib, oracle_mod):
    """Periodic magnitudes found by search: up to a quarter of all positions pass the preamble
    and both gates (noise: 1 %), i.e. 20-30x the density the kernel's wave-private regions,
    candidate regions, hit staging and AP segments are sized for.  Rounds, flushes and the
    per-chunk fallback must leave the result exactly the oracle's."""
    from dump1090_rs_amd import Context
    n = 6 * 131072
    iq = synth.make_iq(n, n_bursts=30, seed=5150)
    for k, amps in enumerate(ADVERSARIAL_PERIODS):
        a, b = k * 131072 + 40000, k * 131072 + 40000 + 60000 + 1000 * k
        tile = np.tile(np.array(amps, dtype=np.int16), (b - a) // len(amps) + 1)[: b - a]
        iq[a:b, 0] = tile
        iq[a:b, 1] = 0
    # one buffer that is periodic from end to end, and noise riding on a periodic carrier
    tile = np.tile(np.array(ADVERSARIAL_PERIODS[0], dtype=np.int16), 131072 // 8)
    iq[4 * 131072: 5 * 131072, 0] = tile
    iq[4 * 131072: 5 * 131072, 1] = 0
    iq[5 * 131072 + 1000: 5 * 131072 + 90000, 0] += np.tile(np.array(ADVERSARIAL_PERIODS[1], dtype=np.int16), 9000)[:89000] // 2
    orc = oracle_mod.Oracle()
    want, st = orc.demod_iq(iq, cap=1 << 20)
    assert st.quiet_pass > 60000                      # ~10 % of all positions were sliced
    for max_chunks in (8, 1):
        with Context(0, max_chunks) as c:
            c.icao_flush()
            got = c.demod_iq(iq, cap=1 << 20)
            assert_same(got, want)
            assert c.stats()["n_candidates"] == st.quiet_pass or c.stats()["retries"] > 0
    # the reference's two-call shape on the fully periodic buffer and on a mixed one: the
    # caller-supplied magnitudes go through the fast scan too, and through its fallback
    from dump1090_rs_amd import MagnitudeBuffer
    with Context(0, 1) as c:
        for k in (4, 5, 0):
            part = iq[k * 131072:(k + 1) * 131072 - (777 if k == 0 else 0)]
            data, length = oracle_mod.Oracle().to_mag(part)
            data[3:40] = np.arange(37, dtype=np.uint16) * 911      # a lead-in that is not zero
            want_m, _ = oracle_mod.Oracle().demodulate2400(data, length, cap=1 << 18)
            mb = MagnitudeBuffer()
            mb.data[:] = data
            mb.length = length
            c.icao_flush()
            assert_same(c.demodulate2400(mb, cap=1 << 18), want_m)
    # and in carry-over mode (the fallback kernel has its own lead-in path)
    from oracle.binding import demod_iq_carry
    carry = np.zeros((326, 2), np.int16)
    want_c, _ = demod_iq_carry(oracle_mod.Oracle(), iq, carry, cap=1 << 20)
    with Context(0, 1) as c:
        c.set_carry_over(True)
        c.icao_flush()
        assert_same(c.demod_iq(iq, cap=1 << 20), want_c)


def test_randomised_soak_all_entry_points(hip_lib, oracle_mod):
    """tests/fuzz_gpu.py: seeded random captures (ragged lengths, mixed frame kinds, bursts at
    buffer edges, saturation, adversarial patches) through every entry point -- blocking,
    pipelined, ring, two-phase shards -- in reference and carry-over semantics, each against
    the oracle.  (Longer runs: python tests/fuzz_gpu.py --cases 1000 --seed N.)"""
    import subprocess
    import sys
    from tests.conftest import ROOT
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "fuzz_gpu.py"), "--cases", "300", "--seed", "7", "--dense", "20", "--mixed", "40", "--multi", "30"],
                       capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    # (+ 20 dense pipelines: device-side order / score; + 40 mixed ones: small and large passes in flight
    # together over shared addresses -- the kind of case that found round 2's cross-stream ordering hole)
    assert "300 cases identical" in r.stdout and "dense_pipeline=20" in r.stdout and "mixed_pipeline=40" in r.stdout and "multi=30" in r.stdout


def test_soak_seed_that_found_the_overlapping_scans_hole(hip_lib, oracle_mod):
    """Round 5: a shard's scan first listed "the addresses whose bit in the superset bitmap it set".  The scans of
    consecutive captures overlap on two streams, so the LATER capture's scan could set an address's bit first, the
    earlier capture's list then lacked it, and the other device's second phase of that earlier capture missed the
    address/parity frames for it.  tests/fuzz_gpu.py --multi found it with this seed (sequence 45: two contexts of 20
    buffers, four captures in flight); the list now has a seen-bitmap of its own per shard.  The first 60 sequences of
    that seed, among them contexts of 17-20 buffers whose shards list their addresses while they scan, order their
    records on the device once a capture was dense, and take the list-overflow fallback."""
    import subprocess
    import sys
    from tests.conftest import ROOT
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "fuzz_gpu.py"), "--cases", "4", "--seed", "27182", "--dense", "0", "--mixed", "0", "--multi", "60",
                        "--no-multi-faults"],
                       capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "4 cases identical" in r.stdout and "multi=60" in r.stdout and "multi:device_ordered_shards=" in r.stdout


def test_four_host_threads_each_with_its_own_context(hip_lib, oracle_mod):
    """Contexts are independent streams (a context itself is not thread-safe): four host threads, each
    with its own context on the one GPU, run blocking and pipelined passes over different captures at
    the same time; every result against the oracle."""
    import threading
    import torch
    from dump1090_rs_amd import Context
    key = lambda m: (m.chunk, m.j, m.try_phase, m.score, m.msg, m.signal_level)
    okey = lambda w: (w["chunk"], w["j"], w["try_phase"], w["score"], w["msg"], w["signal_level"])
    jobs = []
    for tid in range(4):
        n = (12 + 6 * tid) * 131072 - 1000 * tid
        iq = synth.make_iq(n, n_bursts=150 + 40 * tid, seed=900 + tid, n_icao=20, df11_every=4)
        jobs.append((n, torch.from_numpy(iq).cuda(), [okey(w) for w in oracle_mod.Oracle().demod_iq(iq)[0]]))
    torch.cuda.synchronize()
    errors = []

    def work(tid):
        try:
            n, dev, want = jobs[tid]
            with Context(0, 12 + 6 * tid) as ctx:
                for it in range(12):
                    ctx.icao_flush()
                    if it % 2:
                        got = [key(m) for m in ctx.demod_iq_device(dev.data_ptr(), n)]
                    else:  # pipelined: the same capture twice, a flush in between
                        ctx.submit_iq_device(dev.data_ptr(), n)
                        ctx.icao_flush()
                        ctx.submit_iq_device(dev.data_ptr(), n)
                        got = [key(m) for m in ctx.collect()]
                        if [key(m) for m in ctx.collect()] != want:
                            errors.append((tid, it, "second"))
                    if got != want:
                        errors.append((tid, it, "first"))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors and not any(t.is_alive() for t in threads), errors


def test_sixteen_contexts_in_flight_at_once(hip_lib, oracle_mod, golden, fixture_iq):
    """Sixteen one-buffer contexts in one process (one per receiver, say), all with passes in flight at
    the same time, created and destroyed twice over (their internal streams come back from the
    process-wide pool): every context is its own stream -- six passes over its capture equal one oracle
    stream of six passes (the first gives the reference's frames, the later ones score against what the
    context's own filter has learned), whatever the fifteen others are doing."""
    import torch
    from dump1090_rs_amd import Context
    fxs = golden["fixtures"]
    devs = [torch.from_numpy(fixture_iq[fx["file"]]).cuda() for fx in fxs]
    torch.cuda.synchronize()
    want = []
    for fx in fxs:
        orc = oracle_mod.Oracle()
        orc.icao_flush()
        want.append([orc.demod_iq(fixture_iq[fx["file"]])[0] for _ in range(6)])
        assert [w["buffer"].hex() for w in want[-1][0]] == fx["frames"]
    for round_ in range(2):
        ctxs = [Context(0, 1) for _ in range(16)]
        try:
            for c in ctxs:
                c.icao_flush()
            for rep in range(3):
                for k, c in enumerate(ctxs):
                    c.submit_iq_device(devs[k % 3].data_ptr(), 131072)
                    c.submit_iq_device(devs[k % 3].data_ptr(), 131072)
                for k, c in enumerate(ctxs):
                    assert_same(c.collect(), want[k % 3][2 * rep])
                    assert_same(c.collect(), want[k % 3][2 * rep + 1])
        finally:
            for c in ctxs:
                c.close()


def test_one_gib_capture_of_2048_buffers(hip_lib, oracle_mod):
    """A capture four times the bench's (2048 buffers = 1 GiB of IQ, one blocking call) against the
    multi-threaded oracle, then twice more: from the second call on the context knows the stream is
    dense, so those passes are ordered (more than 1024 buckets) and scored on the device."""
    import torch
    from dump1090_rs_amd import Context
    chunks = 2048
    n = chunks * 131072 - 12345
    dev = synth.make_iq_torch(n, n_bursts=40 * chunks // 8, seed=4711, n_icao=300, df11_every=6, device="cuda")
    torch.cuda.synchronize()
    host = dev.cpu().numpy()
    want, _ = oracle_mod.Oracle().demod_iq(host, cap=1 << 22, threads=min(64, len(__import__("os").sched_getaffinity(0))))
    del host
    assert len(want) > 9000
    with Context(0, chunks) as c:
        for rep in range(3):
            c.icao_flush()
            assert_same(c.demod_iq_device(dev.data_ptr(), n, cap=1 << 22), want)
            assert c.stats()["retries"] == 0
        assert c._L.adsb_host_replays(c._h) == 1      # the first call only: the repeats were the device's


def test_inputs_longer_than_the_context_was_sized_for(hip_lib, oracle_mod):
    """A context created for 2 buffers: the blocking device call cuts a 5-buffer input into passes
    it can hold (same frames, filter carried along); one submission of more than 2 is refused."""
    import torch
    from dump1090_rs_amd import Context
    from dump1090_rs_amd._lib import AdsbError, ADSB_ERR_INVALID
    n = 5 * 131072 - 321
    iq = synth.make_iq(n, n_bursts=60, seed=606, n_icao=8, df11_every=5)
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    dev = torch.from_numpy(iq).cuda()
    with Context(0, 2) as c:
        c.icao_flush()
        assert_same(c.demod_iq_device(dev.data_ptr(), n), want)
        c.icao_flush()
        assert_same(c.demod_iq(iq), want)
        with pytest.raises(AdsbError) as e:
            c.submit_iq_device(dev.data_ptr(), 3 * 131072)
        assert e.value.status == ADSB_ERR_INVALID
        assert c.pending() == 0


def test_abi_misuse_is_refused_not_crashed(hip_lib):
    """Bad arguments and calls out of order come back as negative statuses (never an abort,
    never a device fault), and the context stays usable afterwards."""
    import torch
    from dump1090_rs_amd._lib import AdsbMsg
    L = hip_lib
    h = C.c_void_p()
    assert L.adsb_create(C.byref(h), 0, 2) == 0
    n = C.c_size_t()
    out = (AdsbMsg * 16)()
    dev = torch.zeros((3 * 131072, 2), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ptr = dev.data_ptr()
    INVALID, BUSY, CAPACITY = -1, -7, -5
    assert L.adsb_demod_iq_device(h, None, 100, out, 16, C.byref(n)) == INVALID
    assert L.adsb_demod_iq_device(h, C.c_void_p(ptr + 4), 1000, out, 16, C.byref(n)) == INVALID     # not 16-byte aligned
    assert L.adsb_demod_iq_device(h, C.c_void_p(ptr), 1000, None, 16, C.byref(n)) == INVALID
    assert L.adsb_demod_iq_device(h, C.c_void_p(ptr), 0, out, 16, C.byref(n)) == 0 and n.value == 0  # empty stream: no frames
    assert L.adsb_collect(h, out, 16, C.byref(n)) == INVALID                                          # nothing pending
    assert L.adsb_submit_iq_device(h, C.c_void_p(ptr), 3 * 131072) == INVALID                         # > max_chunks
    assert L.adsb_submit_iq_device(h, C.c_void_p(ptr), 0) == INVALID
    assert L.adsb_shard_finish(h, None, 0, None, 0, C.byref(n)) == INVALID                            # no shard parked
    assert L.adsb_ring_submit(h, 100) == INVALID                                                      # no ring yet
    assert L.adsb_ring_create(h, 3 * 131072) == INVALID                                               # > max_chunks
    assert L.adsb_ring_create(h, 131072) == 0
    assert L.adsb_ring_create(h, 131072) == INVALID                                                   # once only
    assert L.adsb_ring_submit(h, 131073) == INVALID
    # adsb_max_in_flight passes in flight (8: this context was created for two buffers per pass; 4 for a large
    # one), one more is refused; blocking calls are refused while passes are pending
    depth = L.adsb_max_in_flight(h)
    assert depth == 8
    for _ in range(depth):
        assert L.adsb_submit_iq_device(h, C.c_void_p(ptr), 131072) == 0
    assert L.adsb_submit_iq_device(h, C.c_void_p(ptr), 131072) == BUSY
    assert L.adsb_demod_iq_device(h, C.c_void_p(ptr), 1000, out, 16, C.byref(n)) == BUSY
    assert L.adsb_set_carry_over(h, 1) == BUSY
    assert L.adsb_pending(h) == depth
    for left in range(depth - 1, -1, -1):
        assert L.adsb_collect(h, out, 16, C.byref(n)) == 0 and n.value == 0 and L.adsb_pending(h) == left
    # too small an output array: the count comes back, the first `cap` entries are written
    iq = synth.make_iq(131072, n_bursts=20, seed=3)
    t = torch.from_numpy(iq).cuda()
    assert L.adsb_icao_flush(h) == 0
    assert L.adsb_demod_iq_device(h, C.c_void_p(t.data_ptr()), 131072, out, 2, C.byref(n)) == CAPACITY and n.value > 2
    need = n.value
    big = (AdsbMsg * need)()
    assert L.adsb_icao_flush(h) == 0
    assert L.adsb_demod_iq_device(h, C.c_void_p(t.data_ptr()), 131072, big, need, C.byref(n)) == 0 and n.value == need
    assert bytes(big[0].msg) == bytes(out[0].msg) and bytes(big[1].msg) == bytes(out[1].msg)
    L.adsb_destroy(h)
    L.adsb_destroy(None)


def test_bench_line_keeps_the_driver_contract(hip_lib, oracle_mod):
    """`python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line with the keys the driver reads:
    the metric / value / unit block, `roofline` (bound, achieved, peak, unit, frac, traffic) for the
    dominant kernel, `cpu_baseline` (value, unit, cores, kind, sample) and the parity flag -- checked on a
    small workload (32 buffers, no ramp; of the extra legs only config 4's, on a capture of 96 buffers:
    `also.config4_sharded_capture` = the capture through adsb_multi_* on one context and on eight, sparse and busy sky,
    each against the threaded oracle over the whole capture)."""
    import json
    import subprocess
    import sys
    from tests.conftest import ROOT
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--chunks", "32",
                        "--ramp-ms", "0", "--also-only", "config4", "--capture-chunks", "96"], capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"] == "IQ Msamples/s demodulated" and d["unit"] == "Msamples/s" and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"], d["scaling"], d["data"], d["vs_baseline"]) == (1, 4, 2, "weak", "synthetic", None)
    assert d["value"] > 0 and abs(d["value"] - 32 * 131072 / d["ms_per_step"] / 1e3) / d["value"] < 0.01
    assert "workload" in d["config"] and "model" not in d["config"] and isinstance(d["dtype"], str)
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"] == "k_scan_fast"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["achieved"] > 0 and "traffic" in rf
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["kernel_avg_ms"] * 1e-3) / 1e9) / rf["achieved"] < 0.01
    assert rf["algorithmic_bytes_per_launch"] == 4 * 32 * 131072 and rf["sustained"]["frac_over_steps"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "Msamples/s" and cb["value"] > 0 and cb["sample"]
    assert d["parity_checked"] is True and d["parity_frames"] > 0 and d["per_rank_ms_per_step"] and d["world_size_seen"] == 1
    c4 = d["also"]["config4_sharded_capture"]
    assert set(d["also"]) == {"config4_sharded_capture"} and c4["parity_checked"] is True
    assert [(r["sky"], r["shards"]) for r in c4["runs"] if r["shards"] in (1, 8)] == [("sparse", 1), ("sparse", 8), ("busy_sky", 1), ("busy_sky", 8)]
    for r in c4["runs"]:
        assert r["parity_checked"] is True and r["parity_frames"] > 0 and r["value"] > 0 and r["steps"] == 4
        assert r["roofline"]["bound"] == "hbm" and r["roofline"]["unit"] == "GB/s" and 0 < r["roofline"]["frac"] < 1
        assert abs(r["roofline"]["achieved"] - 4 * 96 * 131072 / (r["ms_per_step"] * 1e-3) / 1e9) / r["roofline"]["achieved"] < 0.01
