"""The reference's own call shape -- one read, one demodulation of 131072 samples at a time
(dump1090_rs/src/main.rs:161-167, benches/demod_benchmark.rs:10-11) -- goes through the library as ONE
launch per pass (k_scan_fast<.., FUSED>: scan, match and records in one kernel, no events, the ring slot
read in place from pinned host memory).  What that form must get right, against the CPU oracle:

* the pinned ring at one buffer per slot (BASELINE config 3 as written), several passes in flight;
* an address learned in pass k that a frame in pass k + 1 needs, while k + 1 was launched before k was
  collected (the host redoes k + 1: adsb_host_rematches);
* inside one pass: an address/parity frame late in the buffer for an address taught early in it (its
  workgroup may match before the teaching workgroup has set the bit: the last workgroup looks again),
  and the reverse (must NOT decode);
* every entry point at 1 .. 17 buffers (16 is the largest one-launch pass, 17 the smallest of three).
"""
import numpy as np
import pytest

from dump1090_rs_amd import synth

pytestmark = pytest.mark.gpu

CHUNK = 131072


def ap_frame(first_bytes: bytes, icao: int) -> bytes:
    return first_bytes + (synth.crc24(first_bytes) ^ icao).to_bytes(3, "big")


def key(m):
    return (m.chunk, m.j, m.try_phase, m.score, m.msg, m.signal_level)


def want_key(w, chunk_offset=0):
    return (w["chunk"] + chunk_offset, w["j"], w["try_phase"], w["score"], w["msg"], w["signal_level"])


def rematches(c) -> int:
    return int(c._L.adsb_host_rematches(c._h))


def ring_stream(c, iq, per_slot, depth, flush_before=()):
    """`iq` through the ring in slots of `per_slot` samples, `depth` passes in flight; (slot, message) pairs."""
    n_slots = (len(iq) + per_slot - 1) // per_slot
    got, collected = [], 0
    for b in range(n_slots):
        if c.pending() == depth:
            got += [(collected, m) for m in c.collect()]
            collected += 1
        if b in flush_before:
            c.icao_flush()
        part = iq[b * per_slot:(b + 1) * per_slot]
        buf = c.ring_acquire()
        buf[: len(part)] = part
        c.ring_submit(len(part))
    while c.pending():
        got += [(collected, m) for m in c.collect()]
        collected += 1
    assert collected == n_slots
    return got


@pytest.mark.parametrize("depth", [1, 3, 4, 8])
def test_ring_of_512kb_slots_equals_one_oracle_stream(hip_lib, oracle_mod, depth):
    """BASELINE config 3 as written: 96 slots of ONE 131072-sample buffer each (the last one ragged),
    `depth` passes in flight, against one oracle stream over the same bytes.  The stream keeps teaching
    the filter new addresses (a pool of 40) and holds address/parity frames for them right behind."""
    from dump1090_rs_amd import Context
    n = 95 * CHUNK + 70001
    iq = synth.make_iq(n, n_bursts=500, seed=9400 + depth, n_icao=40, df11_every=5)
    icao = 0x4B1A2C
    df4 = ap_frame(bytes([0x20, 0x00, 0x05, 0x30]), icao)
    df20 = ap_frame(bytes([0xA0, 0x00, 0x05, 0x30, 1, 2, 3, 4, 5, 6, 7]), icao)
    at = lambda chunk, j: 5 * (chunk * CHUNK + j)
    synth.add_bursts(iq, [synth.Burst(at(30, 50000), 22000, 1, df4),                       # too early
                          synth.Burst(at(31, 100000) + 2, 22000, 2, synth.df17_frame(icao, 7)),
                          synth.Burst(at(32, 300) + 1, 22000, 3, df4),                     # the very next pass
                          synth.Burst(at(33, 60000) + 3, 22000, 4, df20),
                          synth.Burst(at(35, 1000), 22000, 5, df4)])
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    with Context(0, 1) as c:
        c.ring_create(CHUNK)
        c.icao_flush()
        got = ring_stream(c, iq, CHUNK, depth)
        assert [(s,) + key(m)[1:] for s, m in got] == [want_key(w) for w in want]
        frames = lambda f: sorted({s for s, m in got if m.buffer() == f})
        assert frames(df4) == [32, 35] and frames(df20) == [33]
        if depth > 1:   # pass 32 was launched while 31 was in flight: it had to be matched again
            assert rematches(c) >= 1
        else:
            assert rematches(c) == 0
        # the same stream once more without a flush: everything is known now, nothing is redone
        before = rematches(c)
        again = ring_stream(c, iq[: 40 * CHUNK], CHUNK, depth)
        orc = oracle_mod.Oracle()
        orc.demod_iq(iq)
        want2, _ = orc.demod_iq(iq[: 40 * CHUNK])
        assert [(s,) + key(m)[1:] for s, m in again] == [want_key(w) for w in want2]
        assert frames(df4) == [32, 35]
        assert sorted({s for s, m in again if m.buffer() == df4}) == [30, 32, 35]    # 30 decodes now
        assert rematches(c) == before


@pytest.mark.parametrize("per_slot,depth", [(3, 2), (3, 8), (6, 4), (16, 8), (16, 1), (20, 4), (20, 2)])
def test_ring_slots_copied_in_front_of_their_pass(hip_lib, oracle_mod, per_slot, depth):
    """Slots of three buffers and more reach their pass through the copy engine, queued on the pass's own scan
    stream in front of its first launch (no event), unless nothing else is in flight (then the first one is read
    in place): 16 is the largest one-launch slot, 20 takes three launches.  A stream that keeps teaching the
    filter, a flush in the middle, the last slot ragged; against one oracle stream."""
    from dump1090_rs_amd import Context
    n_slots = 13
    n = (n_slots - 1) * per_slot * CHUNK + 31007
    iq = synth.make_iq(n, n_bursts=30 * n_slots * per_slot, seed=5100 + per_slot + depth, n_icao=30, df11_every=4)
    flush_before = {0, 7}
    orc = oracle_mod.Oracle()
    want = []
    for b in range(n_slots):
        if b in flush_before:
            orc.icao_flush()
        want += [(b,) + want_key(w) for w in orc.demod_iq(iq[b * per_slot * CHUNK:(b + 1) * per_slot * CHUNK])[0]]
    assert len(want) > 20 * n_slots
    with Context(0, per_slot) as c:
        c.ring_create(per_slot * CHUNK)
        for rep in range(2):
            got = ring_stream(c, iq, per_slot * CHUNK, depth, flush_before)
            assert [(s,) + key(m) for s, m in got] == want


@pytest.mark.parametrize("n_buf", [1, 8])
def test_address_taught_early_in_a_buffer_reaches_a_frame_late_in_it_and_not_the_reverse(hip_lib, oracle_mod, n_buf):
    """One launch, 17 workgroups per buffer that finish in any order: a DF17 in one tile and address/parity
    frames for its address in the tiles behind it must decode (score 1000), the mirrored buffer -- frames
    first, DF17 last -- must not; both a hundred times over, blocking and pipelined.  (Eight buffers: 136
    workgroups -- a multiple of eight, the grid shape the large passes place by XCD -- with the pair in the
    last buffer, tiles next to each other included.)"""
    import torch
    from dump1090_rs_amd import Context
    icao = 0x3C6589
    df4 = ap_frame(bytes([0x20, 0x00, 0x05, 0x30]), icao)
    df21 = ap_frame(bytes([0xA8, 0x00, 0x05, 0x30, 9, 8, 7, 6, 5, 4, 3]), icao)
    base = 5 * (n_buf - 1) * CHUNK
    fwd = synth.noise_numpy(n_buf * CHUNK, seed=77)
    synth.add_bursts(fwd, [synth.Burst(base + 5 * 700 + 2, 23000, 1, synth.df17_frame(icao, 1))] +
                     [synth.Burst(base + 5 * (9000 + 16000 * q) + q % 5, 21000, q, df4 if q % 2 else df21) for q in range(7)])
    rev = synth.noise_numpy(n_buf * CHUNK, seed=78)
    synth.add_bursts(rev, [synth.Burst(base + 5 * (1000 + 16000 * q) + q % 5, 21000, q, df4 if q % 2 else df21) for q in range(7)] +
                     [synth.Burst(base + 5 * 126000 + 2, 23000, 1, synth.df17_frame(icao, 1))])
    orc = oracle_mod.Oracle()
    w_fwd, _ = orc.demod_iq(fwd)
    orc.icao_flush()
    w_rev, _ = orc.demod_iq(rev)
    ap = lambda ws: [w for w in ws if w["buffer"] in (df4, df21)]
    assert len(ap(w_fwd)) >= 7 and all(w["score"] == 1000 for w in ap(w_fwd)) and not ap(w_rev)
    d_fwd, d_rev = torch.from_numpy(fwd).cuda(), torch.from_numpy(rev).cuda()
    torch.cuda.synchronize()
    CHUNKS = n_buf * CHUNK
    with Context(0, n_buf) as c:
        # polls = 0: no workgroup waits for the tiles before it (adsb_selftest_set_order_polls), so whenever the
        # pass learns an address its last workgroup looks at every list once more -- the fallback, on purpose
        for polls in (200, 0):
            assert c._L.adsb_selftest_set_order_polls(c._h, polls) == 0
            for rep in range(100):
                c.icao_flush()
                assert [key(m) for m in c.demod_iq_device(d_fwd.data_ptr(), CHUNKS)] == [want_key(w) for w in w_fwd]
                c.icao_flush()
                assert [key(m) for m in c.demod_iq_device(d_rev.data_ptr(), CHUNKS)] == [want_key(w) for w in w_rev]
            for rep in range(50):   # pipelined, a flush in front of each: four one-launch passes in flight
                for d in (d_fwd, d_rev, d_fwd, d_rev):
                    c.icao_flush()
                    c.submit_iq_device(d.data_ptr(), CHUNKS)
                for w in (w_fwd, w_rev, w_fwd, w_rev):
                    assert [key(m) for m in c.collect()] == [want_key(x) for x in w]
            # caller-supplied magnitudes take the same one-launch form (adsb_demodulate2400)
            if n_buf == 1:
                c.icao_flush()
                assert [key(m) for m in c.demodulate2400(c.to_mag(fwd))] == [want_key(w) for w in w_fwd]


@pytest.mark.parametrize("n_chunks,cut", [(1, 0), (1, 50000), (2, 131071), (5, 4321), (16, 0), (16, 99), (17, 0), (17, 5000)])
def test_every_entry_point_at_the_sizes_around_the_one_launch_limit(hip_lib, oracle_mod, n_chunks, cut):
    """Host pointer (read in place from pinned staging), device pointer, submit / collect, the ring: a
    capture of n_chunks buffers (the last cut short) through each, same frames as the oracle."""
    import torch
    from dump1090_rs_amd import Context
    n = n_chunks * CHUNK - cut
    iq = synth.make_iq(n, n_bursts=20 * n_chunks, seed=300 + 7 * n_chunks + cut % 11, n_icao=8, df11_every=3)
    want = [want_key(w) for w in oracle_mod.Oracle().demod_iq(iq)[0]]
    dev = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    with Context(0, n_chunks) as c:
        c.icao_flush()
        assert [key(m) for m in c.demod_iq(iq)] == want
        c.icao_flush()
        assert [key(m) for m in c.demod_iq_device(dev.data_ptr(), n)] == want
        c.icao_flush()
        c.submit_iq_device(dev.data_ptr(), n)
        c.icao_flush()
        c.submit_iq_device(dev.data_ptr(), n)
        assert [key(m) for m in c.collect()] == want and [key(m) for m in c.collect()] == want
        c.ring_create(n_chunks * CHUNK)
        c.icao_flush()
        got = ring_stream(c, iq, n_chunks * CHUNK, 2)
        assert [key(m) for _, m in got] == want
    # a context sized for one buffer takes the same capture buffer by buffer (the filter persists)
    with Context(0, 1) as c:
        c.icao_flush()
        assert [key(m) for m in c.demod_iq(iq)] == want


def test_one_launch_passes_between_long_passes_and_flushes(hip_lib, oracle_mod):
    """One-launch passes, three-launch passes and icao_flush interleaved in one pipeline over captures
    that share addresses: every ordering edge between the two forms (an event recorded on the spot where a
    long pass has to wait for a one-launch pass, the bitmap a flush retires, the unsynchronised-pass
    rule) against one oracle stream."""
    import torch
    from dump1090_rs_amd import Context
    rng = np.random.default_rng(2024)
    sizes = [1, 24, 2, 1, 40, 1, 16, 3, 30, 1, 1, 20, 5, 1]
    caps = []
    for k, nc in enumerate(sizes):
        n = nc * CHUNK - int(rng.integers(0, 3000))
        caps.append(synth.make_iq(n, n_bursts=15 * nc, seed=7000 + k, n_icao=5, df11_every=3))
    flush_before = {0, 4, 5, 9, 12}
    orc = oracle_mod.Oracle()
    want = []
    for k, iq in enumerate(caps):
        if k in flush_before:
            orc.icao_flush()
        want.append([want_key(w) for w in orc.demod_iq(iq)[0]])
    devs = [torch.from_numpy(iq).cuda() for iq in caps]
    torch.cuda.synchronize()
    with Context(0, 40) as c:
        for rep in range(6):
            depth = 1 + rep % 4
            got, k_out = [], 0
            for k, d in enumerate(devs):
                if c.pending() == depth:
                    got.append([key(m) for m in c.collect()])
                if k in flush_before:
                    c.icao_flush()
                c.submit_iq_device(d.data_ptr(), len(caps[k]))
            while c.pending():
                got.append([key(m) for m in c.collect()])
            assert got == want, (rep, [i for i, (g, w) in enumerate(zip(got, want)) if g != w])


@pytest.mark.parametrize("n_buf", [1, 8])
def test_a_buffer_packed_with_frames_takes_every_record_path(hip_lib, oracle_mod, n_buf):
    """Frames back to back (one every 300 samples: ~25 a tile, each decoding at several positions and phases):
    more hits in a tile than its staging holds (32: the rest goes through the hit list and the record
    builder at the end, behind the records built in place), address/parity frames matched inline for
    addresses taught a few frames earlier; eight such buffers overflow the pass's record capacity, and
    the pass is redone buffer by buffer.  One-launch passes throughout, against the oracle."""
    from dump1090_rs_amd import Context
    n = n_buf * CHUNK
    iq = synth.noise_numpy(n, seed=4242 + n_buf)
    bursts = []
    for k in range(n // 300 - 2):
        icao = 0x400000 + (k % 37) * 0x101
        kind = k % 4
        frame = (synth.df17_frame(icao, k) if kind in (0, 1) else synth.df11_frame(icao) if kind == 2
                 else ap_frame(bytes([0x20, 0x00, 0x05, 0x30]), icao))
        bursts.append(synth.Burst(5 * (300 * k + 40) + k % 5, 18000 + 500 * (k % 9), k, frame))
    synth.add_bursts(iq, bursts)
    want, _ = oracle_mod.Oracle().demod_iq(iq, cap=1 << 20)
    assert len(want) > 300 * n_buf
    with Context(0, n_buf) as c:
        for rep in range(3):
            c.icao_flush()
            got = c.demod_iq(iq, cap=1 << 20)
            assert [key(m) for m in got] == [want_key(w) for w in want]
            st = c.stats()
            assert st["n_records"] >= len(want)
            if n_buf == 1:
                assert st["retries"] == 0 and st["n_records"] > 2 * 32 * 17   # ~94 hits a tile: staging holds 32
            if n_buf == 8:
                assert st["retries"] > 0       # (4096 + 8 * 1024 records do not hold this pass: the fallback ran)


def test_samples_in_a_registered_host_buffer_are_read_in_place(hip_lib, oracle_mod):
    """adsb_host_register: the caller's own buffer (the reference's main loop reads the SDR into one Vec,
    main.rs:154-167) pinned and mapped once; adsb_demod_iq on samples inside it is one launch that reads them
    where they are -- whole, a window of it at a 16-byte aligned offset, rewritten between calls; a window at an
    odd offset, samples outside it, and more buffers than one launch takes go the usual way.  Same frames."""
    from dump1090_rs_amd import Context
    n = 6 * CHUNK
    big = np.zeros((n + 64, 2), dtype=np.int16)
    iq = synth.make_iq(n, n_bursts=200, seed=6060, n_icao=12, df11_every=4)
    other = synth.make_iq(2 * CHUNK, n_bursts=60, seed=6061, n_icao=12, df11_every=4)
    big[:n] = iq
    orc = oracle_mod.Oracle()
    with Context(0, 6) as c:
        c.host_register(big)
        with pytest.raises(Exception):
            c.host_register(big[100:200])                       # overlaps
        for a, b in ((0, n), (4 * 1000, 4 * 1000 + 2 * CHUNK + 777), (3, CHUNK + 3), (8, 8 + CHUNK)):
            orc.icao_flush()
            c.icao_flush()
            want = [want_key(w) for w in orc.demod_iq(big[a:b])[0]]
            assert [key(m) for m in c.demod_iq(big[a:b])] == want
            assert c.stats()["n_samples"] == b - a
        # the host rewrites its buffer between calls, as an SDR read does
        for rep in range(20):
            src = other if rep % 2 else iq[CHUNK:3 * CHUNK]
            big[:2 * CHUNK] = src
            orc.icao_flush()
            c.icao_flush()
            assert [key(m) for m in c.demod_iq(big[:2 * CHUNK])] == [want_key(w) for w in orc.demod_iq(src)[0]]
        # a buffer that is not registered, and the registered one after unregistering
        orc.icao_flush()
        c.icao_flush()
        assert [key(m) for m in c.demod_iq(other)] == [want_key(w) for w in orc.demod_iq(other)[0]]
        c.host_unregister(big)
        with pytest.raises(Exception):
            c.host_unregister(big)
        orc.icao_flush()
        c.icao_flush()
        assert [key(m) for m in c.demod_iq(big[:n])] == [want_key(w) for w in orc.demod_iq(big[:n])[0]]


@pytest.mark.parametrize("depth", [1, 2, 8])
def test_icao_flush_before_every_pipelined_pass_is_exact_and_redoes_nothing(hip_lib, oracle_mod, depth):
    """The reference's benchmark shape (benches/demod_benchmark.rs:9-11: icao_flush, then one buffer) pipelined:
    every pass opens a new epoch of the filter on the next (folded) bitmap of the rotation and clears it itself, waits
    for no other pass, and no pass is redone for what a pass from BEFORE its flush taught the filter.  200 passes
    over 12 distinct buffers that all teach the same aircraft; each equal to the oracle behind a flush."""
    import torch
    from dump1090_rs_amd import Context
    bufs = [synth.make_iq(CHUNK - (977 if k == 5 else 0), n_bursts=14, seed=31000 + k, n_icao=4, df11_every=3) for k in range(12)]
    wants = []
    for b in bufs:
        orc = oracle_mod.Oracle()
        wants.append([want_key(w) for w in orc.demod_iq(b)[0]])
    assert all(len(w) >= 8 for w in wants)
    devs = [torch.from_numpy(b).cuda() for b in bufs]
    torch.cuda.synchronize()
    with Context(0, 1) as c:
        got, sub = [], 0
        for i in range(200):
            if c.pending() == depth:
                got.append([key(m) for m in c.collect()])
            c.icao_flush()
            c.submit_iq_device(devs[i % 12].data_ptr(), len(bufs[i % 12]))
            sub += 1
        while c.pending():
            got.append([key(m) for m in c.collect()])
        assert len(got) == 200
        for i, g in enumerate(got):
            assert g == wants[i % 12], f"pass {i}"
        assert rematches(c) == 0
        # the ring the same way (slots read in place)
        c.ring_create(CHUNK)
        got = []
        for i in range(64):
            if c.pending() == depth:
                got.append([key(m) for m in c.collect()])
            c.icao_flush()
            buf = c.ring_acquire()
            buf[: len(bufs[i % 12])] = bufs[i % 12]
            c.ring_submit(len(bufs[i % 12]))
        while c.pending():
            got.append([key(m) for m in c.collect()])
        for i, g in enumerate(got):
            assert g == wants[i % 12], f"ring pass {i}"
        assert rematches(c) == 0


@pytest.mark.parametrize("depth", [2, 4, 8])
def test_passes_right_behind_a_flush_share_the_bitmap_the_flushed_pass_clears(hip_lib, oracle_mod, depth):
    """icao_flush, then a pass that teaches an address, then -- launched while that one is still in flight, on other
    streams -- passes with address/parity frames for it, for an address from BEFORE the flush (must not decode any
    more) and one that teaches another address a later pass needs: the passes behind the flush share the bitmap the
    flushed pass clears when it starts, so they are ordered behind it; against one oracle stream, over and over."""
    import torch
    from dump1090_rs_amd import Context
    old, a, b = 0x3C6589, 0x4840D6, 0x7C1B2A
    df4 = lambda icao, k: ap_frame(bytes([0x20, 0x00, 0x05, 0x30 + k]), icao)
    at = lambda j, ph: 5 * j + ph
    bufs = []
    for k in range(6):
        iq = synth.make_iq(CHUNK, n_bursts=6, seed=32000 + k, n_icao=3)
        if k == 0:      # the pass behind the flush: teaches a (late in the buffer), holds a frame for `old`
            synth.add_bursts(iq, [synth.Burst(at(120000, 2), 22000, 1, synth.df17_frame(a, 3)), synth.Burst(at(4000, 1), 22000, 2, df4(old, 0))])
        if k == 1:      # needs a; teaches b
            synth.add_bursts(iq, [synth.Burst(at(3000, 3), 22000, 3, df4(a, 1)), synth.Burst(at(90000, 0), 22000, 4, synth.df11_frame(b))])
        if k == 2:      # needs a and b, and `old` again
            synth.add_bursts(iq, [synth.Burst(at(2700, 1), 22000, 5, df4(a, 2)), synth.Burst(at(5000, 4), 22000, 6, df4(b, 2)),
                                  synth.Burst(at(60000, 2), 22000, 7, df4(old, 2))])
        if k == 5:      # teaches `old` again for the next round's first frames (which come after a flush: must not count)
            synth.add_bursts(iq, [synth.Burst(at(30000, 3), 22000, 8, synth.df17_frame(old, 9))])
        bufs.append(iq)
    orc = oracle_mod.Oracle()
    wants = []
    for rnd in range(8):
        for k, iq in enumerate(bufs):
            if k == 0:
                orc.icao_flush()
            wants.append([want_key(w) for w in orc.demod_iq(iq)[0]])
    frames_of = lambda ws, f: sum(1 for w in ws if bytes(w[4][:7]) == f)
    assert frames_of(wants[6 + 1], df4(a, 1)) >= 1 and frames_of(wants[6 + 2], df4(b, 2)) >= 1
    assert frames_of(wants[6 + 0], df4(old, 0)) == 0 and frames_of(wants[6 + 2], df4(old, 2)) == 0   # flushed away
    devs = [torch.from_numpy(x).cuda() for x in bufs]
    torch.cuda.synchronize()
    with Context(0, 1) as c:
        got = []
        for i in range(len(wants)):
            if c.pending() == depth:
                got.append([key(m) for m in c.collect()])
            if i % 6 == 0:
                c.icao_flush()
            c.submit_iq_device(devs[i % 6].data_ptr(), CHUNK)
        while c.pending():
            got.append([key(m) for m in c.collect()])
        for i, (g, w) in enumerate(zip(got, wants)):
            assert g == w, f"pass {i}"
