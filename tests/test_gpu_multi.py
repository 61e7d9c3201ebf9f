"""adsb_multi_*: ONE process, N GPUs, one filter -- BASELINE config 4 behind the C ABI.

The reference is one loop with one process-global ICAO filter (dump1090_rs/src/main.rs:154-167,
src/icao_filter.rs:8-9).  An adsb_multi cuts a capture into contiguous ranges of 131072-sample buffers, one per
device, runs the two shard phases on a thread per device, unites the learned addresses in memory and replays all
shards' trial records once through ONE filter.  On the one-GPU test box `devices = [0] * 8` (eight contexts,
eight threads, one device) stands in for eight GPUs; wherever the box has several devices the same tests run over
`range(device_count)` as well.  Every result is compared with the CPU oracle over the whole capture AND with the
single-stream GPU result.
"""
import ctypes as C
import subprocess

import numpy as np
import pytest

from dump1090_rs_amd import synth
from tests.conftest import GOLDEN, ROOT
from tests.test_gpu_shard8 import (CHUNK, DF4_A, DF4_B, DF20_C, ICAO_A, ICAO_B, ICAO_C, N_SAMPLES, coupled_capture8, key,
                                   preface_capture, want_key)

pytestmark = pytest.mark.gpu


def device_sets():
    """[0] * 8 and [0] * 3 always; every real device once, and twice round, when the box has more than one."""
    import torch
    sets = [[0] * 8, [0] * 3, [0]]
    n = torch.cuda.device_count()
    if n >= 2:
        sets += [list(range(n)), [k % n for k in range(2 * n)], list(range(n - 1, -1, -1))]
    return sets


def to_devices(iq, multi, torch):
    """The capture's contiguous ranges, each resident on its device: (tensors, pointers, sample counts)."""
    tensors, ptrs, ns = [], [], []
    for dev, (a, n) in zip(multi.devices, multi.shard_ranges(len(iq))):
        t = torch.from_numpy(np.ascontiguousarray(iq[a:a + n])).to(f"cuda:{dev}") if n else None
        tensors.append(t)
        ptrs.append(t.data_ptr() if n else 0)
        ns.append(n)
    for dev in set(multi.devices):
        torch.cuda.synchronize(dev)
    return tensors, ptrs, ns


@pytest.mark.parametrize("which", range(6))
def test_one_capture_over_n_devices_with_a_preloaded_filter_equals_single_stream_and_oracle(hip_lib, oracle_mod, which):
    """The coupled capture of tests/test_gpu_shard8.py (38 buffers, the last ragged; an address taught in one shard
    that frames in three others need; a DF11-taught address for a 112-bit DF20 three shards on; a filter that is NOT
    empty when the capture starts) through adsb_multi_demod_iq (host capture) and adsb_multi_demod_iq_device
    (resident shards): equal to the oracle and to one context demodulating the capture alone."""
    import torch
    from dump1090_rs_amd import Context
    from dump1090_rs_amd.multi import MultiContext
    sets = device_sets()
    if which >= len(sets):
        pytest.skip("one HIP device: the [0] * n forms cover the code, more devices add the placement")
    devices = sets[which]
    iq, pre = coupled_capture8(), preface_capture()
    orc = oracle_mod.Oracle()
    want_pre, _ = orc.demod_iq(pre)
    want, _ = orc.demod_iq(iq)                       # the filter persists from the preface
    cold_want, _ = oracle_mod.Oracle().demod_iq(iq)

    with Context(devices[0], 8) as solo:
        solo.icao_flush()
        solo.demod_iq(pre)
        single = solo.demod_iq(iq)
    assert [key(m) for m in single] == [want_key(w) for w in want]

    per = -(-38 // len(devices))
    with MultiContext(devices, per) as multi:
        assert multi.shard_ranges(N_SAMPLES)[0][0] == 0 and sum(n for _, n in multi.shard_ranges(N_SAMPLES)) == N_SAMPLES
        if len(devices) == 8:
            assert [-(-n // CHUNK) for _, n in multi.shard_ranges(N_SAMPLES)] == [5, 5, 5, 5, 5, 5, 4, 4]
        # host capture
        multi.icao_flush()
        assert [key(m) for m in multi.demod_iq(pre)] == [want_key(w) for w in want_pre]
        assert ICAO_B in multi.filter_table().tolist() and np.count_nonzero(multi.filter_table()) == 1
        got = multi.demod_iq(iq)
        assert [key(m) for m in got] == [want_key(w) for w in want]
        assert [key(m) for m in got] == [key(m) for m in single]
        st = multi.stats()
        assert st["n_devices"] == len(devices) and st["n_messages"] == len(want) and st["n_samples"] == N_SAMPLES
        assert st["n_chunks"] == 38 and st["retries"] == 0 and st["n_records"] >= len(want)
        if len(devices) > 1:
            assert st["n_addrs_exchanged"] >= 2        # A and C at least travelled between shards
        # the couplings really are in the capture
        chunks_of = lambda frame: sorted({m.chunk for m in got if m.buffer() == frame})
        assert chunks_of(DF4_A) == [8, 31, 37]          # not 2: A is not known yet
        assert chunks_of(DF4_B) == [0, 27]              # only because the filter held B already
        assert chunks_of(DF20_C) == [22]                # not 12
        table = multi.filter_table().tolist()
        assert ICAO_A in table and ICAO_B in table and ICAO_C in table
        # resident shards, behind a flush: no preface, so B's frames are gone
        tensors, ptrs, ns = to_devices(iq, multi, torch)
        multi.icao_flush()
        cold = multi.demod_iq_device(ptrs, ns)
        assert [key(m) for m in cold] == [want_key(w) for w in cold_want]
        assert not [m for m in cold if m.buffer() == DF4_B]
        # ... and again without a flush: the filter is warm, on both sides
        orc2 = oracle_mod.Oracle()
        orc2.demod_iq(iq)
        warm_want, _ = orc2.demod_iq(iq)
        assert [key(m) for m in multi.demod_iq_device(ptrs, ns)] == [want_key(w) for w in warm_want]
        del tensors


def test_captures_in_flight_four_deep_with_flushes_between_some(hip_lib, oracle_mod):
    """adsb_multi_submit_iq_device / adsb_multi_collect: ten captures over eight contexts, four in flight, an
    icao_flush in front of some: results in submission order, each equal to ONE oracle stream that flushes at the
    same points -- the scans of capture i + 1 run while capture i is exchanged, matched and replayed, and the
    address a capture learns must reach the next one's match although that was scanned before the replay."""
    import torch
    from dump1090_rs_amd.multi import MultiContext
    seeds = [4108, 4109, 4110, 4111, 4112, 4113, 4114, 4115, 4116, 4117]
    flush_before = {0, 3, 4, 8}
    caps = []
    for i, s in enumerate(seeds):
        iq = coupled_capture8(s)
        if i % 3 == 1:
            iq = iq[: 21 * CHUNK + 555]              # a shorter capture: other ranges, some shards nearly empty
        if i == 6:
            iq = iq[: 3 * CHUNK]                     # three buffers over eight devices: five empty shards
        caps.append(np.ascontiguousarray(iq))
    orc = oracle_mod.Oracle()
    wants = []
    for i, iq in enumerate(caps):
        if i in flush_before:
            orc.icao_flush()
        wants.append(orc.demod_iq(iq)[0])
    # (capture 1 follows capture 0 without a flush: the DF4 for A in its buffer 2 is found now)
    assert sum(w["buffer"] == DF4_A and w["chunk"] == 2 for w in wants[1]) == 1
    assert sum(w["buffer"] == DF4_A and w["chunk"] == 2 for w in wants[0]) == 0
    with MultiContext([0] * 8, 5) as multi:
        depth = multi.max_in_flight()
        assert depth == 4
        resident = [to_devices(iq, multi, torch) for iq in caps]
        gots, refused = [], 0
        for i in range(len(caps)):
            if multi.pending() == depth:
                with pytest.raises(Exception):           # a fifth capture in flight is refused, not queued
                    multi.submit_iq_device(resident[i][1], resident[i][2])
                refused += 1
                assert multi.pending() == depth
                gots.append(multi.collect())
            if i in flush_before:
                multi.icao_flush()
            multi.submit_iq_device(resident[i][1], resident[i][2])
        assert refused == len(caps) - depth
        while multi.pending():
            gots.append(multi.collect())
        assert len(gots) == len(caps)
        for i, (g, w) in enumerate(zip(gots, wants)):
            assert [key(m) for m in g] == [want_key(x) for x in w], f"capture {i}"


def test_host_captures_in_flight_from_pinned_and_from_ordinary_memory(hip_lib, oracle_mod):
    """adsb_multi_submit_iq: HOST captures asynchronously, four in flight over three contexts -- out of memory from
    adsb_multi_host_alloc (every device thread's copy is a DMA in front of its scan) and out of ordinary numpy
    memory, mixed with device-resident captures, an icao_flush in front of some: one oracle stream flushed at the
    same points.  A capture longer than the contexts hold together is refused; pinned memory cannot be freed
    under a capture in flight."""
    import torch
    from dump1090_rs_amd.multi import MultiContext
    seeds = [4301, 4302, 4303, 4304, 4305, 4306, 4307]
    flush_before = {0, 2, 5}
    caps = []
    for i, s in enumerate(seeds):
        iq = coupled_capture8(s)
        caps.append(np.ascontiguousarray(iq[: (20 - 3 * (i % 3)) * CHUNK + (777 if i % 2 else 0)]))
    orc = oracle_mod.Oracle()
    wants = []
    for i, iq in enumerate(caps):
        if i in flush_before:
            orc.icao_flush()
        wants.append(orc.demod_iq(iq)[0])
    with MultiContext([0] * 3, 7) as multi:
        with pytest.raises(Exception):
            multi.submit_iq(np.zeros((22 * CHUNK, 2), dtype=np.int16))       # 22 buffers > 3 x 7
        assert multi.pending() == 0
        pinned = [multi.host_alloc(21 * CHUNK) for _ in range(3)]
        resident = {}
        gots = []
        for i, iq in enumerate(caps):
            if multi.pending() == multi.max_in_flight():
                gots.append(multi.collect())
            if i in flush_before:
                multi.icao_flush()
            if i % 3 == 2:                              # a device-resident capture between the host ones
                resident[i] = to_devices(iq, multi, torch)
                multi.submit_iq_device(resident[i][1], resident[i][2])
            elif i % 3 == 0:                            # out of pinned memory (a slot nothing in flight reads)
                slot = pinned[(i // 3) % 3]
                slot[: len(iq)] = iq
                multi.submit_iq(slot[: len(iq)])
            else:
                multi.submit_iq(iq)                     # out of ordinary memory
        with pytest.raises(Exception):
            multi.host_free(pinned[0])                  # captures in flight
        while multi.pending():
            gots.append(multi.collect())
        for i, (g, w) in enumerate(zip(gots, wants)):
            assert [key(m) for m in g] == [want_key(x) for x in w], f"capture {i}"
        multi.host_free(pinned[0])
        with pytest.raises(Exception):
            multi.host_free(pinned[0])                  # twice
        # (the other two blocks go with the MultiContext)


def dense_capture(seed, n_buffers, bursts_per_buffer, n_icao=40):
    return synth.make_iq(n_buffers * CHUNK - 4321, n_bursts=n_buffers * bursts_per_buffer, seed=seed, n_icao=n_icao, df11_every=5)


def test_dense_shards_order_their_records_on_the_device_and_sparse_ones_do_not(hip_lib, oracle_mod):
    """A busy sky leaves tens of records per buffer: from the second dense capture on, a shard's second phase hands
    its records over in replay order (buckets per buffer and tile, rank sort in the records kernel -- what dense
    single-stream passes do) and the host sorts nothing; a sparse capture switches back.  Dense, dense, flush + dense,
    sparse, sparse, dense, four in flight over three contexts of 24 buffers: every capture equal to one oracle stream."""
    import torch
    from dump1090_rs_amd.multi import MultiContext
    plan = [("dense", 5101, False), ("dense", 5102, False), ("dense", 5103, True), ("dense", 5104, False),
            ("sparse", 5105, False), ("sparse", 5106, False), ("sparse", 5107, True), ("dense", 5108, False), ("dense", 5109, False)]
    caps = [dense_capture(seed, 66, 6) if kind == "dense" else synth.make_iq(66 * CHUNK - 999, n_bursts=30, seed=seed, n_icao=12)
            for kind, seed, _ in plan]
    orc = oracle_mod.Oracle()
    wants = []
    for (kind, seed, flush), iq in zip(plan, caps):
        if flush:
            orc.icao_flush()
        wants.append(orc.demod_iq(iq)[0])
    assert len(wants[0]) > 66 * 4 and len(wants[4]) < 66
    with MultiContext([0] * 3, 24) as multi:
        resident = [to_devices(iq, multi, torch) for iq in caps]
        gots = []
        for i, (kind, seed, flush) in enumerate(plan):
            if multi.pending() == multi.max_in_flight():
                gots.append(multi.collect(cap=1 << 16))
            if flush:
                multi.icao_flush()
            multi.submit_iq_device(resident[i][1], resident[i][2])
        while multi.pending():
            gots.append(multi.collect(cap=1 << 16))
        for i, (g, w) in enumerate(zip(gots, wants)):
            assert [key(m) for m in g] == [want_key(x) for x in w], f"capture {i}"
        ctr = multi.selftest_counters()
        assert ctr["device_ordered_shards"] >= 3 and ctr["fresh_list_fallbacks"] == 0
        # one at a time now (the density of capture i is known when capture i + 1 starts): dense ones after the first
        # arrive in order, nothing is sorted on the host for them
        before = multi.selftest_counters()
        for i in (0, 1, 2):
            multi.icao_flush()
            got = multi.demod_iq_device(resident[i][1], resident[i][2], cap=1 << 16)
            orc2 = oracle_mod.Oracle()
            assert [key(m) for m in got] == [want_key(x) for x in orc2.demod_iq(caps[i])[0]]
        after = multi.selftest_counters()
        assert after["device_ordered_shards"] - before["device_ordered_shards"] >= 6
        assert after["shards_sorted_on_host"] - before["shards_sorted_on_host"] <= 3


def test_captures_scored_by_several_host_threads_equal_the_serial_replay(hip_lib, oracle_mod):
    """adsb_multi_collect scores a capture of many records with several threads at once: every record against the filter
    as it was plus the positions at which the capture's new addresses enter it (csrc/adsb_replay_host.h).  With the
    threshold at 1 every capture goes that way -- the coupled ones (an address taught in one shard that frames in
    three others need, a pre-loaded filter), dense ones, with flushes, pipelined: same messages as the oracle's one
    stream, same filter table slot for slot."""
    import torch
    from dump1090_rs_amd.multi import MultiContext
    caps = [coupled_capture8(4401), dense_capture(4402, 60, 6), coupled_capture8(4403)[: 30 * CHUNK + 999], dense_capture(4404, 64, 9, n_icao=300),
            dense_capture(4405, 64, 2, n_icao=25), coupled_capture8(4406)]
    flush_before = {0, 3, 5}
    orc = oracle_mod.Oracle()
    wants = []
    for i, iq in enumerate(caps):
        if i in flush_before:
            orc.icao_flush()
        wants.append([want_key(x) for x in orc.demod_iq(iq)[0]])
    want_table = list(orc.filter.a)
    for devices in ([0] * 8, [0] * 3, [0]):
        per = -(-max(len(c) for c in caps) // CHUNK // len(devices)) + 1
        with MultiContext(devices, max(per, 17)) as multi:
            multi.selftest_tune(parallel_min=1, score_mode=1)   # (the host scores: no shard is scored on its device)
            resident = [to_devices(iq, multi, torch) for iq in caps]
            gots = []
            for i in range(len(caps)):
                if multi.pending() == 3:
                    gots.append(multi.collect(cap=1 << 16))
                if i in flush_before:
                    multi.icao_flush()
                multi.submit_iq_device(resident[i][1], resident[i][2])
            while multi.pending():
                gots.append(multi.collect(cap=1 << 16))
            for i, (g, w) in enumerate(zip(gots, wants)):
                assert [key(m) for m in g] == w, f"capture {i} over {len(devices)} contexts"
            assert list(multi.filter_table()) == list(want_table)
            assert multi.selftest_counters()["parallel_scored_captures"] == len(caps)


def test_dense_shards_scored_on_their_devices_equal_the_ordered_replay(hip_lib, oracle_mod):
    """A dense stream's shards are scored where they are: k_score / k_emit behind the second phase's records kernel,
    against the context's exact bitmap (every capture's additions committed to it on every device, an icao_flush
    switching to the cleared one) and the additions of the shards BEFORE this one in the capture (ScoreDev::earlier).
    The collector only concatenates.  Coupled captures (an address taught in shard 1 that frames in shards 1, 6, 7 need;
    DF11-taught; a pre-loaded filter), dense ones with 25 and 300 aircraft, sparse ones in between (host-scored), flushes,
    four in flight, over 3 and 2 contexts: every capture equal to the oracle's ONE stream, the filter table slot for
    slot -- with the results used, and with every result refused (the path a filter table about to fill up takes: the
    records are fetched from the devices and replayed by the host)."""
    import torch
    from dump1090_rs_amd.multi import MultiContext
    caps = [dense_capture(4501, 60, 6), coupled_capture8(4502), dense_capture(4503, 60, 9, n_icao=300), dense_capture(4504, 57, 5, n_icao=25),
            synth.make_iq(60 * CHUNK - 999, n_bursts=30, seed=4505, n_icao=12), dense_capture(4506, 60, 7), coupled_capture8(4507)[: 33 * CHUNK + 4444],
            dense_capture(4508, 60, 6, n_icao=60), dense_capture(4509, 60, 6, n_icao=60)]
    flush_before = {0, 3, 7}
    orc = oracle_mod.Oracle()
    wants = []
    for i, iq in enumerate(caps):
        if i in flush_before:
            orc.icao_flush()
        wants.append([want_key(x) for x in orc.demod_iq(iq)[0]])
    want_table = list(orc.filter.a)
    for devices, mode in (([0] * 3, 0), ([0] * 2, 0), ([0] * 3, 2)):
        with MultiContext(devices, 30) as multi:
            multi.selftest_tune(score_mode=mode)
            resident = [to_devices(iq, multi, torch) for iq in caps]
            gots = []
            for i in range(len(caps)):
                if multi.pending() == multi.max_in_flight():
                    gots.append(multi.collect(cap=1 << 16))
                if i in flush_before:
                    multi.icao_flush()
                multi.submit_iq_device(resident[i][1], resident[i][2])
            while multi.pending():
                gots.append(multi.collect(cap=1 << 16))
            for i, (g, w) in enumerate(zip(gots, wants)):
                assert [key(m) for m in g] == w, f"capture {i} over {len(devices)} contexts, score_mode {mode}"
            assert list(multi.filter_table()) == want_table
            ctr = multi.selftest_counters()
            assert ctr["device_scored_shards"] >= 2 * len(devices), ctr
            if mode == 0:
                assert ctr["scored_results_used"] >= 2 * len(devices) and ctr["scored_results_refused"] == 0, ctr
            else:
                assert ctr["scored_results_used"] == 0 and ctr["scored_results_refused"] >= 2 * len(devices), ctr


def test_a_capture_that_teaches_more_addresses_than_the_fresh_list_holds(hip_lib, oracle_mod):
    """A shard's scan lists the addresses its trials can add to the filter (all the exchange needs); a capture
    with more aircraft than the list holds reads them out of its trial records instead.  With the list cut to
    three entries every shard of these captures takes that path, sparse and dense (device-ordered) ones, pipelined,
    with and without a flush: equal to the oracle, and to the same captures with the list at its full size."""
    import torch
    from dump1090_rs_amd.multi import MultiContext
    caps = [dense_capture(5201 + i, 60, 5 if i % 2 == 0 else 1, n_icao=30) for i in range(5)]
    orc = oracle_mod.Oracle()
    wants = []
    for i, iq in enumerate(caps):
        if i == 3:
            orc.icao_flush()
        wants.append([want_key(x) for x in orc.demod_iq(iq)[0]])
    for cap_n in (3, 0):
        with MultiContext([0] * 3, 20) as multi:
            multi.selftest_tune(fresh_cap=cap_n)
            resident = [to_devices(iq, multi, torch) for iq in caps]
            gots = []
            for i in range(len(caps)):
                if multi.pending() == 2:
                    gots.append(multi.collect(cap=1 << 16))
                if i == 3:
                    multi.icao_flush()
                multi.submit_iq_device(resident[i][1], resident[i][2])
            while multi.pending():
                gots.append(multi.collect(cap=1 << 16))
            for i, (g, w) in enumerate(zip(gots, wants)):
                assert [key(m) for m in g] == w, f"capture {i}, fresh list of {cap_n or 16384}"
            ctr = multi.selftest_counters()
            assert (ctr["fresh_list_fallbacks"] >= 6) == (cap_n == 3), ctr


def test_small_ragged_and_empty_captures(hip_lib, oracle_mod):
    """Fewer buffers than devices, a capture shorter than one buffer, an empty capture: shards without samples
    still take part in the exchange (their superset must learn what the others taught)."""
    from dump1090_rs_amd.multi import MultiContext
    icao = 0x3C6589
    body = bytes([0x20, 0x00, 0x05, 0x30])
    df4 = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
    a = synth.make_iq(2 * CHUNK, n_bursts=10, seed=321, n_icao=3)
    synth.add_bursts(a, [synth.Burst(5 * 50000 + 2, 22000, 3, synth.df17_frame(icao, 9))])
    b = synth.make_iq(7 * CHUNK + 99, n_bursts=20, seed=322, n_icao=3)
    synth.add_bursts(b, [synth.Burst(5 * (k * CHUNK + 4000 + 17 * k) + k % 5, 21000, k, df4) for k in range(7)])
    c = synth.make_iq(4321, n_bursts=0, seed=323)
    orc = oracle_mod.Oracle()
    want = [orc.demod_iq(x)[0] for x in (a, b, c, a[:0], b)]
    assert sum(w["buffer"] == df4 for w in want[1]) >= 7       # (a burst may decode at j and j + 1, tests/test.rs:24-25)
    with MultiContext([0] * 8, 2) as multi:
        multi.icao_flush()
        for iq, w in zip((a, b, c, a[:0], b), want):
            # (b is 8 buffers over 8 devices; a leaves six devices without samples, yet b's DF4s on THEIR devices
            # need the address a's DF17 taught)
            assert [key(m) for m in multi.demod_iq(iq)] == [want_key(x) for x in w]
        # a capture longer than the contexts hold together goes in pieces through the same filter
        long_iq = np.concatenate([b, a, b])
        orc3 = oracle_mod.Oracle()
        multi.icao_flush()
        assert [key(m) for m in multi.demod_iq(long_iq)] == [want_key(x) for x in orc3.demod_iq(long_iq)[0]]


def test_a_shard_denser_than_the_lists_goes_buffer_by_buffer_and_stays_exact(hip_lib, oracle_mod):
    """One shard of the capture overflows the fast scan's lists (a one-buffer context given a periodic stretch
    that passes every gate at a quarter of all positions): that shard takes both phases buffer by buffer through
    the worst-case lists while the others go the usual way; the merged result is still the oracle's."""
    from dump1090_rs_amd.multi import MultiContext
    from tests.test_gpu_parity import ADVERSARIAL_PERIODS
    icao = 0x3C6589
    body = bytes([0x20, 0x00, 0x05, 0x30])
    df4 = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
    iq = synth.make_iq(4 * CHUNK, n_bursts=0, seed=900)
    synth.add_bursts(iq, [synth.Burst(5 * (30000 * k + 777) + k, 21000, k, synth.df17_frame(icao, k)) for k in range(1, 4)])
    a, b = CHUNK + 40000, CHUNK + 125000                     # inside buffer 1 = shard 1
    per = np.array(ADVERSARIAL_PERIODS[1], dtype=np.int16)
    iq[a:b, 0] = np.tile(per, (b - a) // len(per) + 1)[: b - a]
    iq[a:b, 1] = 0
    synth.add_bursts(iq, [synth.Burst(5 * (CHUNK * k + 9000 + 333 * k) + k, 21000, k, df4) for k in range(1, 4)])
    want, _ = oracle_mod.Oracle().demod_iq(iq, cap=1 << 18)
    assert sum(w["buffer"] == df4 for w in want) >= 3
    with MultiContext([0] * 4, 1) as multi:
        multi.icao_flush()
        got = multi.demod_iq(iq, cap=1 << 18)
        assert multi.stats()["retries"] >= 1                 # the scenario really went through the fallback
        assert [key(m) for m in got] == [want_key(w) for w in want]
        # and the handle is as good as new afterwards
        multi.icao_flush()
        assert [key(m) for m in multi.demod_iq(iq, cap=1 << 18)] == [want_key(w) for w in want]


def test_a_dense_shard_with_one_overfull_buffer_bucket_goes_buffer_by_buffer(hip_lib, oracle_mod):
    """Contexts of 20 buffers (full bitmaps: the scan lists the shard's addresses, a dense stream's shards put their hits
    into per-buffer buckets): one buffer packed with back-to-back frames holds more hits than its bucket of 1024 while
    no other list is anywhere near full.  That shard must be flagged and go through both phases buffer by buffer while
    the other shard and the captures in flight around it go the usual way -- every capture equal to the oracle's stream."""
    import torch
    from dump1090_rs_amd.multi import MultiContext
    n = 40 * CHUNK
    host = [synth.make_iq(n, n_bursts=3000, seed=6200 + k, n_icao=40) for k in range(4)]
    packed = synth.noise_numpy(CHUNK, seed=78)
    synth.add_bursts(packed, [synth.Burst(5 * (200 + 300 * q) + q % 5, 14000 + 10 * q, q % 16,
                                          synth.df17_frame(0xA00000 + 0x101 * (q % 40), q)) for q in range(430)])
    host[2][25 * CHUNK:26 * CHUNK] = packed               # shard 1 of the third capture
    orc = oracle_mod.Oracle()
    wants = [[want_key(x) for x in orc.demod_iq(h)[0]] for h in host]
    with MultiContext([0] * 2, 20) as multi:
        resident = [to_devices(h, multi, torch) for h in host]
        multi.icao_flush()
        gots = [multi.demod_iq_device(resident[0][1], resident[0][2], cap=1 << 16)]    # (tells the contexts how dense the stream is)
        retries = []
        for i in (1, 2, 3):
            multi.submit_iq_device(resident[i][1], resident[i][2])
        for i in (1, 2, 3):
            gots.append(multi.collect(cap=1 << 16))
            retries.append(multi.stats()["retries"])
        assert retries == [0, 1, 0], retries                 # the overfull bucket did overflow, and only that shard
        for i, (g, w) in enumerate(zip(gots, wants)):
            assert [key(m) for m in g] == w, f"capture {i}"
        assert multi.selftest_counters()["device_ordered_shards"] >= 5


def test_output_array_too_small_and_misuse(hip_lib, oracle_mod):
    """ADSB_ERR_CAPACITY hands out the first `cap` messages and keeps the list (the capture is consumed: the
    filter has moved on); bad arguments are refused, never crash."""
    import torch
    from dump1090_rs_amd import _lib
    from dump1090_rs_amd._lib import AdsbMsg
    from dump1090_rs_amd.multi import MultiContext
    L = hip_lib
    iq = coupled_capture8()
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    with MultiContext([0] * 3, 13) as multi:
        multi.icao_flush()
        out, n = (AdsbMsg * 4)(), C.c_size_t()
        st = L.adsb_multi_demod_iq(multi._h, iq.ctypes.data, len(iq), out, 4, C.byref(n))
        assert st == _lib.ADSB_ERR_CAPACITY and n.value == len(want) > 4
        assert [bytes(m.msg) for m in out] == [w["msg"] for w in want[:4]]
        full = (AdsbMsg * n.value)()
        assert L.adsb_multi_fetch_messages(multi._h, full, n.value, C.byref(n)) == 0
        assert [(m.chunk, m.j, m.score) for m in full] == [(w["chunk"], w["j"], w["score"]) for w in want]
        # the Python mirror fetches by itself
        multi.icao_flush()
        assert [key(m) for m in multi.demod_iq(iq, cap=2)] == [want_key(w) for w in want]
        # misuse
        tensors, ptrs, ns = to_devices(iq, multi, torch)
        bad = list(ns)
        bad[0] -= 5                                           # a ragged shard in the middle of the capture
        with pytest.raises(Exception):
            multi.demod_iq_device(ptrs, bad)
        with pytest.raises(Exception):
            multi.demod_iq_device([p + 4 for p in ptrs], ns)  # not 16-byte aligned
        with pytest.raises(Exception):
            multi.demod_iq_device(ptrs, [14 * CHUNK, 0, 0])   # more than the context was created for
        assert L.adsb_multi_collect(multi._h, None, 0, None) == _lib.ADSB_ERR_INVALID      # nothing in flight
        assert L.adsb_multi_fetch_messages(multi._h, None, 0, None) == _lib.ADSB_ERR_CAPACITY   # (the cap=2 call above left its list)
        assert [key(m) for m in multi.demod_iq_device(ptrs, ns)] is not None                 # still usable
    h = C.c_void_p()
    assert L.adsb_multi_create(C.byref(h), None, 2, 1) == _lib.ADSB_ERR_INVALID
    assert L.adsb_multi_create(C.byref(h), (C.c_int * 1)(0), 0, 1) == _lib.ADSB_ERR_INVALID
    assert L.adsb_multi_create(C.byref(h), (C.c_int * 2)(0, 99), 2, 1) == _lib.ADSB_ERR_NO_DEVICE and not h.value
    assert L.adsb_multi_create(C.byref(h), (C.c_int * 1)(-1), 1, 1) == _lib.ADSB_ERR_NO_DEVICE
    assert L.adsb_multi_shard_range(10, 0, 0, None, None) == _lib.ADSB_ERR_INVALID
    assert L.adsb_multi_shard_range(10, 2, 2, None, None) == _lib.ADSB_ERR_INVALID
    L.adsb_multi_destroy(None)


def test_shard_ranges_are_the_even_contiguous_split():
    """adsb_multi_shard_range == sharding.sample_range (whole buffers, sizes that differ by at most one)."""
    from dump1090_rs_amd import _lib, sharding
    L = _lib.lib()
    for n_samples in (0, 1, CHUNK - 1, CHUNK, CHUNK + 1, 38 * CHUNK - 4321, 4096 * CHUNK):
        for world in (1, 2, 3, 8, 64):
            for k in range(world):
                a, n = C.c_size_t(), C.c_size_t()
                assert L.adsb_multi_shard_range(n_samples, world, k, C.byref(a), C.byref(n)) == 0
                lo, hi = sharding.sample_range(n_samples, world, k)
                assert (a.value, a.value + n.value) == (lo, hi)


def test_compiled_c_host_drives_the_multi_entry_points(hip_lib, golden):
    """tests/abi_host.c --multi N: a plain C program over include/adsb_hip.h, no Python and no process group, puts
    N copies of a reference capture through adsb_multi_demod_iq as ONE capture over N contexts; the first buffer's
    frames are exactly the reference's (tests/test.rs:22-28)."""
    exe = ROOT / "tests" / "abi_host"
    assert exe.exists(), "tests/abi_host was not built (dump1090_rs_amd.build.build_abi_host)"
    fx = golden["fixtures"][0]
    for n in (1, 3, 8):
        r = subprocess.run([str(exe), "--multi", str(n), str(GOLDEN / fx["file"]), *fx["frames"]], capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = r.stdout.splitlines()
        assert [ln.split()[0] for ln in lines[:-3]] == fx["frames"] and lines[-3].startswith(f"multi: {n} devices")
        # ... a failed shard, ADSB_ERR_POISONED behind it, the restart (include/adsb_hip.h: "When a capture fails")
        assert lines[-1].startswith("multi: a failed shard poisoned the handle")
        # ... and the same capture three times in flight out of adsb_multi_host_alloc memory (adsb_multi_submit_iq)
        assert lines[-2] == "multi: three pinned host captures in flight, each equal to the blocking call"
    assert subprocess.run([str(exe), "--multi", "3", str(GOLDEN / fx["file"]), *fx["frames"][:-1]], capture_output=True).returncode == 1


def capture_through_one_device_and_through_eight_contexts(oracle_mod, n_buffers, n_bursts, seed, threads):
    import torch
    from dump1090_rs_amd.multi import MultiContext
    n = n_buffers * CHUNK
    dev = synth.make_iq_torch(n, n_bursts=n_bursts, seed=seed, device=torch.device("cuda", 0))
    torch.cuda.synchronize()
    want, _ = oracle_mod.Oracle().demod_iq(dev.cpu().numpy(), cap=1 << 18, threads=threads)
    assert len(want) >= n_bursts * 9 // 10
    for devices, per in (([0], n_buffers), ([0] * 8, n_buffers // 8)):
        with MultiContext(devices, per) as multi:
            ranges = multi.shard_ranges(n)
            ptrs = [dev.data_ptr() + 4 * a for a, _ in ranges]
            ns = [k for _, k in ranges]
            multi.icao_flush()
            assert [key(m) for m in multi.demod_iq_device(ptrs, ns, cap=1 << 18)] == [want_key(w) for w in want]
            # pipelined, a flush in front of each: the same list every time
            for _ in range(6):
                if multi.pending() == multi.max_in_flight():
                    assert [key(m) for m in multi.collect(cap=1 << 18)] == [want_key(w) for w in want]
                multi.icao_flush()
                multi.submit_iq_device(ptrs, ns)
            while multi.pending():
                assert [key(m) for m in multi.collect(cap=1 << 18)] == [want_key(w) for w in want]


def test_256_mib_capture_through_one_device_and_through_eight_contexts(hip_lib, oracle_mod):
    """512 buffers = 256 MiB: one device holding everything, then eight contexts of 64 buffers -- both equal to the
    oracle, blocking and four captures in flight."""
    capture_through_one_device_and_through_eight_contexts(oracle_mod, 512, 400, 77001, 8)


def test_two_gib_capture_through_one_device_and_through_eight_contexts(hip_lib, oracle_mod):
    """BASELINE config 4's capture, the real size: 4096 buffers = 2 GiB through one context of 4096 buffers and through
    eight of 512, against the threaded oracle over the whole capture (sixteen threads: a second or two)."""
    capture_through_one_device_and_through_eight_contexts(oracle_mod, 4096, 3200, 77002, 16)


# ------------------------------------------------------------------------------------------------ when a capture fails
def fault_kinds():
    from dump1090_rs_amd import _lib
    return [_lib.ADSB_FAULT_PHASE1, _lib.ADSB_FAULT_PHASE2, _lib.ADSB_FAULT_RECORDS]


@pytest.mark.parametrize("per,kind_idx,wait", [(5, 0, 0), (5, 1, 2), (5, 2, 1), (24, 0, 2), (24, 1, 0), (24, 2, 0)])
def test_a_failed_shard_poisons_the_handle_and_a_flush_starts_the_stream_over(hip_lib, oracle_mod, per, kind_idx, wait):
    """include/adsb_hip.h, "When a capture fails": one of eight shards of the third capture of a pipeline fails (phase 1,
    phase 2, or its records refused).  The captures in front of it are the oracle's; ITS collect returns the shard's
    error and names the device; the captures in flight behind it and a further submission return ADSB_ERR_POISONED --
    never a silently different list; adsb_multi_icao_flush with captures still in flight is refused; with nothing in
    flight it resets every context, and what follows equals ONE fresh oracle stream, learned addresses and
    address/parity hits included; destroy returns.  Contexts of 5 buffers (folded supersets, one-launch phases) and of 24
    (full bitmaps, exact bitmaps, device-scored dense shards); spinning and blocking device threads."""
    import torch
    from dump1090_rs_amd import _lib
    from dump1090_rs_amd._lib import AdsbError
    from dump1090_rs_amd.multi import MultiContext
    kind = fault_kinds()[kind_idx]
    n_buf = 8 * per - 3
    caps = [np.ascontiguousarray(coupled_capture8(4300 + i)[: n_buf * CHUNK]) if per == 5 else dense_capture(6300 + i, n_buf, 6, n_icao=12)
            for i in range(7)]
    orc = oracle_mod.Oracle()
    wants_before = [orc.demod_iq(caps[i])[0] for i in (0, 1)]
    with MultiContext([0] * 8, per) as multi:
        multi.set_wait(wait)
        resident = [to_devices(iq, multi, torch) for iq in caps]
        multi.icao_flush()
        multi.submit_iq_device(*resident[0][1:])
        multi.submit_iq_device(*resident[1][1:])
        multi.selftest_fail(0, 5, kind)                       # shard 5 of the NEXT capture
        multi.submit_iq_device(*resident[2][1:])
        multi.submit_iq_device(*resident[3][1:])
        for i in (0, 1):
            assert [key(m) for m in multi.collect()] == [want_key(w) for w in wants_before[i]], f"capture {i}"
        with pytest.raises(AdsbError) as e:
            multi.collect()
        assert e.value.status == _lib.ADSB_ERR_HIP and "shard 5" in str(e.value) and "injected" in str(e.value)
        assert multi.stats()["n_messages"] == 0
        with pytest.raises(AdsbError) as e:                    # behind it: submitted before the failure was known
            multi.submit_iq_device(*resident[4][1:])
        assert e.value.status == _lib.ADSB_ERR_POISONED and multi.pending() == 1
        with pytest.raises(AdsbError) as e:                    # the restart wants nothing in flight
            multi.icao_flush()
        assert e.value.status == _lib.ADSB_ERR_BUSY
        with pytest.raises(AdsbError) as e:
            multi.collect()
        assert e.value.status == _lib.ADSB_ERR_POISONED and multi.pending() == 0
        with pytest.raises(AdsbError) as e:
            multi.demod_iq(caps[4])
        assert e.value.status == _lib.ADSB_ERR_POISONED and multi.selftest_counters()["poisoned"] == 1
        multi.icao_flush()                                     # the restart
        assert multi.selftest_counters()["poisoned"] == 0
        fresh = oracle_mod.Oracle()
        fresh.icao_flush()
        for i in (4, 5, 6, 2):
            multi.submit_iq_device(*resident[i][1:])
        for i in (4, 5, 6, 2):
            assert [key(m) for m in multi.collect(cap=1 << 16)] == [want_key(w) for w in fresh.demod_iq(caps[i])[0]], f"capture {i} after the restart"
        assert np.array_equal(multi.filter_table(), np.ctypeslib.as_array(fresh.filter.a))


def test_a_device_that_stops_answering_is_given_up_and_destroy_returns(hip_lib, oracle_mod):
    """The timeout path: a shard phase that "never finishes" (the hook makes the device thread blind to it) fails its
    capture after the handle's timeout, the device is given up, everything behind it is refused, the restart is refused
    too (a dead device cannot be reset), and adsb_multi_destroy returns instead of waiting for it."""
    import time
    import torch
    from dump1090_rs_amd import _lib
    from dump1090_rs_amd._lib import AdsbError
    from dump1090_rs_amd.multi import MultiContext
    iq = np.ascontiguousarray(coupled_capture8(4400)[: 16 * CHUNK])
    want = oracle_mod.Oracle().demod_iq(iq)[0]
    multi = MultiContext([0] * 4, 4)
    multi.set_timeout_ms(250)
    res = to_devices(iq, multi, torch)
    multi.icao_flush()
    assert [key(m) for m in multi.demod_iq_device(*res[1:])] == [want_key(w) for w in want]
    multi.selftest_fail(1, 2, _lib.ADSB_FAULT_HANG)           # shard 2 of the capture AFTER the next
    multi.submit_iq_device(*res[1:])
    multi.submit_iq_device(*res[1:])
    multi.submit_iq_device(*res[1:])
    t0 = time.perf_counter()
    assert len(multi.collect()) > 0                           # (no flush in between: more address/parity hits than the first time)
    with pytest.raises(AdsbError) as e:
        multi.collect()
    waited = time.perf_counter() - t0
    assert e.value.status == _lib.ADSB_ERR_HIP and "timeout" in str(e.value) and 0.2 < waited < 5.0
    with pytest.raises(AdsbError) as e:
        multi.collect()
    assert e.value.status == _lib.ADSB_ERR_POISONED
    with pytest.raises(AdsbError) as e:
        multi.icao_flush()
    assert e.value.status == _lib.ADSB_ERR_HIP and "destroy" in str(e.value)
    t0 = time.perf_counter()
    multi.close()
    assert time.perf_counter() - t0 < 5.0
    # ... and the process goes on: a new handle on the same device works
    with MultiContext([0] * 4, 4) as again:
        again.icao_flush()
        assert [key(m) for m in again.demod_iq_device(*res[1:])] == [want_key(w) for w in want]


def test_blocking_wait_mode_gives_the_same_results_and_auto_follows_the_cpus(hip_lib, oracle_mod):
    """adsb_multi_set_wait: the device threads asleep between looks (ADSB_WAIT_BLOCK) instead of polling mapped memory;
    AUTO resolves to BLOCK when the process may use fewer CPUs than 2 x (devices + 3) (affinity mask, cgroup quota) and to
    SPIN otherwise."""
    import os
    import torch
    from dump1090_rs_amd import _lib
    from dump1090_rs_amd.multi import MultiContext
    iq = dense_capture(6400, 8 * 20 - 2, 6, n_icao=12)
    want = oracle_mod.Oracle().demod_iq(iq)[0]
    before = os.sched_getaffinity(0)
    try:
        with MultiContext([0] * 8, 20) as multi:
            res = to_devices(iq, multi, torch)
            for mode in (_lib.ADSB_WAIT_BLOCK, _lib.ADSB_WAIT_SPIN, _lib.ADSB_WAIT_BLOCK):
                multi.set_wait(mode)
                assert multi.get_wait() == mode
                for _ in range(6):
                    if multi.pending() == multi.max_in_flight():
                        assert [key(m) for m in multi.collect(cap=1 << 16)] == [want_key(w) for w in want]
                    multi.icao_flush()
                    multi.submit_iq_device(*res[1:])
                while multi.pending():
                    assert [key(m) for m in multi.collect(cap=1 << 16)] == [want_key(w) for w in want]
            multi.set_wait(_lib.ADSB_WAIT_AUTO)
            cpu_max = open("/sys/fs/cgroup/cpu.max").read().split() if os.path.exists("/sys/fs/cgroup/cpu.max") else ["max"]
            usable = len(before) if cpu_max[0] == "max" else min(len(before), max(1, int(cpu_max[0]) // int(cpu_max[1])))
            assert multi.get_wait() == (_lib.ADSB_WAIT_SPIN if usable >= 22 else _lib.ADSB_WAIT_BLOCK)
            os.sched_setaffinity(0, sorted(before)[:4])
            multi.set_wait(_lib.ADSB_WAIT_AUTO)
            assert multi.get_wait() == _lib.ADSB_WAIT_BLOCK
    finally:
        os.sched_setaffinity(0, before)
