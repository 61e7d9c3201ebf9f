"""The host-only half of the product under -fsanitize=address,undefined.

dump1090_rs_amd/csrc/adsb_replay_host.cpp -- the ordered replay (reference src/demod_2400.rs:149-207 with
src/mode_s/mod.rs:34-139 scoring, src/icao_filter.rs:11-97, src/crc.rs:263-282), its radix / insertion sort,
adsb_replay_records, adsb_format_raw, the learned-address union of the sharded capture -- has no HIP in it, so
plain g++ builds it with the sanitizers (the GPU pool offers no device sanitizer).  A child process with libasan
preloaded feeds it every trial the oracle slices from the reference captures and from a synthetic stream, and
adversarial records (duplicates, `chunk` near 2^32, `pad` bits set with wrong hashes, n = 0 / 1 / 97, unsorted,
a filter table at its 4096 entries), and compares with the oracle's own scoring replayed in Python.  CPU only."""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

from tests.conftest import GOLDEN, ROOT

SRC = ROOT / "dump1090_rs_amd" / "csrc" / "adsb_replay_host.cpp"
OUT = ROOT / "tests" / "libadsb_hostonly_asan.so"

CHILD = r'''
import sys, json, ctypes as C, random
sys.path.insert(0, %(root)r)
import numpy as np
from oracle import binding
from dump1090_rs_amd import synth
from dump1090_rs_amd._lib import AdsbMsg, AdsbTrial

H = C.CDLL(%(lib)r)
O = binding.lib()
vp, sz = C.c_void_p, C.c_size_t
H.adsb_replay_records.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
H.adsb_format_raw.argtypes = [vp, C.c_char_p, sz]
H.adsb_selftest_learned_union.argtypes = [vp, sz, vp, sz, vp, sz, C.POINTER(sz)]
H.adsb_selftest_crc_table.argtypes = [vp]
H.adsb_strerror.restype = C.c_char_p
H.adsb_selftest_parallel_replay.argtypes = [vp, vp, sz, C.c_int, C.c_int, C.c_int, vp, sz, C.POINTER(sz), C.POINTER(C.c_int)]
went_parallel = {0: 0, 1: 0}

def parallel(recs, table_words, runs, parts, threads):
    """the same records through ParallelReplay (csrc/adsb_replay_host.h): `runs` shards, `parts` parts, `threads` threads"""
    table = (C.c_uint32 * 4096)(*(table_words if table_words is not None else [0] * 4096))
    cap = len(recs) + 8
    out, n, par = (AdsbMsg * cap)(), sz(), C.c_int(-1)
    st = H.adsb_selftest_parallel_replay(table, as_array(recs), len(recs), runs, parts, threads, out, cap, C.byref(n), C.byref(par))
    assert st == 0 and par.value in (0, 1)
    went_parallel[par.value] += 1
    return [(bytes(m.msg), m.len, m.try_phase, m.score, m.j, m.chunk, m.signal_level) for m in out[:n.value]], list(table), par.value

def trials_of(iq):
    out = []
    for c, off in enumerate(range(0, len(iq), 131072)):
        mb = binding.OrcMagBuf()
        O.orc_to_mag(np.ascontiguousarray(iq[off:off + 131072]).ctypes.data, min(131072, len(iq) - off), C.byref(mb))
        buf = (AdsbTrial * (5 * 131072 // 8))()
        n = O.orc_all_trials(C.byref(mb), c, buf, len(buf))
        assert n <= len(buf)
        out += [bytes(buf[i]) for i in range(n)]
    return out

def as_array(recs):
    return (AdsbTrial * len(recs)).from_buffer_copy(b"".join(recs)) if recs else (AdsbTrial * 1)()

def product(recs, table=None, cap=None):
    table = table if table is not None else (C.c_uint32 * 4096)()
    cap = len(recs) + 8 if cap is None else cap
    out, n = (AdsbMsg * max(cap, 1))(), sz()
    st = H.adsb_replay_records(table, as_array(recs), len(recs), out, cap, C.byref(n))
    return st, [(bytes(m.msg), m.len, m.try_phase, m.score, m.j, m.chunk, m.signal_level) for m in out[:min(cap, n.value)]], n.value, table

def reference(recs, filt):
    """demod_2400.rs:149-207 over records in (chunk, j, try_phase) order, scored by the oracle's score_modes_message."""
    arr = as_array(recs)
    order = sorted(range(len(recs)), key=lambda i: (arr[i].chunk, arr[i].j_tp & 0xFFFFFF, arr[i].j_tp >> 24))
    out, i = [], 0
    while i < len(order):
        pos = (arr[order[i]].chunk, arr[order[i]].j_tp & 0xFFFFFF)
        best, best_score, best_len = None, -2, 7
        while i < len(order) and (arr[order[i]].chunk, arr[order[i]].j_tp & 0xFFFFFF) == pos:
            r = arr[order[i]]
            i += 1
            ln, sc = C.c_int(), C.c_int32()
            if not O.orc_score_modes_message(C.byref(filt), bytes(r.msg), 14, C.byref(ln), C.byref(sc)):
                continue
            if sc.value > best_score:
                best, best_score, best_len = r, sc.value, ln.value
        if best is None or best_score < 0:
            continue
        level = float(best.power & ((1 << 40) - 1)) / 65535.0 / 65535.0 / 33.0
        out.append((bytes(best.msg), best_len, best.j_tp >> 24, best_score, pos[1], pos[0], level))
    return out

def check(recs, table_words=None, what=""):
    filt = binding.OrcFilter()
    table = (C.c_uint32 * 4096)()
    if table_words is not None:
        for k, v in enumerate(table_words):
            filt.a[k] = v
            table[k] = v
    st, got, n, table = product(recs, table)
    assert st == 0, (what, st)
    want = reference(recs, filt)
    assert got == want, (what, len(got), len(want))
    assert list(table) == list(filt.a), what     # the filter ends where the oracle's does
    # ... and by several threads at once: same messages, same table slot for slot, however the records are cut
    for runs, parts, threads in ((1, 2, 2), (3, 7, 3), (8, 8, 4), (2, 33, 5)):
        pgot, ptable, par = parallel(recs, table_words, runs, parts, threads)
        assert pgot == want and ptable == list(filt.a), (what, runs, parts, threads, par, len(pgot), len(want))
        if what in ("nearly full table", "full table"):
            assert par == 0, what                # add() can give up: membership is no set's any more, the plan is refused
    return got

# 1. every trial of the three reference captures -> the golden frames
golden = json.load(open(%(golden)r))
for fx in golden["fixtures"]:
    raw = np.fromfile(%(gdir)r + "/" + fx["file"], dtype="<i2").reshape(-1, 2)
    recs = trials_of(np.ascontiguousarray(raw[:, ::-1]))
    got = check(recs, what=fx["file"])
    assert [m[0][:m[1]].hex() for m in got] == fx["frames"]

# 2. a synthetic stream of several buffers, its trials shuffled (radix sort), in reverse (insertion / radix), as they come
rng = random.Random(20261002)
iq = synth.make_iq(3 * 131072 + 999, n_bursts=60, seed=77, n_icao=6, df11_every=3)
recs = trials_of(iq)
want, _ = binding.Oracle().demod_iq(iq)
got = check(recs, what="stream")
assert [(m[5], m[4], m[2], m[3], m[0][:m[1]]) for m in got] == [(w["chunk"], w["j"], w["try_phase"], w["score"], w["buffer"]) for w in want]
sh = recs[:]
rng.shuffle(sh)
assert check(sh, what="shuffled") == got
assert check(recs[::-1], what="reversed") == got

# 3. adversarial records: random bytes with decodable DFs, duplicates, chunk near 2^32, n = 0 / 1 / 97, unsorted
def rec(chunk, j, tp, msg, power=0, pad=0):
    t = AdsbTrial()
    t.power, t.chunk, t.j_tp, t.pad = power, chunk & 0xFFFFFFFF, (j & 0xFFFFFF) | (tp << 24), pad
    for k in range(14):
        t.msg[k] = msg[k]
    return bytes(t)

def crc_fix(msg, nbytes):
    """make the last three bytes the CRC of the rest (a clean DF11 / DF17 / DF18)"""
    body = bytes(msg[:nbytes - 3]) + b"\0\0\0"
    c = O.orc_modes_checksum(body, nbytes * 8)
    return list(body[:nbytes - 3]) + [(c >> 16) & 255, (c >> 8) & 255, c & 255] + list(msg[nbytes:])

def random_records(n, chunks, seed, dups=True):
    r = random.Random(seed)
    addrs = [r.randrange(1, 1 << 24) for _ in range(12)]
    out = []
    for _ in range(n):
        df = r.choice([0, 4, 5, 11, 11, 16, 17, 17, 17, 18, 20, 21, 24, 31, 1, 9, 19])
        msg = [(df << 3) | r.randrange(8)] + [r.randrange(256) for _ in range(13)]
        if df in (11, 17, 18) and r.random() < 0.7:
            a = r.choice(addrs)
            msg[1:4] = [(a >> 16) & 255, (a >> 8) & 255, a & 255]
            msg = crc_fix(msg, 14 if df >= 16 else 7)
            if df == 11 and r.random() < 0.3:
                msg[6] ^= r.randrange(1, 128)         # a non-zero interrogator id
        elif r.random() < 0.3:
            # an address/parity frame whose residual is one of the addresses
            a = r.choice(addrs)
            nb = 14 if df >= 16 else 7
            msg = crc_fix(msg, nb)
            msg[nb - 3] ^= (a >> 16) & 255
            msg[nb - 2] ^= (a >> 8) & 255
            msg[nb - 1] ^= a & 255
        if r.random() < 0.02:
            msg = [0] * 14                            # the reference's None
        out.append(rec(r.choice(chunks), r.randrange(0, 131072), r.randrange(4, 9), msg, power=r.randrange(1 << 38)))
    if dups:
        out += [r.choice(out) for _ in range(n // 5)] if out else []   # exact duplicates
    r.shuffle(out)
    return out

emitted, scores = 0, set()
for n in (0, 1, 2, 96, 97, 98, 500, 5000):
    for chunks in ([0], [0, 1, 2, 3], [0, 0xFFFFFFFF, 0xFFFFFFFE, 0x80000000, 7]):
        for dups in (True, False):   # (with duplicates the parallel plan is refused: two records of one position)
            got = check(random_records(n, chunks, 1000 * n + len(chunks), dups), what=("random", n, len(chunks), dups))
            emitted += len(got)
            scores |= {m[3] for m in got}
assert emitted > 1000 and scores == {750, 1000, 1400, 1600, 1800}, (emitted, scores)   # every score of mode_s/mod.rs:56-135 occurs

# 4. a filter table at its 4096 entries: icao_filter_add gives up (src/icao_filter.rs:46-62), test() walks the whole table
full = [0x100000 + 3 * k for k in range(4096)]
tab = [0] * 4096
f = binding.OrcFilter()
for a in full[:4090]:
    O.orc_icao_filter_add(C.byref(f), a)
check(random_records(800, [0, 1], 4242), table_words=list(f.a), what="nearly full table")
for a in full[4090:]:
    O.orc_icao_filter_add(C.byref(f), a)
assert all(v != 0 for v in f.a)
check(random_records(800, [0, 1], 4343), table_words=list(f.a), what="full table")

# 5. `pad` bits set with wrong residuals / hashes: the replay trusts them (they come from its own device code), so the
#    answers need not be the oracle's -- but nothing may be read or written out of bounds
bad = []
for k, rb in enumerate(random_records(3000, [0, 5], 99)):
    t = AdsbTrial.from_buffer_copy(rb)
    t.pad = rng.choice([1, 3, 3, 0xFFF3, 0xFFFF, 0x0013, 2])
    t.power = (rng.randrange(1 << 24) << 40) | rng.randrange(1 << 40)
    bad.append(bytes(t))
st, got, n, _ = product(bad)
assert st == 0 and n <= len(bad)

# 6. output arrays: too small -> ADSB_ERR_CAPACITY with the required count, the first `cap` written; null arguments refused
recs = random_records(400, [0], 5)
st, full_out, n_full, _ = product(recs)
st, part, n, _ = product(recs, cap=3)
assert n_full > 3 and st == -5 and n == n_full and part == full_out[:3]
assert H.adsb_replay_records(None, None, 0, None, 0, None) == -1
cnt = sz()
assert H.adsb_replay_records((C.c_uint32 * 4096)(), None, 0, None, 0, C.byref(cnt)) == 0 and cnt.value == 0

# 7. adsb_format_raw (dump1090_rs/src/main.rs:172-176)
m = AdsbMsg()
for k in range(14):
    m.msg[k] = (0x8D + 17 * k) & 255
for ln in (7, 14):
    m.len = ln
    buf = C.create_string_buffer(32)
    k = H.adsb_format_raw(C.byref(m), buf, 32)
    assert k == 2 * ln + 3 and buf.value == b"*" + bytes(m.msg[:ln]).hex().encode() + b";\n"
    exact = C.create_string_buffer(2 * ln + 4)
    assert H.adsb_format_raw(C.byref(m), exact, 2 * ln + 4) == k and exact.value == buf.value
    small = C.create_string_buffer(2 * ln + 3)
    assert H.adsb_format_raw(C.byref(m), small, 2 * ln + 3) == -5
for ln in (0, 6, 8, 15, 255):
    m.len = ln
    assert H.adsb_format_raw(C.byref(m), C.create_string_buffer(64), 64) == -1
assert H.adsb_format_raw(None, C.create_string_buffer(64), 64) == -1

# 8. the address exchange of a sharded capture: learned addresses, sorted, without the known ones
recs = random_records(2000, [0, 1, 2], 31337)
arr = as_array(recs)
want = set()
for t in arr:
    df = t.msg[0] >> 3
    if df == 17 or (df == 11 and O.orc_modes_checksum(bytes(t.msg), 56) == 0):
        want.add((t.msg[1] << 16) | (t.msg[2] << 8) | t.msg[3])
known = sorted(want)[::3] + [5, 5, 0xFFFFFF]
out, cnt = (C.c_uint32 * 4096)(), sz()
assert H.adsb_selftest_learned_union(arr, len(recs), (C.c_uint32 * len(known))(*known), len(known), out, 4096, C.byref(cnt)) == 0
assert list(out[:cnt.value]) == sorted(want - set(known)) and cnt.value > 0
assert H.adsb_selftest_learned_union(arr, len(recs), None, 0, out, 2, C.byref(cnt)) == -5 and cnt.value == len(want)
assert H.adsb_selftest_learned_union(None, 0, None, 0, None, 0, C.byref(cnt)) == 0 and cnt.value == 0

# 9. the CRC table the replay scores with == the oracle's regenerated reference table
t = (C.c_uint32 * 256)()
assert H.adsb_selftest_crc_table(t) == 0 and list(t) == [O.orc_crc_table_entry(i) for i in range(256)]
assert b"no CPU fallback" in H.adsb_strerror(-2) and H.adsb_strerror(12345) == b"unknown status"
assert went_parallel[1] > 60 and went_parallel[0] >= 8, went_parallel
print("sanitized host-only code ok")
'''


def test_host_only_code_under_address_and_ub_sanitizers(oracle_mod):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not Path(asan).exists() or shutil.which("g++") is None:
        pytest.skip("no libasan / g++ in this environment")
    hdrs = [SRC.parent / n for n in ("adsb_replay_host.h", "adsb_record.h", "mode_s_host.hpp")] + [ROOT / "include" / "adsb_hip.h"]
    if not OUT.exists() or OUT.stat().st_mtime < max(p.stat().st_mtime for p in [SRC, *hdrs]):
        subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-shared", "-fPIC", "-pthread", str(SRC), "-o", str(OUT)],
                       check=True)
    code = CHILD % {"root": str(ROOT), "lib": str(OUT), "golden": str(GOLDEN / "reference_frames.json"), "gdir": str(GOLDEN)}
    env = dict(__import__("os").environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0 and "sanitized host-only code ok" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_replay_pool_under_thread_sanitizer():
    """The threads of the host replay (csrc/adsb_replay_host.h: ReplayPool + ParallelReplay -- jobs handed out by claim
    flags, late wakers, parts handed back to the thread that had them, streamed copy-out) under -fsanitize=thread:
    tests/replay_pool_tsan.cpp replays 300 random captures serially and through ONE pool, filters carried over and flushed,
    2-41 parts on six threads; same messages, same table, no report.  CPU only, no oracle needed (serial against parallel;
    the serial replay is pinned to the oracle above)."""
    tsan = subprocess.run(["gcc", "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
    if not tsan or not Path(tsan).exists() or shutil.which("g++") is None:
        pytest.skip("no libtsan / g++ in this environment")
    exe = ROOT / "tests" / "replay_pool_tsan"
    src = [ROOT / "tests" / "replay_pool_tsan.cpp", SRC]
    hdrs = [SRC.parent / n for n in ("adsb_replay_host.h", "adsb_record.h", "mode_s_host.hpp")] + [ROOT / "include" / "adsb_hip.h"]
    if not exe.exists() or exe.stat().st_mtime < max(p.stat().st_mtime for p in [*src, *hdrs]):
        subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-fsanitize=thread", "-pthread",
                        *map(str, src), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe), "300"], capture_output=True, text=True, timeout=900,
                       env=dict(__import__("os").environ, TSAN_OPTIONS="halt_on_error=1"))
    if "unexpected memory mapping" in r.stderr:   # (a kernel whose address-space layout this libtsan does not know)
        pytest.skip("ThreadSanitizer cannot run on this kernel")
    assert r.returncode == 0 and "replay pool ok: 300 captures" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
