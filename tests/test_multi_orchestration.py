"""adsb_multi's orchestration on the CPU, under ThreadSanitizer and under AddressSanitizer + UBSan.

csrc/adsb_multi.cpp -- the device threads, the lock-free step state machine (Step::state, p1_left / p2_left, the command
rings), dispatch_ready by whichever thread lands last, the collector, the replay pool behind it, the poisoned-handle rules
and the restart -- is compiled AS IT IS by g++ and linked against tests/multi_fake_backend.cpp, which fakes what lies below
it: the handful of HIP calls it makes itself and the shard_* entry points of csrc/adsb_shard.cpp, as a "device" that slices
its shard with the oracle, keeps the address superset the real context keeps, scores about half of its shards itself (as
k_score / k_emit do, here by the oracle's score_modes_message), and lands its phases from another thread after random delays.  The driver in that file runs random sequences -- 1-8 devices, up to four captures in flight, flushes,
host and device forms, spin and block waits, injected failures of every kind (the product's own hook and the fake's)
followed by the restart -- and compares every capture with ONE oracle stream; then it makes the n-th `operator new` from now
throw std::bad_alloc, on whichever thread it falls, around single library calls: nothing may cross the ABI, terminate or hang,
and after the restart the handle must give the oracle's list again.  CPU only; the same scenarios run on the
GPU through the real backend in tests/test_gpu_multi.py."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from tests.conftest import ROOT

CSRC = ROOT / "dump1090_rs_amd" / "csrc"
SOURCES = [ROOT / "tests" / "multi_fake_backend.cpp", CSRC / "adsb_multi.cpp", CSRC / "adsb_replay_host.cpp"]
ORACLE_C = ROOT / "oracle" / "dump1090_oracle.c"
HIP_INCLUDE = Path("/opt/rocm/include")
CHUNK = 131072


def make_arena(path: Path, n_chunks: int = 20, seed: int = 20261003) -> None:
    """Whole buffers of noise with Mode-S bursts of a dozen aircraft: extended squitters and all-call replies (what the
    filter learns from), short and long address/parity replies to the same aircraft (what only an address the exchange
    delivered lets through), a fifth of them hugging a buffer edge = often a shard boundary."""
    from dump1090_rs_amd import synth
    rng = np.random.default_rng(seed)
    icaos = [int(x) for x in rng.integers(1, 1 << 24, size=14)]
    n = n_chunks * CHUNK
    iq = synth.noise_numpy(n, seed=int(rng.integers(1, 1 << 30)))
    bursts = []
    for _ in range(9 * n_chunks):
        icao = icaos[int(rng.integers(0, len(icaos)))]
        kind = rng.random()
        if kind < 0.30:
            frame = synth.df17_frame(icao, int(rng.integers(0, 1 << 56)))
        elif kind < 0.42:
            frame = synth.df11_frame(icao)
        elif kind < 0.8:
            body = bytes([int(rng.choice([0x00, 0x20, 0x28])) | int(rng.integers(0, 8))]) + bytes(rng.integers(0, 256, 3).tolist())
            frame = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
        else:
            body = bytes([int(rng.choice([0x80, 0xA0, 0xA8])) | int(rng.integers(0, 8))]) + bytes(rng.integers(0, 256, 10).tolist())
            frame = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
        tick = int(rng.integers(2000, 5 * (n - 400)))
        if rng.random() < 0.2:
            tick = max(2000, 5 * (CHUNK * int(rng.integers(1, n_chunks)) - int(rng.integers(0, 330))) + int(rng.integers(0, 5)))
        bursts.append(synth.Burst(tick, int(rng.integers(6000, 30000)), int(rng.integers(0, 16)), frame))
    synth.add_bursts(iq, bursts)
    iq.tofile(path)


def build(exe: Path, sanitize: str) -> Path:
    hdrs = [CSRC / n for n in ("adsb_ctx.h", "adsb_device.h", "adsb_replay_host.h", "adsb_record.h", "mode_s_host.hpp")]
    hdrs += [ROOT / "include" / "adsb_hip.h", ROOT / "oracle" / "dump1090_oracle.h"]
    if exe.exists() and exe.stat().st_mtime >= max(p.stat().st_mtime for p in [*SOURCES, ORACLE_C, *hdrs]):
        return exe
    obj = exe.with_suffix(".oracle.o")
    # (the oracle uninstrumented: it is the checker, single-threaded per call, and slices ~2 ms per buffer this way)
    subprocess.run(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-c", str(ORACLE_C), "-o", str(obj)], check=True)
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror", f"-fsanitize={sanitize}", "-fno-omit-frame-pointer",
                    "-fno-sanitize-recover=all", "-pthread", "-D__HIP_PLATFORM_AMD__", f"-I{HIP_INCLUDE}", *map(str, SOURCES), str(obj),
                    "-lm", "-o", str(exe)], check=True)
    return exe


def have(lib: str) -> bool:
    p = subprocess.run(["gcc", f"-print-file-name={lib}"], capture_output=True, text=True).stdout.strip()
    return bool(p) and Path(p).exists() and shutil.which("g++") is not None and (HIP_INCLUDE / "hip" / "hip_runtime.h").exists()


@pytest.fixture(scope="module")
def arena(tmp_path_factory):
    path = tmp_path_factory.mktemp("multi_orchestration") / "arena.iq"
    make_arena(path)
    return path


def test_orchestration_under_thread_sanitizer(arena):
    """300 random sequences (about 2000 captures, 90 injected failures, 50 restarts, a dozen given-up devices), no report
    from the sanitizer, every capture equal to the oracle's."""
    if not have("libtsan.so"):
        pytest.skip("no libtsan / g++ / HIP headers in this environment")
    exe = build(ROOT / "tests" / "multi_orchestration_tsan", "thread")
    r = subprocess.run([str(exe), str(arena), "300"], capture_output=True, text=True, timeout=1500,
                       env=dict(__import__("os").environ, TSAN_OPTIONS="halt_on_error=1"))
    if "unexpected memory mapping" in r.stderr:   # (a kernel whose address-space layout this libtsan does not know)
        pytest.skip("ThreadSanitizer cannot run on this kernel")
    assert r.returncode == 0 and "multi orchestration ok: 300 sequences" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    # the run did what it is for: failures of both kinds were injected, handles restarted, some devices given up, both wait modes
    words = [ln for ln in r.stdout.splitlines() if ln.startswith("multi orchestration ok")][0].split()
    count = lambda what: int(words[words.index(what) - 1])   # noqa: E731
    assert count("failures") >= 40 and count("restarts,") >= 20 and count("dead") >= 3 and count("poisoned") >= 5
    assert count("blocking") >= 50 and int(words[words.index("captures,") - 1]) >= 1200
    # ... shards scored by their "device" were taken as they are, and refused (their records fetched by the shard's own thread)
    assert int(words[words.index("used,") - 6]) >= 300 and int(words[words.index("refused,") - 1]) >= 100
    assert int(words[words.index("resets,") - 2]) >= 5     # restarts whose reset failed first: the handle stayed poisoned
    assert int(words[words.index("creates") - 2]) >= 5     # adsb_multi_create with a context that cannot be made: undone, nothing left
    # ... and the allocation failures: std::bad_alloc from the n-th operator new on whichever thread, around single library
    # calls -- nothing thrown across the ABI, nothing terminated, every handle restarted and equal to the oracle again
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("allocation failures ok")]
    assert line, r.stdout[-1500:]
    w = line[0].split()
    assert int(w[w.index("armed") - 1]) >= 400 and int(w[w.index("allocations") - 1]) >= 100 and int(w[w.index("calls", 5) - 1]) >= 50


def test_orchestration_under_address_and_ub_sanitizers(arena):
    if not have("libasan.so"):
        pytest.skip("no libasan / g++ / HIP headers in this environment")
    exe = build(ROOT / "tests" / "multi_orchestration_asan", "address,undefined")
    # (detect_leaks=0: a handle whose device was given up leaks that device's context by design -- include/adsb_hip.h)
    r = subprocess.run([str(exe), str(arena), "80", "77"], capture_output=True, text=True, timeout=1500,
                       env=dict(__import__("os").environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and "multi orchestration ok: 80 sequences" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
