"""Synthetic 2.4 MSPS i16 IQ for the bench and the parity tests (BASELINE.json configs 2-5).

Integer-only, so every box produces identical bytes:

* noise: sample n, component c (0 = re, 1 = im) is a sum of five 11-bit uniforms cut
  out of splitmix64(seed * 2^40 + 2n + c), minus 5118 -> roughly Gaussian, sigma ~ 1322
  (the reference fixtures sit at 1250..1640).
* frames: Mode-S pulse-position envelopes built on the 12 MHz grid (5 ticks per
  2.4 MHz sample; preamble pulses at 0, 1.0, 3.5, 4.5 us, data from 8 us, a 1 bit is
  high in its first half-microsecond), box-averaged to samples and added to I and Q
  with one of 16 fixed carrier angles.

`noise_numpy` / `noise_torch` give the same values; torch is used to fill a whole
256 MiB buffer on the GPU in milliseconds.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Sequence

import numpy as np

MASK64 = (1 << 64) - 1
SEED_DEFAULT = 0x10902400

# round(1024*cos(2*pi*a/16)), a = 0..15
_COS16 = [1024, 946, 724, 392, 0, -392, -724, -946, -1024, -946, -724, -392, 0, 392, 724, 946]
_SIN16 = _COS16[12:] + _COS16[:12]


# ----------------------------------------------------------------------------- noise
def _splitmix64_np(x: np.ndarray) -> np.ndarray:
    z = x + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def noise_numpy(n_samples: int, seed: int = SEED_DEFAULT, first_sample: int = 0) -> np.ndarray:
    """(n_samples, 2) int16 rows [re, im]."""
    with np.errstate(over="ignore"):
        idx = np.arange(2 * first_sample, 2 * (first_sample + n_samples), dtype=np.uint64)
        h = _splitmix64_np(idx + np.uint64((seed << 40) & MASK64))
        acc = np.zeros(idx.shape, dtype=np.int64)
        for t in range(5):
            acc += ((h >> np.uint64(11 * t)) & np.uint64(2047)).astype(np.int64)
    return (acc - 5118).astype(np.int16).reshape(-1, 2)


def _i64(v: int) -> int:
    v &= MASK64
    return v - (1 << 64) if v >= (1 << 63) else v


def noise_torch(n_samples: int, seed: int = SEED_DEFAULT, first_sample: int = 0, device="cpu",
                piece: int = 1 << 24):
    """Same values as noise_numpy, as a torch int16 tensor (n_samples, 2) on `device`."""
    import torch

    out = torch.empty((n_samples, 2), dtype=torch.int16, device=device)
    flat = out.view(-1)

    def lsr(z, k):  # logical shift right on int64
        return (z >> k) & ((1 << (64 - k)) - 1)

    for lo in range(0, 2 * n_samples, piece):
        hi = min(2 * n_samples, lo + piece)
        idx = torch.arange(2 * first_sample + lo, 2 * first_sample + hi, dtype=torch.int64, device=device)
        z = idx + _i64(seed << 40) + _i64(0x9E3779B97F4A7C15)
        z = (z ^ lsr(z, 30)) * _i64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * _i64(0x94D049BB133111EB)
        h = z ^ lsr(z, 31)
        acc = torch.zeros_like(h)
        for t in range(5):
            acc += lsr(h, 11 * t) & 2047 if t else h & 2047
        flat[lo:hi] = (acc - 5118).to(torch.int16)
    return out


# ----------------------------------------------------------------------------- Mode-S frames
def _crc_table() -> List[int]:
    tab = []
    for i in range(256):
        c = i << 16
        for _ in range(8):
            c = ((c << 1) ^ 0xFFF409) if c & 0x800000 else (c << 1)
        tab.append(c & 0xFFFFFF)
    return tab


_CRC = _crc_table()


def crc24(data: bytes) -> int:
    """Mode-S CRC-24 (generator 0xFFF409) of `data`."""
    rem = 0
    for b in data:
        rem = ((rem << 8) ^ _CRC[b ^ ((rem >> 16) & 0xFF)]) & 0xFFFFFF
    return rem


def df17_frame(icao: int, me: int) -> bytes:
    """A valid 112-bit extended squitter: DF17, CA=5, 24-bit address, 56-bit ME, parity."""
    body = bytes([0x8D]) + icao.to_bytes(3, "big") + (me & ((1 << 56) - 1)).to_bytes(7, "big")
    return body + crc24(body).to_bytes(3, "big")


def df11_frame(icao: int) -> bytes:
    """A valid 56-bit all-call reply: DF11, CA=5, address, parity with IID 0."""
    body = bytes([0x5D]) + icao.to_bytes(3, "big")
    return body + crc24(body).to_bytes(3, "big")


@dataclass(frozen=True)
class Burst:
    tick: int        # 12 MHz tick (5 per sample) at which the first preamble pulse starts
    amplitude: int   # peak |I + jQ| added, in i16 counts
    angle: int       # carrier angle index 0..15
    frame: bytes     # 7 or 14 bytes


def burst_envelope(frame: bytes) -> np.ndarray:
    """High/low per 12 MHz tick from the first preamble pulse to the end of the last bit."""
    nbits = 8 * len(frame)
    env = np.zeros(96 + 12 * nbits, dtype=np.int64)
    for start in (0, 12, 42, 54):  # 0, 1.0, 3.5, 4.5 us
        env[start:start + 6] = 1
    for n in range(nbits):
        bit = (frame[n >> 3] >> (7 - (n & 7))) & 1
        t0 = 96 + 12 * n + (0 if bit else 6)
        env[t0:t0 + 6] = 1
    return env


def add_bursts(iq: np.ndarray, bursts: Iterable[Burst], first_sample: int = 0) -> None:
    """Add bursts in place to (N, 2) int16 [re, im] rows (saturating at i16)."""
    n = iq.shape[0]
    for b in bursts:
        env = burst_envelope(b.frame)
        s0 = b.tick // 5
        lead = b.tick - 5 * s0
        padded = np.concatenate([np.zeros(lead, np.int64), env])
        padded = np.concatenate([padded, np.zeros((-len(padded)) % 5, np.int64)])
        per_sample = padded.reshape(-1, 5).sum(axis=1)  # 0..5 high ticks per sample
        lo = s0 - first_sample
        a, z = max(lo, 0), min(lo + len(per_sample), n)
        if a >= z:
            continue
        e = per_sample[a - lo:z - lo]
        # integer arithmetic only; // floors, which is well defined for negatives too
        di = (b.amplitude * _COS16[b.angle & 15] * e) // (5 * 1024)
        dq = (b.amplitude * _SIN16[b.angle & 15] * e) // (5 * 1024)
        seg = iq[a:z].astype(np.int64)
        seg[:, 0] += di
        seg[:, 1] += dq
        iq[a:z] = np.clip(seg, -32768, 32767).astype(np.int16)


def plan_bursts(n_samples: int, count: int, seed: int = SEED_DEFAULT, n_icao: int = 200,
                df11_every: int = 0) -> List[Burst]:
    """`count` non-overlapping bursts spread over n_samples, addresses drawn from a pool of
    `n_icao`; amplitude 8000..30000, every angle and sub-sample phase represented."""
    if count <= 0:
        return []
    x = np.arange(4 * count, dtype=np.uint64) + np.uint64(((seed ^ 0x5EED) << 20) & MASK64)
    with np.errstate(over="ignore"):
        r = _splitmix64_np(x).reshape(count, 4)
    slot = (5 * n_samples) // count          # ticks per burst slot
    span = 5 * 300                           # a burst covers < 300 samples
    if slot <= span + 5 * 8:
        raise ValueError("too many bursts for this many samples")
    out = []
    for i in range(count):
        jitter = int(r[i, 0] % np.uint64(slot - span))
        tick = i * slot + jitter
        icao = 0xA00000 + int(r[i, 1] % np.uint64(n_icao)) * 0x101
        amp = 8000 + int(r[i, 2] % np.uint64(22001))
        angle = int((r[i, 2] >> np.uint64(32)) % np.uint64(16))
        if df11_every and i % df11_every == df11_every - 1:
            frame = df11_frame(icao)
        else:
            frame = df17_frame(icao, int(r[i, 3]))
        out.append(Burst(tick, amp, angle, frame))
    return out


def make_iq(n_samples: int, n_bursts: int = 0, seed: int = SEED_DEFAULT, n_icao: int = 200,
            df11_every: int = 0) -> np.ndarray:
    """Noise + bursts on the CPU (numpy).  (n_samples, 2) int16 [re, im]."""
    iq = noise_numpy(n_samples, seed)
    add_bursts(iq, plan_bursts(n_samples, n_bursts, seed, n_icao, df11_every))
    return iq


def make_iq_torch(n_samples: int, n_bursts: int = 0, seed: int = SEED_DEFAULT, n_icao: int = 200,
                  df11_every: int = 0, device="cpu"):
    """Same bytes as make_iq, built on `device`: noise with torch, bursts patched in from
    the CPU (each burst touches < 300 samples)."""
    import torch

    iq = noise_torch(n_samples, seed, device=device)
    for b in plan_bursts(n_samples, n_bursts, seed, n_icao, df11_every):
        s0 = b.tick // 5
        a, z = max(s0, 0), min(s0 + 300, n_samples)
        if a >= z:
            continue
        patch = noise_numpy(z - a, seed, first_sample=a)  # identical to the torch values
        add_bursts(patch, [b], first_sample=a)
        iq[a:z] = torch.from_numpy(patch).to(device)
    return iq
