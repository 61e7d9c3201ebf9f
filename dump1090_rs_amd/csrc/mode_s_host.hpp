// mode_s_host.hpp -- the sequential tail of the hot path, kept on the host.
//
// The ICAO address filter is the one piece of state in demodulate2400 that makes
// trial N depend on trial N-1 (scoring reads AND writes it: reference
// src/mode_s/mod.rs:71,80-84,97-104,115,130), so the device hands back the few
// trials that can matter and this file replays them in the reference's order.
// Semantics follow src/icao_filter.rs, src/crc.rs and src/mode_s/mod.rs; the state
// is per-context instead of process-global.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>

namespace adsb {

// src/icao_filter.rs.  Table B of the reference is never written (only flushed),
// so probing it succeeds exactly for address 0 (an empty slot equals 0).
class IcaoFilter {
  public:
    static constexpr uint32_t kSize = 4096;           // :5
    static constexpr uint32_t kAdsbNt = 1u << 25;     // :6

    IcaoFilter() { flush(); }
    void flush() { table_.fill(0); }                  // :11-17

    static uint32_t hash(uint32_t addr)               // :19-43
    {
        uint64_t h = 0;
        const uint64_t a = addr;
        const uint64_t bytes[3] = {a & 0xff, (a >> 8) & 0xff, (a >> 16) & 0xff};
        for (uint64_t b : bytes) {
            h += b;
            h += h << 10;
            h ^= h >> 6;
        }
        h += h << 3;
        h ^= h >> 11;
        h += h << 15;
        return static_cast<uint32_t>(h) & (kSize - 1);
    }

    void add(uint32_t addr) { add(addr, hash(addr)); }  // :46-62
    // `start` = hash(addr), when the caller already has it (the device computes it per record)
    void add(uint32_t addr, uint32_t start)
    {
        uint32_t h = start;
        do {
            if (table_[h] == addr) return;
            if (table_[h] == 0) {
                table_[h] = addr;
                inserts_++;
                return;
            }
            h = (h + 1) & (kSize - 1);
        } while (h != start);
        // full: the reference prints "icao24 hash table full" and inserts nothing
    }

    bool test(uint32_t addr) const { return test(addr, hash(addr)); }  // :65-97
    bool test(uint32_t addr, uint32_t start) const
    {
        uint32_t h = start;
        while (table_[h] != 0 && table_[h] != addr) {
            h = (h + 1) & (kSize - 1);
            if (h == start) break;
        }
        return table_[h] == addr || addr == 0 /* table B */;
    }

    const std::array<uint32_t, kSize> &table() const { return table_; }
    // how many values add() has newly put into the table so far (never reset: callers compare)
    uint64_t inserts() const { return inserts_; }
    void load(const uint32_t *t) { std::memcpy(table_.data(), t, sizeof(uint32_t) * kSize); }
    void store(uint32_t *t) const { std::memcpy(t, table_.data(), sizeof(uint32_t) * kSize); }

  private:
    std::array<uint32_t, kSize> table_;
    uint64_t inserts_ = 0;
};

// src/crc.rs: Mode-S CRC-24, generator 0xFFF409.
struct Crc24 {
    uint32_t t[256];
    Crc24()
    {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i << 16;
            for (int k = 0; k < 8; k++) c = (c & 0x800000u) ? ((c << 1) ^ 0xFFF409u) : (c << 1);
            t[i] = c & 0xFFFFFFu;
        }
    }
    // :263-282; residual = syndrome of the first n-3 bytes XOR the last three
    uint32_t residual(const uint8_t *m, int nbytes) const
    {
        uint32_t rem = 0;
        for (int i = 0; i < nbytes - 3; i++)
            rem = ((rem << 8) ^ t[m[i] ^ ((rem >> 16) & 0xff)]) & 0xFFFFFFu;
        return rem ^ (uint32_t(m[nbytes - 3]) << 16 | uint32_t(m[nbytes - 2]) << 8 | m[nbytes - 1]);
    }
};

struct Score {
    bool some;    // false == the reference's None
    int len;      // 7 | 14
    int32_t value;
};

// src/mode_s/mod.rs:34-139 on a 14-byte trial message whose CRC residual (over its own length:
// 14 bytes for DF >= 16, else 7) is already known.
// `hash`: icao_hash of the value this message's DF asks the filter about (the residual for the
// address/parity DFs, the address for DF11/17/18) when the device supplied it, else -1.
// `Filter`: IcaoFilter, or a stand-in with the same test(addr, start) / add(addr, start) (the position-aware view of
// the parallel replay, adsb_replay_host.cpp).
template <class Filter>
inline Score score_modes_message(Filter &filter, uint32_t residual, const uint8_t msg[14], int hash = -1)
{
    const uint32_t df = msg[0] >> 3;                            // :41
    const int len = (df & 0x10) ? 14 : 7;                       // :42-46
    uint64_t w0, w1;                                            // :51-53 all 14 bytes zero -> None
    std::memcpy(&w0, msg, 8);
    std::memcpy(&w1, msg + 6, 8);
    if ((w0 | w1) == 0) return {false, len, 0};

    const uint32_t addr = uint32_t(msg[1]) << 16 | uint32_t(msg[2]) << 8 | msg[3];
    int32_t v = -2;
    switch (df) {
    case 0: case 4: case 5:                                     // :56-72
        v = filter.test(residual, hash >= 0 ? (uint32_t)hash : IcaoFilter::hash(residual)) ? 1000 : -1;
        break;
    case 11: {                                                  // :73-90
        const uint32_t c = residual;
        const uint32_t h = hash >= 0 ? (uint32_t)hash : IcaoFilter::hash(addr);
        const bool known = filter.test(addr, h);
        if ((c & 0xFFFF80u) != 0) {
            v = -2;
        } else if ((c & 0x7f) == 0) {
            if (known) {
                v = 1600;
            } else {
                filter.add(addr, h);
                v = 750;
            }
        } else {
            v = known ? 1000 : -1;
        }
        break;
    }
    case 17: case 18: {                                         // :91-109
        const uint32_t c = residual;
        const uint32_t h = hash >= 0 ? (uint32_t)hash : IcaoFilter::hash(addr);  // the hash takes 24 bits
        if (c != 0) {
            v = -2;
        } else if (filter.test(addr, h)) {
            v = 1800;
        } else {
            filter.add(df == 17 ? addr : (addr | IcaoFilter::kAdsbNt), h);
            v = 1400;
        }
        break;
    }
    case 16: case 20: case 21:                                  // :110-120
    case 24: case 25: case 26: case 27: case 28: case 29: case 30: case 31:  // :121-135
        v = filter.test(residual, hash >= 0 ? (uint32_t)hash : IcaoFilter::hash(residual)) ? 1000 : -2;
        break;
    default:
        v = -2;                                                 // :136
    }
    return {true, len, v};
}

// ... computing the residual here (src/crc.rs:263-282)
template <class Filter>
inline Score score_modes_message(Filter &filter, const Crc24 &crc, const uint8_t msg[14])
{
    return score_modes_message(filter, crc.residual(msg, (msg[0] & 0x80) ? 14 : 7), msg);
}

}  // namespace adsb
