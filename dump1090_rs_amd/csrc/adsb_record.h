// adsb_record.h -- the one record type the device hands the host replay (no HIP in here: the host-only
// unit adsb_replay_host.cpp is also compiled by plain g++ under the sanitizers, tests/test_host_sanitizers.py).
#pragma once
#include <stdint.h>

namespace adsb {

// One trial message handed to the host replay (32 bytes).
struct TrialRecord {
    uint64_t power;     // bits 0..39: sum of the 33 squared magnitudes from j+19 (demod_2400.rs:
                        // 191-196; < 2^38).  With pad bit 0, bits 40..63: the CRC residual of msg
    uint32_t chunk;
    uint32_t j_tp;      // j | try_phase << 24
    uint8_t msg[14];
    uint16_t pad;       // bit 0: `power` carries the residual (records built on the device); bit 1: bits 4..15 are
                        // icao_hash of the value the DF asks the filter about (residual or address)
};
static_assert(sizeof(TrialRecord) == 32, "TrialRecord layout");

}  // namespace adsb
