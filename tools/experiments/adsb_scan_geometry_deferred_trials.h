// adsb_scan_geometry.h -- tile geometry of the fast scan kernel (adsb_scan_fast.hip),
// shared with the host, which builds the field-addressing table for it.
#pragma once
#include "adsb_device.h"

namespace adsb {
namespace fastgeo {

// One workgroup = one tile of kTile preamble positions j of one chunk.
// 17 tiles of 7712 cover the 131072 positions of a chunk (the last one is short).
constexpr int kTile = 7712;
constexpr int kTilesPerChunk = (kChunkSamples + kTile - 1) / kTile;  // 17
constexpr int kPad = 2;   // 326 + 2 = 4 * 82: LDS slot 0 sits on a 16-byte IQ boundary
constexpr int kSlots = kTile + kPad + kReach;  // 8004 magnitudes a tile touches
static_assert(kTile % 4 == 0 && (kLead + kPad) % 4 == 0, "aligned dwordx4 IQ loads");

// sign planes: bit k of plane (kind, res) = decision at slot 12k + res
constexpr int kPlaneBits = (kSlots + 11) / 12;         // 667 per residue
constexpr int kPlaneBytes = (kPlaneBits + 7) / 8;      // 84
static_assert(kPlaneBytes % 4 == 0, "planes are whole dwords");
constexpr int kPlaneDw = kPlaneBytes / 4 + 1;          // 22: one dword of read slack (always zero)

// P5 runs whole passes only (twelve candidates x five trial phases = 60 of a wave's 64 lanes); the
// candidates a tile leaves over wait for the next tile's and run with them, on the planes of their
// own tile: the planes are double-buffered, and a row holds buffer 0, then buffer 1
// (adsb_scan_fast.hip).  -DADSB_DEFER_TRIALS=0 builds the form in which every tile finishes its own
// trials (one buffer), for A/B measurements.
#ifndef ADSB_DEFER_TRIALS
#define ADSB_DEFER_TRIALS 1
#endif
constexpr bool kDefer = ADSB_DEFER_TRIALS != 0;
constexpr int kPlaneBufs = kDefer ? 2 : 1;
constexpr int kRowDw = kPlaneBufs * kPlaneDw;          // dwords between plane rows

}  // namespace fastgeo

// bytes between plane rows in the fast scan's LDS (for the host-built field table)
inline uint32_t fast_plane_bytes() { return fastgeo::kRowDw * 4; }
}  // namespace adsb
