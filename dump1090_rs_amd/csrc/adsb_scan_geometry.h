// adsb_scan_geometry.h -- tile geometry of the fast scan kernel (adsb_scan_fast.hip),
// shared with the host, which builds the field-addressing table for it.
#pragma once
#include "adsb_device.h"

namespace adsb {
namespace fastgeo {

// One workgroup = one tile of kTile preamble positions j of one chunk.
// 17 tiles of 7712 cover the 131072 positions of a chunk (the last one is short).
// (round 5: 18 tiles of 7284 -- 9216 tiles of a 256 MiB pass = exactly nine per workgroup of the persistent grid instead
// of eight or nine -- measured 5-6 % SLOWER, 105.7 against 100.9 us: a tile costs what its 256 threads' rounds cost, P2's
// 240 items and P3's 240 take the same one round as 252, so fewer positions per tile are simply less work per round;
// profiles/r5_tile_ab.txt.  -DADSB_TILE=7284 builds it.)
#ifndef ADSB_TILE
#define ADSB_TILE 7712
#endif
constexpr int kTile = ADSB_TILE;
constexpr int kTilesPerChunk = (kChunkSamples + kTile - 1) / kTile;  // 17
constexpr int kPad = 2;   // 326 + 2 = 4 * 82: LDS slot 0 sits on a 16-byte IQ boundary
constexpr int kSlots = kTile + kPad + kReach;  // 8004 magnitudes a tile touches
static_assert(kTile % 4 == 0 && (kLead + kPad) % 4 == 0, "aligned dwordx4 IQ loads");

// sign planes: bit k of plane (kind, res) = decision at slot 12k + res
constexpr int kPlaneBits = (kSlots + 11) / 12;         // 667 per residue
constexpr int kPlaneBytes = ((kPlaneBits + 7) / 8 + 3) & ~3;   // whole dwords (the slots beyond kSlots are loaded and sliced like any other; no position uses them)
static_assert(kPlaneBytes % 4 == 0, "planes are whole dwords");
constexpr int kPlaneDw = kPlaneBytes / 4 + 1;          // 22: one dword of read slack (always zero)

}  // namespace fastgeo

// bytes between plane rows in the fast scan's LDS (for the host-built field table)
inline uint32_t fast_plane_bytes() { return fastgeo::kPlaneDw * 4; }
}  // namespace adsb
