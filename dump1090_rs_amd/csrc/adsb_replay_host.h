// adsb_replay_host.h -- the host-only half of the library: the ordered replay every demod call ends with and the
// small helpers around it.  Nothing here touches HIP (adsb_replay_host.cpp builds with plain g++ as well).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/adsb_hip.h"
#include "adsb_record.h"
#include "mode_s_host.hpp"

namespace adsb {
namespace host {

// Ordered replay (src/demod_2400.rs:149-207 with src/mode_s/mod.rs:34-139 scoring against src/icao_filter.rs):
// records in any order, replayed by (chunk, j, try_phase) and left where they are; `chunk_offset` is added to
// the messages' chunk.  host_sorts: counted up when the records had to be put in order here.
void replay(IcaoFilter &filter, const Crc24 &crc, const TrialRecord *rec, size_t n, uint64_t chunk_offset,
            std::vector<adsb_msg> &out, uint64_t *host_sorts = nullptr);

bool replay_order(const TrialRecord *rec, size_t n, std::vector<uint32_t> &order_out);
bool sort_records(const TrialRecord *rec, size_t n, std::vector<TrialRecord> &sorted_out);

// The sorted union of several sorted, duplicate-free address lists (the shards' learned addresses), appended to
// `out` (cleared first); what is already in `known` (sorted, duplicate-free) is left out.
void union_sorted(const std::vector<const std::vector<uint32_t> *> &lists, const std::vector<uint32_t> &known,
                  std::vector<uint32_t> &out);

// mode_s/mod.rs:80-84 (DF11, IID 0) and :97-99 (DF17): the addresses the replay of these records can add.
void learned_addresses(const Crc24 &crc, const TrialRecord *rec, size_t n, std::vector<uint32_t> &addrs);

}  // namespace host
}  // namespace adsb
