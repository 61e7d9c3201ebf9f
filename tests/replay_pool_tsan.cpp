// ThreadSanitizer driver for the host replay's threads (csrc/adsb_replay_host.h: ReplayPool + ParallelReplay), built and run by
// tests/test_host_sanitizers.py on the CPU.  Random captures of trial records (a few aircraft; DF17 / DF18 / DF11 with clean and
// with broken CRCs, interrogator ids, address/parity frames for known and unknown addresses, duplicates) are replayed
// serially and by the pool -- scan, merge, score, copy-out -- through ONE pool for hundreds of captures back to back, with
// filters that carry over from capture to capture and are flushed now and then: same messages, same table, and no report
// from the sanitizer.  The pool's size and the number of parts change from capture to capture (late wakers, stolen parts).
#include <cstdio>
#include <cstring>
#include <random>

#include "../dump1090_rs_amd/csrc/adsb_replay_host.h"

using namespace adsb;
using namespace adsb::host;

static void crc_fix(const Crc24 &crc, uint8_t *m, int nbytes, uint32_t xor_with)
{
    m[nbytes - 3] = m[nbytes - 2] = m[nbytes - 1] = 0;
    const uint32_t c = crc.residual(m, nbytes) ^ xor_with;
    m[nbytes - 3] = (uint8_t)(c >> 16), m[nbytes - 2] = (uint8_t)(c >> 8), m[nbytes - 1] = (uint8_t)c;
}

int main(int argc, char **argv)
{
    const int captures = argc > 1 ? std::atoi(argv[1]) : 300;
    std::mt19937_64 rng(20261002);
    const Crc24 crc;
    std::vector<uint32_t> aircraft;
    for (int k = 0; k < 40; k++) aircraft.push_back((uint32_t)(rng() % 0xFFFFFEu) + 1);
    static const uint32_t dfs[] = {0, 4, 5, 11, 11, 16, 17, 17, 17, 18, 20, 21, 24, 1};
    IcaoFilter serial, parallel;
    ReplayPool pool(5);
    ParallelReplay pr;
    size_t total_msgs = 0, went_parallel = 0;
    for (int cap = 0; cap < captures; cap++) {
        if (rng() % 7 == 0) serial.flush(), parallel.flush();
        const int runs_n = 1 + (int)(rng() % 5);
        std::vector<std::vector<TrialRecord>> shards((size_t)runs_n);
        for (auto &sh : shards) {
            const size_t n = rng() % 3000;
            for (size_t i = 0; i < n; i++) {
                TrialRecord r{};
                const uint32_t df = dfs[rng() % (sizeof(dfs) / sizeof(dfs[0]))];
                for (auto &b : r.msg) b = (uint8_t)rng();
                r.msg[0] = (uint8_t)(df << 3 | (rng() & 7));
                const int nbytes = df >= 16 ? 14 : 7;
                const uint32_t a = aircraft[rng() % aircraft.size()];
                if ((df == 11 || df == 17 || df == 18) && rng() % 10 < 8) {
                    r.msg[1] = (uint8_t)(a >> 16), r.msg[2] = (uint8_t)(a >> 8), r.msg[3] = (uint8_t)a;
                    crc_fix(crc, r.msg, nbytes, df == 11 && rng() % 4 == 0 ? (uint32_t)(rng() % 127 + 1) : 0u);
                } else if (rng() % 10 < 5) {
                    crc_fix(crc, r.msg, nbytes, a);   // an address/parity frame for one of the aircraft
                }
                r.chunk = (uint32_t)(rng() % 6);
                r.j_tp = (uint32_t)(rng() % 131072) | (uint32_t)(4 + rng() % 5) << 24;
                r.power = rng() & ((1ull << 38) - 1);
                if (rng() & 1) {   // as the device hands them over: residual and hash along
                    const uint32_t c = crc.residual(r.msg, (r.msg[0] & 0x80) ? 14 : 7);
                    const bool ap = ((0xFF310031u >> df) & 1u) != 0;
                    const uint32_t addr = uint32_t(r.msg[1]) << 16 | uint32_t(r.msg[2]) << 8 | r.msg[3];
                    r.power |= (uint64_t)c << 40;
                    r.pad = (uint16_t)(3u | IcaoFilter::hash(ap ? c : addr) << 4);
                }
                sh.push_back(r);
                if (rng() % 9 == 0) sh.push_back(r);   // twice
            }
            std::vector<TrialRecord> sorted;
            if (sort_records(sh.data(), sh.size(), sorted)) sh.swap(sorted);
        }
        std::vector<RecordRun> runs;
        uint64_t base = 0;
        for (auto &sh : shards) {
            runs.push_back({sh.data(), sh.size(), base});
            base += 6;
        }
        std::vector<adsb_msg> want, got;
        for (const RecordRun &r : runs) replay_sorted(serial, crc, r.rec, r.n, r.chunk_offset, want);
        const int parts = 2 + (int)(rng() % 40);
        std::vector<ParallelReplay::Adders> adders(runs.size());
        std::vector<const ParallelReplay::Adders *> adders_of;
        if (rng() % 3 == 0)   // the runs bring their first adders along (a device thread's work in adsb_multi)
            for (size_t k = 0; k < runs.size(); k++) {
                first_adders(crc, runs[k].rec, runs[k].n, adders[k]);
                adders_of.push_back(&adders[k]);
            }
        bool ok = pr.plan(parallel, crc, runs, parts, rng() & 1, adders_of.empty() ? nullptr : &adders_of);
        if (ok) {
            if (pr.scan_needed()) pool.run(pr, &ParallelReplay::scan_part);
            ok = pr.merge();
        }
        if (ok) {
            pool.run(pr, &ParallelReplay::score_part);
            if (rng() & 1) {
                got.resize(pr.message_count());
                pr.copy_to(got.data());
                pool.run(pr, &ParallelReplay::copy_part);
                pr.apply_adds(parallel);
            } else {
                pr.finish(parallel, got);
            }
            went_parallel++;
        } else {
            for (const RecordRun &r : runs) replay_sorted(parallel, crc, r.rec, r.n, r.chunk_offset, got);
        }
        if (want.size() != got.size() || (want.size() && std::memcmp(want.data(), got.data(), want.size() * sizeof(adsb_msg)) != 0) ||
            serial.table() != parallel.table()) {
            std::printf("capture %d: the pool's replay differs from the serial one (%zu messages against %zu)\n", cap, got.size(), want.size());
            return 1;
        }
        total_msgs += want.size();
    }
    std::printf("replay pool ok: %d captures, %zu messages, %zu replayed by the pool\n", captures, total_msgs, went_parallel);
    return went_parallel * 2 > (size_t)captures && total_msgs > 1000 ? 0 : 2;
}
