// adsb_feed -- the receive loop of dump1090_rs/src/main.rs:154-201 with the SDR replaced by a
// file or a pipe and the demodulation done by libadsb_hip: read 2.4 MSPS i16 IQ, demodulate,
// print every frame as "*<hex>;" and send "*<hex>;\n" to raw-TCP clients (port 30002 in the
// reference, main.rs:46) so that downstream tools (adsb_deku's radar, anything that speaks
// the dump1090 raw format) can consume it.
//
//   adsb_feed [--device N] [--port P] [--quiet] [--mem-order] [--buffers K] [--latency-ms T] [--readers R] <capture.iq | ->
//
// Input is the reference's capture format (src/utils.rs:8-20, save_test_data): little-endian
// i16 pairs, im first; --mem-order takes {re, im} pairs instead.  The stream is cut into
// 131072-sample buffers exactly as consecutive SDR reads would be (no carry-over between
// buffers, src/lib.rs:36-44); up to K of them (default 64) travel to the GPU per pass through the
// pinned ring (adsb_ring_*), so reading, the copy and the scan overlap.  A slot does not wait to
// fill up: T ms (default 100) after its first whole buffer was complete, the whole buffers read so
// far are submitted as a shorter pass -- a deadline, not an idle timer: a live 2.4 MSPS pipe delivers
// data all the time, a buffer every 55 ms, and would otherwise sit 3.5 s in a 64-buffer slot -- and
// the begun buffer moves on to the next slot, so the cuts stay where consecutive reads of 131072
// samples put them.  Raw-TCP clients are written without blocking: one whose socket buffer stays
// full for T ms (it stopped reading) is dropped, as the reference drops a client whose write fails
// (main.rs:184-200); one that is merely slower than a burst of output is waited for that long, so an
// offline replay, which produces lines far faster than real time, loses nobody who reads.  New
// clients are accepted while waiting for input too.  A pass with more frames than the output array gets them all
// (adsb_fetch_messages).  The ICAO filter is never flushed, as in the reference's loop.
// A regular file cannot trickle, so its slots are filled by R threads (default 4) that each pread() and
// swap their own part of the slot: one thread in read() copies out of the page cache at 6-10 GB/s, a
// fifth of what the ring takes.  Pipes and sockets are read by the one loop below.
// No GPU -> exits non-zero.
#include <arpa/inet.h>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <netinet/in.h>
#include <poll.h>
#include <condition_variable>
#include <mutex>
#include <string>
#include <sys/socket.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/adsb_hip.h"

namespace {

struct Clients {
    int listener = -1;
    std::vector<int> socks;
    unsigned long long dropped = 0;

    bool listen_on(int port)
    {
        listener = ::socket(AF_INET, SOCK_STREAM, 0);
        if (listener < 0) return false;
        int one = 1;
        ::setsockopt(listener, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in a{};
        a.sin_family = AF_INET;
        a.sin_addr.s_addr = htonl(INADDR_ANY);  // main.rs:38: 0.0.0.0 by default
        a.sin_port = htons((uint16_t)port);
        if (::bind(listener, (sockaddr *)&a, sizeof(a)) != 0 || ::listen(listener, 16) != 0) return false;
        ::fcntl(listener, F_SETFL, ::fcntl(listener, F_GETFL, 0) | O_NONBLOCK);  // main.rs:150
        return true;
    }
    void accept_new()  // main.rs:155-158
    {
        if (listener < 0) return;
        for (;;) {
            const int s = ::accept(listener, nullptr, nullptr);
            if (s < 0) break;
            // never block on a client: a generous send buffer for bursts, then EAGAIN means it is not reading
            ::fcntl(s, F_SETFL, ::fcntl(s, F_GETFL, 0) | O_NONBLOCK);
            int sndbuf = 1 << 20;  // ~30 000 frames of backlog (the kernel may grant less)
            ::setsockopt(s, SOL_SOCKET, SO_SNDBUF, &sndbuf, sizeof(sndbuf));
            socks.push_back(s);
        }
    }
    // main.rs:184-199: drop a client when its write fails.  A full socket buffer (EAGAIN) is given
    // `budget_ms` to drain -- the client is reading, only slower than this burst -- and counts as a
    // failed write after that (a partly written line could not be completed later without queueing
    // per client).
    void send_all(const std::string &lines, int budget_ms)
    {
        for (size_t i = 0; i < socks.size();) {
            size_t off = 0;
            bool dead = false;
            const auto t0 = std::chrono::steady_clock::now();
            while (off < lines.size()) {
                const ssize_t w = ::send(socks[i], lines.data() + off, lines.size() - off, MSG_NOSIGNAL | MSG_DONTWAIT);
                if (w <= 0) {
                    if (w < 0 && errno == EINTR) continue;
                    if (w < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) {
                        const long spent = (long)std::chrono::duration_cast<std::chrono::milliseconds>(
                                               std::chrono::steady_clock::now() - t0).count();
                        pollfd p{socks[i], POLLOUT, 0};
                        if (spent < budget_ms && ::poll(&p, 1, (int)(budget_ms - spent)) > 0 && (p.revents & POLLOUT)) continue;
                    }
                    dead = true;
                    dropped++;
                    break;
                }
                off += (size_t)w;
            }
            if (dead) {
                ::close(socks[i]);
                socks.erase(socks.begin() + (long)i);
            } else {
                i++;
            }
        }
    }
    ~Clients()
    {
        for (int s : socks) ::close(s);
        if (listener >= 0) ::close(listener);
    }
};

int die(adsb_ctx *ctx, const char *what, int st)
{
    std::fprintf(stderr, "adsb_feed: %s: %s %s\n", what, adsb_strerror(st), ctx ? adsb_last_error(ctx) : "");
    if (ctx) adsb_destroy(ctx);
    return 1;
}

// Wait up to idle_ms for input (idle_ms < 0: for ever) and read once into dst[have, want).
// Returns the new fill; *eof at end of input, *idle when nothing arrived in time.
size_t read_once(int fd, char *dst, size_t have, size_t want, int idle_ms, bool *eof, bool *idle)
{
    *idle = false;
    for (;;) {
        pollfd p{fd, POLLIN, 0};
        const int pr = ::poll(&p, 1, idle_ms);
        if (pr == 0) {
            *idle = true;
            return have;
        }
        if (pr < 0) {
            if (errno == EINTR) continue;
            *eof = true;
            return have;
        }
        const ssize_t r = ::read(fd, dst + have, want - have);
        if (r < 0 && (errno == EINTR || errno == EAGAIN)) continue;
        if (r <= 0) {
            *eof = true;
            return have;
        }
        return have + (size_t)r;
    }
}

// file pairs are [im][re]: swap into the in-memory {re, im} (utils.rs:29-31).  A 16-bit rotate of each pair as
// one word: the compiler vectorises it; pair by pair the swap was the slowest stage of the feed.
void swap_pairs(char *bytes, size_t n_pairs)
{
    typedef uint32_t __attribute__((may_alias)) pair_word;
    pair_word *w = reinterpret_cast<pair_word *>(bytes);
    for (size_t k = 0; k < n_pairs; k++) w[k] = (w[k] << 16) | (w[k] >> 16);
}

// Fills slots from a regular file with `n` threads (the caller is one of them): every thread pread()s its
// own 4-byte-aligned part of [offset, offset + bytes) into its part of the slot and swaps it there.
class FileFill {
public:
    FileFill(int fd, int n, bool swap) : fd_(fd), swap_(swap)
    {
        for (int k = 1; k < n; k++) workers_.emplace_back([this, k] { run(k); });
    }
    ~FileFill()
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
            round_++;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    // Reads up to `bytes` at `offset` into dst; returns what the file had (short only at its end).
    size_t fill(char *dst, off_t offset, size_t bytes)
    {
        const size_t parts = workers_.size() + 1;
        if (bytes < (size_t(1) << 20) || parts == 1) return part(dst, offset, bytes);
        {
            std::lock_guard<std::mutex> g(mu_);
            dst_ = dst, offset_ = offset, bytes_ = bytes;
            left_ = (int)workers_.size();
            got_.assign(parts, 0);
            round_++;
        }
        cv_.notify_all();
        const size_t mine = share(0, nullptr);
        got_[0] = part(dst, offset, mine);
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [this] { return left_ == 0; });
        size_t total = 0;   // the parts are contiguous: a short one ends the file
        for (size_t k = 0; k < parts; k++) {
            size_t want = 0;
            share(k, &want);
            total += got_[k];
            if (got_[k] < want) break;
        }
        return total;
    }

private:
    size_t share(size_t k, size_t *want) const   // part k = [k * per, min((k + 1) * per, bytes))
    {
        const size_t parts = workers_.size() + 1;
        const size_t per = ((bytes_ + parts - 1) / parts + 4095) & ~size_t(4095);
        const size_t lo = k * per < bytes_ ? k * per : bytes_;
        const size_t hi = lo + per < bytes_ ? lo + per : bytes_;
        if (want) *want = hi - lo;
        return k == 0 ? hi - lo : lo;
    }
    size_t part(char *dst, off_t offset, size_t bytes) const
    {
        size_t have = 0;
        while (have < bytes) {
            const ssize_t r = ::pread(fd_, dst + have, bytes - have, offset + (off_t)have);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) break;
            have += (size_t)r;
        }
        if (swap_) swap_pairs(dst, have / 4);   // (a capture ends on a whole pair or its last bytes are dropped below)
        return have;
    }
    void run(int k)
    {
        unsigned long long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> g(mu_);
            cv_.wait(g, [&] { return round_ != seen; });
            seen = round_;
            if (stop_) return;
            size_t want = 0;
            const size_t lo = share((size_t)k, &want);
            char *dst = dst_ + lo;
            const off_t at = offset_ + (off_t)lo;
            g.unlock();
            const size_t got = part(dst, at, want);
            g.lock();
            got_[(size_t)k] = got;
            if (--left_ == 0) done_.notify_one();
        }
    }
    int fd_;
    bool swap_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    unsigned long long round_ = 0;
    bool stop_ = false;
    char *dst_ = nullptr;
    off_t offset_ = 0;
    size_t bytes_ = 0;
    int left_ = 0;
    std::vector<size_t> got_;
};

}  // namespace

int main(int argc, char **argv)
{
    int device = 0, port = 0, buffers = 64, latency_ms = 100, out_cap = 0, readers = 4;
    bool quiet = false, mem_order = false;
    const char *path = nullptr;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--port" && i + 1 < argc) port = std::atoi(argv[++i]);
        else if (a == "--buffers" && i + 1 < argc) buffers = std::atoi(argv[++i]);
        else if (a == "--latency-ms" && i + 1 < argc) latency_ms = std::atoi(argv[++i]);
        else if (a == "--readers" && i + 1 < argc) readers = std::atoi(argv[++i]);
        else if (a == "--out-cap" && i + 1 < argc) out_cap = std::atoi(argv[++i]);  // frames the output array starts with (it grows)
        else if (a == "--quiet") quiet = true;
        else if (a == "--mem-order") mem_order = true;
        else if (a == "--help" || a == "-h") path = nullptr, i = argc;
        else path = argv[i];
    }
    if (!path || buffers < 1) {
        std::fprintf(stderr, "usage: adsb_feed [--device N] [--port P] [--quiet] [--mem-order] [--buffers K] [--latency-ms T] [--readers R] <capture.iq | ->\n");
        return 2;
    }
    std::signal(SIGPIPE, SIG_IGN);
    const int in = std::strcmp(path, "-") == 0 ? 0 : ::open(path, O_RDONLY);
    if (in < 0) {
        std::fprintf(stderr, "adsb_feed: cannot open %s: %s\n", path, std::strerror(errno));
        return 1;
    }
    Clients clients;
    if (port > 0 && !clients.listen_on(port)) {
        std::fprintf(stderr, "adsb_feed: cannot listen on port %d: %s\n", port, std::strerror(errno));
        return 1;
    }

    adsb_ctx *ctx = nullptr;
    int st = adsb_create(&ctx, device, (size_t)buffers);
    if (st != ADSB_OK) return die(nullptr, "adsb_create", st);
    const size_t slot_samples = (size_t)buffers * ADSB_MODES_MAG_BUF_SAMPLES;
    if ((st = adsb_ring_create(ctx, slot_samples)) != ADSB_OK) return die(ctx, "adsb_ring_create", st);

    // a 112 us frame every 120 us would be ~450 per 55 ms buffer; neighbouring preamble
    // positions can each emit one, so leave an order of magnitude of room
    std::vector<adsb_msg> out(out_cap > 0 ? (size_t)out_cap : (size_t)buffers * 4096);
    unsigned long long total_samples = 0, total_frames = 0;
    auto drain_one = [&]() -> int {
        size_t n = 0;
        int rc = adsb_collect(ctx, out.data(), out.size(), &n);
        if (rc == ADSB_ERR_CAPACITY) {  // more frames than the array holds: the context kept them all
            out.resize(n);
            rc = adsb_fetch_messages(ctx, out.data(), out.size(), &n);
        }
        if (rc != ADSB_OK) return rc;
        std::string lines;
        char line[40];
        for (size_t i = 0; i < n; i++) {
            const int len = adsb_format_raw(&out[i], line, sizeof(line));
            if (len > 0) lines.append(line, (size_t)len);
        }
        total_frames += n;
        if (!quiet && !lines.empty()) {
            std::fwrite(lines.data(), 1, lines.size(), stdout);
            std::fflush(stdout);
        }
        clients.accept_new();
        if (!lines.empty()) clients.send_all(lines, latency_ms);
        return ADSB_OK;
    };

    struct stat sb{};
    const bool regular = in != 0 && ::fstat(in, &sb) == 0 && S_ISREG(sb.st_mode);
    FileFill file_fill(in, regular ? (readers < 1 ? 1 : readers > 16 ? 16 : readers) : 1, !mem_order);
    off_t file_at = 0;

    const auto t_start = std::chrono::steady_clock::now();
    const size_t buf_bytes = (size_t)ADSB_MODES_MAG_BUF_SAMPLES * 4;
    std::vector<char> begun(buf_bytes);  // the buffer a short pass left unfinished: the next slot starts with it
    size_t begun_bytes = 0;
    unsigned long long short_passes = 0;
    bool eof = false;
    while (!eof) {
        int16_t *buf = nullptr;
        size_t cap = 0;
        st = adsb_ring_acquire(ctx, &buf, &cap);
        if (st == ADSB_ERR_BUSY) {  // every slot in flight: finish the oldest first
            if ((st = drain_one()) != ADSB_OK) return die(ctx, "adsb_collect", st);
            continue;
        }
        if (st != ADSB_OK) return die(ctx, "adsb_ring_acquire", st);
        char *dst = (char *)buf;
        size_t fill = begun_bytes;
        if (fill) std::memcpy(dst, begun.data(), fill);
        begun_bytes = 0;
        // until the slot is full, the input ends, or latency_ms have gone by since the slot's first whole
        // buffer was complete (a deadline: data that keeps arriving does not put it off); meanwhile
        // finished passes are handed on and new clients accepted
        bool have_whole = fill >= buf_bytes;
        auto t_whole = std::chrono::steady_clock::now();
        if (regular) {   // all of it at once, by all the readers, swapped where it lands
            const size_t got = file_fill.fill(dst, file_at, cap * 4);
            file_at += (off_t)got;
            fill = got;
            eof = got < cap * 4;
            clients.accept_new();
        }
        while (!regular && !eof && fill < cap * 4) {
            int wait_ms = -1;   // nothing to hand on and no whole buffer yet: as long as it takes
            if (have_whole) {
                const long spent = (long)std::chrono::duration_cast<std::chrono::milliseconds>(
                                       std::chrono::steady_clock::now() - t_whole).count();
                if (spent >= latency_ms) break;   // a short pass
                wait_ms = (int)(latency_ms - spent);
            } else if (adsb_pending(ctx) > 0 || clients.listener >= 0) {
                wait_ms = latency_ms;
            }
            bool idle = false;
            fill = read_once(in, dst, fill, cap * 4, wait_ms, &eof, &idle);
            if (!have_whole && fill >= buf_bytes) {
                have_whole = true;
                t_whole = std::chrono::steady_clock::now();
            }
            clients.accept_new();
            if (idle && adsb_pending(ctx) > 0)  // idle input: do not sit on results
                if ((st = drain_one()) != ADSB_OK) return die(ctx, "adsb_collect", st);
        }
        size_t bytes = fill;
        if (!eof && fill < cap * 4) {  // short pass: whole buffers only, the begun one waits for its rest
            bytes = fill / buf_bytes * buf_bytes;
            begun_bytes = fill - bytes;
            std::memcpy(begun.data(), dst + bytes, begun_bytes);
            short_passes++;
        }
        const size_t n = bytes / 4;
        if (!mem_order && !regular) swap_pairs(dst, n);
        if (n == 0) break;
        if ((st = adsb_ring_submit(ctx, n)) != ADSB_OK) return die(ctx, "adsb_ring_submit", st);
        total_samples += n;
        clients.accept_new();
    }
    while (adsb_pending(ctx) > 0)
        if ((st = drain_one()) != ADSB_OK) return die(ctx, "adsb_collect", st);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    std::fprintf(stderr, "adsb_feed: %llu samples, %llu frames in %.3f s (%.1f Msamples/s); %llu short passes, %llu clients dropped\n",
                 total_samples, total_frames, secs, secs > 0 ? total_samples / secs / 1e6 : 0.0, short_passes, clients.dropped);
    if (in != 0) ::close(in);
    adsb_destroy(ctx);
    return 0;
}
