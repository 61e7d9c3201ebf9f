// adsb_feed -- the receive loop of dump1090_rs/src/main.rs:154-201 with the SDR replaced by a
// file or a pipe and the demodulation done by libadsb_hip: read 2.4 MSPS i16 IQ, demodulate,
// print every frame as "*<hex>;" and send "*<hex>;\n" to raw-TCP clients (port 30002 in the
// reference, main.rs:46) so that downstream tools (adsb_deku's radar, anything that speaks
// the dump1090 raw format) can consume it.
//
//   adsb_feed [--device N] [--port P] [--quiet] [--mem-order] [--buffers K] <capture.iq | ->
//
// Input is the reference's capture format (src/utils.rs:8-20, save_test_data): little-endian
// i16 pairs, im first; --mem-order takes {re, im} pairs instead.  The stream is cut into
// 131072-sample buffers exactly as consecutive SDR reads would be (no carry-over between
// buffers, src/lib.rs:36-44); K of them (default 64) travel to the GPU per pass through the
// pinned double-buffered ring (adsb_ring_*), so reading, the copy and the scan overlap.
// The ICAO filter is never flushed, as in the reference's loop.  No GPU -> exits non-zero.
#include <arpa/inet.h>
#include <cerrno>
#include <chrono>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <netinet/in.h>
#include <string>
#include <sys/socket.h>
#include <unistd.h>
#include <vector>

#include "../../include/adsb_hip.h"

namespace {

struct Clients {
    int listener = -1;
    std::vector<int> socks;

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
            socks.push_back(s);
        }
    }
    void send_all(const std::string &lines)  // main.rs:184-199: drop a client when its write fails
    {
        for (size_t i = 0; i < socks.size();) {
            size_t off = 0;
            bool dead = false;
            while (off < lines.size()) {
                const ssize_t w = ::send(socks[i], lines.data() + off, lines.size() - off, MSG_NOSIGNAL);
                if (w <= 0) {
                    dead = true;
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

// read up to `want` bytes (short only at end of input)
size_t read_full(FILE *f, void *dst, size_t want)
{
    size_t got = 0;
    while (got < want) {
        const size_t r = std::fread((char *)dst + got, 1, want - got, f);
        if (r == 0) break;
        got += r;
    }
    return got;
}

}  // namespace

int main(int argc, char **argv)
{
    int device = 0, port = 0, buffers = 64;
    bool quiet = false, mem_order = false;
    const char *path = nullptr;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--port" && i + 1 < argc) port = std::atoi(argv[++i]);
        else if (a == "--buffers" && i + 1 < argc) buffers = std::atoi(argv[++i]);
        else if (a == "--quiet") quiet = true;
        else if (a == "--mem-order") mem_order = true;
        else if (a == "--help" || a == "-h") path = nullptr, i = argc;
        else path = argv[i];
    }
    if (!path || buffers < 1) {
        std::fprintf(stderr, "usage: adsb_feed [--device N] [--port P] [--quiet] [--mem-order] [--buffers K] <capture.iq | ->\n");
        return 2;
    }
    std::signal(SIGPIPE, SIG_IGN);
    FILE *in = std::strcmp(path, "-") == 0 ? stdin : std::fopen(path, "rb");
    if (!in) {
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
    std::vector<adsb_msg> out((size_t)buffers * 4096);
    unsigned long long total_samples = 0, total_frames = 0;
    auto drain_one = [&]() -> int {
        size_t n = 0;
        const int rc = adsb_collect(ctx, out.data(), out.size(), &n);
        if (rc == ADSB_ERR_CAPACITY) {  // the first out.size() were written
            std::fprintf(stderr, "adsb_feed: %zu frames in one pass, %zu dropped\n", n, n - out.size());
            n = out.size();
        } else if (rc != ADSB_OK) {
            return rc;
        }
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
        if (!lines.empty()) clients.send_all(lines);
        return ADSB_OK;
    };

    const auto t_start = std::chrono::steady_clock::now();
    bool eof = false;
    while (!eof) {
        int16_t *buf = nullptr;
        size_t cap = 0;
        st = adsb_ring_acquire(ctx, &buf, &cap);
        if (st == ADSB_ERR_BUSY) {  // both slots in flight: finish the oldest first
            if ((st = drain_one()) != ADSB_OK) return die(ctx, "adsb_collect", st);
            continue;
        }
        if (st != ADSB_OK) return die(ctx, "adsb_ring_acquire", st);
        const size_t bytes = read_full(in, buf, cap * 4);
        const size_t n = bytes / 4;
        eof = bytes < cap * 4;
        if (!mem_order)  // file pairs are [im][re]: swap into the in-memory {re, im} (utils.rs:29-31)
            for (size_t k = 0; k < n; k++) {
                const int16_t im = buf[2 * k];
                buf[2 * k] = buf[2 * k + 1];
                buf[2 * k + 1] = im;
            }
        if (n == 0) break;
        if ((st = adsb_ring_submit(ctx, n)) != ADSB_OK) return die(ctx, "adsb_ring_submit", st);
        total_samples += n;
        clients.accept_new();
    }
    while (adsb_pending(ctx) > 0)
        if ((st = drain_one()) != ADSB_OK) return die(ctx, "adsb_collect", st);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    std::fprintf(stderr, "adsb_feed: %llu samples, %llu frames in %.3f s (%.1f Msamples/s)\n", total_samples,
                 total_frames, secs, secs > 0 ? total_samples / secs / 1e6 : 0.0);
    if (in != stdin) std::fclose(in);
    adsb_destroy(ctx);
    return 0;
}
