// Stand-alone driver of csrc/host_pool.h for the sanitizer legs of the CPU suite (tests/test_host_sanitizers.py):
// many caller threads copy buffers whose sizes straddle the pool's thresholds (below 2 pieces: plain memcpy; around
// 2 x PIECE; many pieces; a ragged tail) through the process-wide pool at the same time, and verify every byte.
// Built with -fsanitize=thread and with -fsanitize=address,undefined; the pool size comes from SDRK_HOST_THREADS
// (0, 1 and 7 helpers are run).  Exit code 0 = every copy was exact.
//
// What it answers: the reference shares state between its reader thread and Flask request threads with no
// synchronisation at all (app/sdr/streamer.py:19-21, app/dashboard/callbacks.py:19,96); this library's host side puts a
// thread pool under every sdrk_exec_host call, so that pool has to be clean under the tools that find races.
#include "../sdr-iq-visualizer_amd/csrc/host_pool.h"

#include <cstdint>
#include <cstdio>
#include <random>

int main(int argc, char** argv) {
    const int callers = argc > 1 ? atoi(argv[1]) : 6;
    const int rounds = argc > 2 ? atoi(argv[2]) : 12;
    constexpr size_t PIECE = 1u << 20;
    const size_t sizes[] = {0, 1, 4096, PIECE - 1, PIECE, 2 * PIECE - 1, 2 * PIECE, 2 * PIECE + 1, 3 * PIECE + 12345, 9 * PIECE + 7};
    sdrk::CopyPool& pool = sdrk::CopyPool::get();
    std::atomic<int> bad{0};
    std::vector<std::thread> ts;
    for (int c = 0; c < callers; ++c)
        ts.emplace_back([&, c] {
            std::mt19937 rng(1234u + (unsigned)c);
            std::vector<uint8_t> src(9 * PIECE + 64), dst(9 * PIECE + 64);
            for (int r = 0; r < rounds; ++r) {
                const size_t n = sizes[rng() % (sizeof sizes / sizeof sizes[0])];
                const size_t so = rng() % 57, dof = rng() % 57;                 // unaligned starts
                const uint8_t tag = (uint8_t)(rng() | 1u);
                for (size_t i = 0; i < n; ++i) src[so + i] = (uint8_t)(tag * (uint8_t)(i + 1) + (uint8_t)(i >> 8));
                std::fill(dst.begin(), dst.end(), (uint8_t)0);
                pool.copy(dst.data() + dof, src.data() + so, n);
                for (size_t i = 0; i < n; ++i)
                    if (dst[dof + i] != src[so + i]) { bad.fetch_add(1); break; }
                if (dof && dst[dof - 1] != 0) bad.fetch_add(1);                  // nothing written outside the range
                if (dst[dof + n] != 0) bad.fetch_add(1);
            }
        });
    for (auto& t : ts) t.join();
    printf("helpers=%d callers=%d rounds=%d bad=%d\n", pool.helpers(), callers, rounds, bad.load());
    return bad.load() ? 1 : 0;
}
