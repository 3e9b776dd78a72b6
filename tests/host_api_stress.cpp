// tests/host_api_stress.cpp — drives the HOST side of the library (csrc/sdrk_api.hip built with g++ against the stand-in
// runtime of tests/fake_hip) through its C ABI from several threads at once, for the sanitizer legs of the CPU suite
// (tests/test_host_sanitizers.py: -fsanitize=thread, and -fsanitize=address,undefined with leak checking).
//
// The stand-in "transform" is row[f][k] = 3 re - im + (k & 1023) (fake_kernels.cpp): every path of sdrk_exec_host — the
// mapped small call, the zero-copy chunks, the three-slot DMA pipeline with pageable, pinned and registered caller arrays,
// overlapped frames, ragged last chunks — must deliver exactly that for every element, whatever the threads around it do.
// Also: the complex epilogue, the chunked per-row feature path, the waterfall ring (rows, host IQ, device IQ enqueued
// without waiting, two-phase decimated read-out, the by-16 companion rows at N = 2^20), the placement probes, the Welch
// accumulation, invalid arguments, create / destroy churn.  Exit code 0 = every check passed.
//
// Reference sites this answers: the reference mutates shared state from its reader thread and from Flask request threads
// without synchronisation (app/sdr/streamer.py:19-21,100-101; app/dashboard/callbacks.py:19,96).
#include "../include/sdrk.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

static std::atomic<int> g_bad{0};
#define CHECK(cond)                                                                          \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            if (g_bad.fetch_add(1) < 20) fprintf(stderr, "CHECK failed %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, sdrk_last_error()); \
        }                                                                                    \
    } while (0)

struct c64 { float re, im; };

static void fill(std::vector<c64>& x, unsigned seed) {
    std::mt19937 rng(seed);
    for (auto& v : x) { v.re = (float)((int)(rng() & 0xFFF) - 2048); v.im = (float)((int)(rng() & 0xFFF) - 2048); }
}
static float want(const c64* x, size_t stride, size_t f, int k) { const c64 v = x[f * stride + (size_t)k]; return 3.0f * v.re - v.im + (float)(k & 1023); }

static void check_rows(const c64* x, size_t stride, const float* rows, size_t nf, int nfft) {
    size_t wrong = 0;
    for (size_t f = 0; f < nf; ++f)
        for (int k = 0; k < nfft; ++k) wrong += rows[f * (size_t)nfft + k] != want(x, stride, f, k);
    CHECK(wrong == 0);
}

// every size class of sdrk_exec_host on one plan, with pageable / pinned / registered arrays
static void exec_host_paths(int id, int rounds) {
    const int nfft = 4096;
    sdrk_plan* p = nullptr;
    CHECK(sdrk_plan_create_ex(0, nfft, 4096, SDRK_WINDOW_HANN, nullptr, 1e-12f, 1, id == 0 ? SDRK_PLAN_TUNE_STAGING : 0u, &p) == SDRK_OK);
    if (!p) return;
    const size_t sizes[] = {1, 8, 9, 64, 1030, 1500};        // 32 KiB (mapped small call) ... 47 MiB (three-slot pipeline, ragged tail)
    std::vector<c64> x(1500 * (size_t)nfft);
    std::vector<float> out(1500 * (size_t)nfft);
    for (int r = 0; r < rounds; ++r) {
        fill(x, 100u * (unsigned)id + (unsigned)r);
        for (size_t nf : sizes) {
            std::fill(out.begin(), out.begin() + (long)(nf * nfft), -1.0f);
            CHECK(sdrk_exec_host(p, x.data(), nf, nfft, out.data()) == SDRK_OK);
            check_rows(x.data(), nfft, out.data(), nf, nfft);
        }
        // overlapped frames cut from one stream (hop = nfft / 2)
        const size_t nf = 700;
        CHECK(sdrk_exec_host(p, x.data(), nf, nfft / 2, out.data()) == SDRK_OK);
        check_rows(x.data(), nfft / 2, out.data(), nf, nfft);
        // pinned input, pageable rows; then both pinned (<= 32 MiB: one launch; larger: DMA straight to the caller's rows)
        void *hx = nullptr, *ho = nullptr;
        CHECK(sdrk_host_alloc(1500 * (size_t)nfft * sizeof(c64), &hx) == SDRK_OK && sdrk_host_alloc(1500 * (size_t)nfft * 4, &ho) == SDRK_OK);
        if (hx && ho) {
            memcpy(hx, x.data(), 1500 * (size_t)nfft * sizeof(c64));
            CHECK(sdrk_host_is_pinned(hx, 4096) == 1 && sdrk_host_is_pinned(static_cast<char*>(hx) + 100, 64) == 1);
            for (size_t nf2 : {(size_t)300, (size_t)1500}) {
                CHECK(sdrk_exec_host(p, hx, nf2, nfft, out.data()) == SDRK_OK);
                check_rows(static_cast<const c64*>(hx), nfft, out.data(), nf2, nfft);
                memset(ho, 0, nf2 * (size_t)nfft * 4);
                CHECK(sdrk_exec_host(p, hx, nf2, nfft, static_cast<float*>(ho)) == SDRK_OK);
                check_rows(static_cast<const c64*>(hx), nfft, static_cast<const float*>(ho), nf2, nfft);
            }
        }
        CHECK(sdrk_host_free(hx) == SDRK_OK && sdrk_host_free(ho) == SDRK_OK);
        CHECK(sdrk_host_free(hx) != SDRK_OK);                       // not ours any more
        // a registered array of the caller
        CHECK(sdrk_host_register(out.data(), out.size() * 4) == SDRK_OK);
        CHECK(sdrk_host_register(out.data(), out.size() * 4) != SDRK_OK);   // twice: refused
        CHECK(sdrk_exec_host(p, x.data(), 1400, nfft, out.data()) == SDRK_OK);
        check_rows(x.data(), nfft, out.data(), 1400, nfft);
        CHECK(sdrk_host_unregister(out.data()) == SDRK_OK);
        // the complex epilogue through the same pipeline
        std::vector<c64> spec(1100 * (size_t)nfft);
        CHECK(sdrk_exec_fft_host(p, x.data(), 1100, nfft, spec.data()) == SDRK_OK);
        size_t wrong = 0;
        for (size_t i = 0; i < spec.size(); ++i) wrong += spec[i].re != x[i].re + 1.0f || spec[i].im != x[i].im - 1.0f;
        CHECK(wrong == 0);
        // invalid arguments never touch anything
        CHECK(sdrk_exec_host(p, nullptr, 4, nfft, out.data()) == SDRK_ERR_INVALID);
        CHECK(sdrk_exec_host(p, x.data(), 5000, nfft, out.data()) == SDRK_ERR_INVALID);   // beyond max_batch
        CHECK(sdrk_exec_host(p, x.data(), 0, nfft, out.data()) == SDRK_OK);
    }
    CHECK(sdrk_plan_destroy(p) == SDRK_OK);
}

// the chunked per-row feature path (> 32 MiB of IQ) and its planes form
static void features_paths(int id) {
    const int nfft = 4096, mp = 8;
    const size_t nf = 1200;
    sdrk_plan* p = nullptr;
    CHECK(sdrk_plan_create(0, nfft, nf, SDRK_WINDOW_RECT, nullptr, 1e-12f, 1, &p) == SDRK_OK);
    if (!p) return;
    std::vector<c64> x(nf * (size_t)nfft);
    fill(x, 7000u + (unsigned)id);
    std::vector<double> stats(nf * 16), thr(nf), planes(nf * SDRK_FEAT_PLANES);
    std::vector<int32_t> idx(nf * mp), cnt(nf);
    std::vector<float> rows(nf * (size_t)nfft);
    CHECK(sdrk_frame_features_host(p, x.data(), nf, nfft, 800, 0.25f, 13, mp, stats.data(), thr.data(), idx.data(), cnt.data(), rows.data()) == SDRK_OK);
    check_rows(x.data(), nfft, rows.data(), nf, nfft);
    size_t wrong = 0;
    for (size_t f = 0; f < nf; ++f) {
        float mx = -INFINITY;
        for (int k = 0; k < nfft; ++k) mx = std::fmax(mx, rows[f * (size_t)nfft + k]);
        wrong += stats[f * 16] != (double)mx || stats[f * 16 + 15] != nfft || thr[f] != (double)mx - 1.0 || cnt[f] != 0 || idx[f * mp] != -1;
    }
    CHECK(wrong == 0);
    CHECK(sdrk_frame_features_host_planes(p, x.data(), nf, nfft, 800, 0.25f, 13, mp, nullptr, planes.data(), idx.data(), nullptr) == SDRK_OK);
    wrong = 0;
    for (size_t f = 0; f < nf; ++f) wrong += planes[f] != stats[f * 16];
    CHECK(wrong == 0);
    CHECK(sdrk_row_features(0, rows.data(), 0, 64, nfft, 800, 0.25f, 13, mp, stats.data(), thr.data(), idx.data(), cnt.data()) == SDRK_OK);
    CHECK(sdrk_plan_destroy(p) == SDRK_OK);
}

// the ring: deque(maxlen) semantics under appends of rows / host IQ / device IQ, plain and two-phase decimated reads
static void waterfall_paths(int id) {
    const int nfft = 4096, L = 5;
    sdrk_waterfall* wf = nullptr;
    sdrk_plan* p = nullptr;
    CHECK(sdrk_waterfall_create(0, nfft, L, &wf) == SDRK_OK && sdrk_plan_create(0, nfft, 64, SDRK_WINDOW_RECT, nullptr, 1e-12f, 1, &p) == SDRK_OK);
    if (!wf || !p) return;
    std::vector<c64> x(12 * (size_t)nfft);
    fill(x, 9000u + (unsigned)id);
    std::vector<float> ref(12 * (size_t)nfft), got(L * (size_t)nfft), dec(L * (size_t)(nfft / 16));
    for (size_t f = 0; f < 12; ++f)
        for (int k = 0; k < nfft; ++k) ref[f * (size_t)nfft + k] = want(x.data(), nfft, f, k);
    size_t n = 0;
    CHECK(sdrk_waterfall_append_iq(wf, p, x.data(), 3, nfft) == SDRK_OK && sdrk_waterfall_rows(wf) == 3);
    CHECK(sdrk_waterfall_append_iq(wf, p, x.data() + 3 * (size_t)nfft, 4, nfft) == SDRK_OK && sdrk_waterfall_rows(wf) == L);   // wraps: frames 2..6
    CHECK(sdrk_waterfall_read(wf, got.data(), L, &n) == SDRK_OK && n == (size_t)L);
    CHECK(memcmp(got.data(), ref.data() + 2 * (size_t)nfft, L * (size_t)nfft * 4) == 0);
    // device IQ, enqueued without waiting, then the two-phase decimated read-out while the next batch is enqueued
    void* d_x = nullptr;
    CHECK(sdrk_dev_alloc(0, x.size() * sizeof(c64), &d_x) == SDRK_OK && sdrk_memcpy_h2d(0, d_x, x.data(), x.size() * sizeof(c64)) == SDRK_OK);
    CHECK(sdrk_waterfall_append_iq_device_async(wf, p, static_cast<c64*>(d_x) + 7 * (size_t)nfft, 2, nfft) == SDRK_OK);           // frames 7, 8
    std::vector<float> dec2(L * (size_t)(nfft / 16));
    CHECK(sdrk_waterfall_read_decimated_begin(wf, dec.data(), 2, 16, 0, &n) == SDRK_OK && n == 2);
    CHECK(sdrk_waterfall_append_iq_device_async(wf, p, static_cast<c64*>(d_x) + 9 * (size_t)nfft, 3, nfft) == SDRK_OK);           // frames 9..11 meanwhile
    CHECK(sdrk_waterfall_read_decimated_begin(wf, dec2.data(), 3, 16, 0, &n) == SDRK_OK && n == 3);                               // a second one in flight
    CHECK(sdrk_waterfall_read_decimated_begin(wf, dec2.data(), 3, 16, 0, &n) == SDRK_ERR_INVALID);                                // ... a third is refused
    CHECK(sdrk_waterfall_read_decimated(wf, dec2.data(), 1, 16, 0, &n) == SDRK_ERR_INVALID);                                      // and so is the one-call form
    CHECK(sdrk_waterfall_read_decimated_end(wf) == SDRK_OK);                                                                      // the OLDEST: frames 7, 8
    size_t wrong = 0;
    for (int r = 0; r < 2; ++r)
        for (int b = 0; b < nfft / 16; ++b) {
            float mx = -INFINITY;
            for (int i = 0; i < 16; ++i) mx = std::fmax(mx, ref[(size_t)(7 + r) * nfft + (size_t)b * 16 + i]);
            wrong += dec[(size_t)r * (nfft / 16) + b] != mx;
        }
    CHECK(wrong == 0);
    CHECK(sdrk_waterfall_read_decimated_end(wf) == SDRK_OK);                                                                      // then frames 9..11
    for (int r = 0; r < 3; ++r)
        for (int b = 0; b < nfft / 16; ++b) {
            float mx = -INFINITY;
            for (int i = 0; i < 16; ++i) mx = std::fmax(mx, ref[(size_t)(9 + r) * nfft + (size_t)b * 16 + i]);
            wrong += dec2[(size_t)r * (nfft / 16) + b] != mx;
        }
    CHECK(wrong == 0);
    CHECK(sdrk_waterfall_read_decimated_end(wf) == SDRK_OK);                                                                      // none left: no-op
    CHECK(sdrk_waterfall_sync(wf, p) == SDRK_OK);
    CHECK(sdrk_waterfall_read(wf, got.data(), L, &n) == SDRK_OK && n == (size_t)L);
    CHECK(memcmp(got.data(), ref.data() + 7 * (size_t)nfft, L * (size_t)nfft * 4) == 0);                                          // frames 7..11
    CHECK(sdrk_waterfall_append_rows(wf, ref.data(), 2) == SDRK_OK);                                                              // finished rows 0, 1
    CHECK(sdrk_waterfall_read(wf, got.data(), 2, &n) == SDRK_OK && n == 2 && memcmp(got.data(), ref.data(), 2 * (size_t)nfft * 4) == 0);
    CHECK(sdrk_waterfall_read_decimated(wf, dec.data(), 1, 4096, 1, &n) == SDRK_OK && n == 1);
    CHECK(sdrk_waterfall_clear(wf) == SDRK_OK && sdrk_waterfall_rows(wf) == 0 && sdrk_waterfall_maxhold16_rows(wf) == 0);
    CHECK(sdrk_dev_free(0, d_x) == SDRK_OK);
    CHECK(sdrk_waterfall_destroy(wf) == SDRK_OK && sdrk_plan_destroy(p) == SDRK_OK);
}

// N = 2^20: the by-16 companion rows beside the ring and the read-outs they serve
static void waterfall_companion_rows() {
    const int nfft = 1 << 20, L = 3;
    sdrk_waterfall* wf = nullptr;
    sdrk_plan* p = nullptr;
    CHECK(sdrk_waterfall_create(0, nfft, L, &wf) == SDRK_OK && sdrk_plan_create(0, nfft, 4, SDRK_WINDOW_RECT, nullptr, 1e-12f, 1, &p) == SDRK_OK);
    if (!wf || !p) return;
    std::vector<c64> x(4 * (size_t)nfft);
    fill(x, 424242u);
    CHECK(sdrk_waterfall_append_iq(wf, p, x.data(), 4, nfft) == SDRK_OK);              // 4 frames into 3 slots: 1..3 remain, ring wrapped
    CHECK(sdrk_waterfall_rows(wf) == L && sdrk_waterfall_maxhold16_rows(wf) == L);
    std::vector<float> full(L * (size_t)nfft), dec(L * 4096);
    size_t n = 0;
    CHECK(sdrk_waterfall_read(wf, full.data(), L, &n) == SDRK_OK && n == (size_t)L);
    check_rows(x.data() + nfft, nfft, full.data(), L, nfft);
    for (int factor : {16, 256, 4096}) {
        CHECK(sdrk_waterfall_read_decimated(wf, dec.data(), L, factor >= 256 ? factor : 256, 0, &n) == SDRK_OK && n == (size_t)L);
        const int f2 = factor >= 256 ? factor : 256, bins = nfft / f2;
        size_t wrong = 0;
        for (int r = 0; r < L; ++r)
            for (int b = 0; b < bins; ++b) {
                float mx = -INFINITY;
                for (int i = 0; i < f2; ++i) mx = std::fmax(mx, full[(size_t)r * nfft + (size_t)b * f2 + i]);
                wrong += dec[(size_t)r * bins + b] != mx;
            }
        CHECK(wrong == 0);
    }
    CHECK(sdrk_waterfall_append_rows(wf, full.data(), 1) == SDRK_OK && sdrk_waterfall_maxhold16_rows(wf) == L - 1);   // a finished row carries none
    CHECK(sdrk_waterfall_read_decimated(wf, dec.data(), L, 256, 0, &n) == SDRK_OK && n == (size_t)L);             // falls back to the rows
    CHECK(sdrk_waterfall_destroy(wf) == SDRK_OK && sdrk_plan_destroy(p) == SDRK_OK);
}

static void placement_and_misc() {
    // resident pair with the no-arithmetic probe, scratch placement of a two-pass plan, Welch accumulation, synth, probes
    void *d_in = nullptr, *d_out = nullptr;
    float probe[3] = {0, 0, 0};
    int chosen = -1;
    CHECK(sdrk_dev_alloc_stream_pair(0, (size_t)1 << 28, (size_t)1 << 27, 3, nullptr, &d_in, &d_out, probe, &chosen) == SDRK_OK);
    CHECK(d_in && d_out && chosen >= 0 && chosen < 3);
    float warm = 0, first = 0, again = 0, kept = 0;
    int launches = 0;
    CHECK(sdrk_placement_report(&warm, &launches, &first, &again, &kept) == 3 && warm >= 55.0f && launches >= 2 && again > 0 && kept > 0);
    CHECK(sdrk_dev_free(0, d_in) == SDRK_OK && sdrk_dev_free(0, d_out) == SDRK_OK);
    sdrk_plan* p = nullptr;
    CHECK(sdrk_plan_create(0, 65536, 8, SDRK_WINDOW_HANN, nullptr, 1e-12f, 1, &p) == SDRK_OK);
    if (p) {
        std::vector<c64> x(6 * 65536);
        fill(x, 5u);
        void *dx = nullptr, *dr = nullptr;
        CHECK(sdrk_dev_alloc(0, x.size() * sizeof(c64), &dx) == SDRK_OK && sdrk_dev_alloc(0, 6 * 65536 * 4, &dr) == SDRK_OK);
        CHECK(sdrk_memcpy_h2d(0, dx, x.data(), x.size() * sizeof(c64)) == SDRK_OK);
        float ms[4];
        CHECK(sdrk_plan_tune_scratch(p, dx, 6, 65536, static_cast<float*>(dr), 4, ms, &chosen) == SDRK_OK && chosen >= 0 && chosen < 4);
        CHECK(sdrk_placement_report(nullptr, nullptr, &first, &again, nullptr) == 4 && first > 0 && again > 0);
        float each[3];
        CHECK(sdrk_exec_device_timed_each(p, dx, 6, 65536, static_cast<float*>(dr), 3, each) == SDRK_OK);
        std::vector<float> rows(6 * 65536);
        CHECK(sdrk_memcpy_d2h(0, rows.data(), dr, rows.size() * 4) == SDRK_OK);
        check_rows(x.data(), 65536, rows.data(), 6, 65536);
        CHECK(sdrk_dev_free(0, dx) == SDRK_OK && sdrk_dev_free(0, dr) == SDRK_OK && sdrk_plan_destroy(p) == SDRK_OK);
    }
    CHECK(sdrk_plan_create(0, 1024, 64, SDRK_WINDOW_HANN, nullptr, 1e-12f, 1, &p) == SDRK_OK);
    if (p) {
        std::vector<c64> x(40 * 1024);
        fill(x, 6u);
        std::vector<float> psd(1024);
        CHECK(sdrk_welch_psd_host(p, x.data(), 40, 1024, 0.5f, psd.data()) == SDRK_OK);
        size_t wrong = 0;
        for (int k = 0; k < 1024; ++k) {
            float acc = 0.0f;
            for (int f = 0; f < 40; ++f) { const c64 v = x[(size_t)f * 1024 + k]; acc += (v.re + 1) * (v.re + 1) + (v.im - 1) * (v.im - 1); }
            wrong += std::fabs(psd[k] - 0.5f * acc) > 1e-3f * std::fabs(acc);
        }
        CHECK(wrong == 0);
        CHECK(sdrk_plan_destroy(p) == SDRK_OK);
    }
    // a chirp-z plan (its inner power-of-two plan, work buffers) and the experimental plan kinds: create, use, destroy
    CHECK(sdrk_plan_create(0, 1000, 16, SDRK_WINDOW_RECT, nullptr, 0.0f, 0, &p) == SDRK_OK);
    if (p) {
        std::vector<c64> x(16 * 1000);
        std::vector<float> rows(16 * 1000);
        fill(x, 8u);
        CHECK(sdrk_exec_host(p, x.data(), 16, 1000, rows.data()) == SDRK_OK);
        CHECK(sdrk_plan_destroy(p) == SDRK_OK);
    }
    for (unsigned flags : {SDRK_PLAN_FUSED64K, SDRK_PLAN_OVERLAP_PASSES}) {
        CHECK(sdrk_plan_create_ex(0, 65536, 4, SDRK_WINDOW_RECT, nullptr, 1e-12f, 1, flags, &p) == SDRK_OK);
        if (p) {
            std::vector<c64> x(4 * 65536);
            std::vector<float> rows(4 * 65536);
            fill(x, 9u);
            CHECK(sdrk_exec_host(p, x.data(), 4, 65536, rows.data()) == SDRK_OK);
            check_rows(x.data(), 65536, rows.data(), 4, 65536);
            CHECK(sdrk_plan_destroy(p) == SDRK_OK);
        }
    }
    CHECK(sdrk_plan_create(0, 3, 0, 0, nullptr, 1e-12f, 1, &p) == SDRK_ERR_INVALID && p == nullptr);
    CHECK(sdrk_plan_create_ex(0, 4096, 1, 0, nullptr, 1e-12f, 1, 0x80u, &p) == SDRK_ERR_INVALID);
    CHECK(sdrk_plan_create(7, 4096, 1, 0, nullptr, 1e-12f, 1, &p) == SDRK_ERR_NO_DEVICE);
    double a, b, c;
    CHECK(sdrk_host_link_probe(0, (size_t)4 << 20, &a, &b, &c) == SDRK_OK);
}

#include <hip/hip_runtime.h>   // the stand-in (tests/fake_hip): fakehip::cus()
namespace sdrk { extern std::atomic<int> g_fake_fused_fail, g_fake_fused_launches; }

// The default nfft = 65536 plan on a device whose CUs make whole sets: the persistent launch from 512 frames, the two tiled
// launches below; a launch that reports a failed hand-over fails the call and latches the plan onto the two launches; two plans
// on two streams (threads here) pass through the device-wide gate.
static void fused_auto_policy() {
    fakehip::cus() = 32;
    const int n = 65536, nf = 512;
    std::vector<c64> x((size_t)n + nf);                                   // frames at stride 1: 512 frames from 66 048 samples
    fill(x, 77u);
    void *dx = nullptr, *dr = nullptr;
    CHECK(sdrk_dev_alloc(0, x.size() * sizeof(c64), &dx) == SDRK_OK && sdrk_dev_alloc(0, (size_t)nf * n * 4, &dr) == SDRK_OK);
    CHECK(sdrk_memcpy_h2d(0, dx, x.data(), x.size() * sizeof(c64)) == SDRK_OK);
    std::vector<float> rows((size_t)nf * n);
    sdrk_plan *p = nullptr, *forced_tiled = nullptr;
    CHECK(sdrk_plan_create(0, n, 1024, SDRK_WINDOW_RECT, nullptr, 1e-12f, 1, &p) == SDRK_OK);
    CHECK(sdrk_plan_create_ex(0, n, 1024, SDRK_WINDOW_RECT, nullptr, 1e-12f, 1, SDRK_PLAN_TILED64K, &forced_tiled) == SDRK_OK);
    sdrk_plan* bad = nullptr;
    CHECK(sdrk_plan_create_ex(0, n, 8, 0, nullptr, 1e-12f, 1, SDRK_PLAN_TILED64K | SDRK_PLAN_FUSED64K, &bad) == SDRK_ERR_INVALID && !bad);
    CHECK(sdrk_plan_create_ex(0, 4096, 8, 0, nullptr, 1e-12f, 1, SDRK_PLAN_TILED64K, &bad) == SDRK_ERR_INVALID && !bad);
    unsigned launches = 99;
    int fell = 9;
    if (p && forced_tiled) {
        const int before = sdrk::g_fake_fused_launches.load();
        CHECK(sdrk_exec_device(p, dx, nf - 1, 1, static_cast<float*>(dr), nullptr) == SDRK_OK && sdrk_plan_sync(p) == SDRK_OK);
        CHECK(sdrk_plan_fused_status(p, &launches, &fell) == SDRK_OK && launches == 0 && fell == 0);
        CHECK(sdrk_exec_device(p, dx, nf, 1, static_cast<float*>(dr), nullptr) == SDRK_OK && sdrk_plan_sync(p) == SDRK_OK);
        CHECK(sdrk_plan_fused_status(p, &launches, &fell) == SDRK_OK && launches == 1 && fell == 0);
        CHECK(sdrk_memcpy_d2h(0, rows.data(), dr, rows.size() * 4) == SDRK_OK);
        check_rows(x.data(), 1, rows.data(), nf, n);
        CHECK(sdrk_exec_device(forced_tiled, dx, nf, 1, static_cast<float*>(dr), nullptr) == SDRK_OK && sdrk_plan_sync(forced_tiled) == SDRK_OK);
        CHECK(sdrk_plan_fused_status(forced_tiled, &launches, &fell) == SDRK_OK && launches == 0 && fell == 0);
        // two persistent launches from two threads on two plans: both pass the gate, both results right
        {
            void* dr2 = nullptr;
            CHECK(sdrk_dev_alloc(0, (size_t)nf * n * 4, &dr2) == SDRK_OK);
            sdrk_plan* p2 = nullptr;
            CHECK(sdrk_plan_create(0, n, 1024, SDRK_WINDOW_RECT, nullptr, 1e-12f, 1, &p2) == SDRK_OK);
            std::thread t([&] { CHECK(sdrk_exec_device(p2, dx, nf, 1, static_cast<float*>(dr2), nullptr) == SDRK_OK && sdrk_plan_sync(p2) == SDRK_OK); });
            CHECK(sdrk_exec_device(p, dx, nf, 1, static_cast<float*>(dr), nullptr) == SDRK_OK && sdrk_plan_sync(p) == SDRK_OK);
            t.join();
            std::vector<float> rows2((size_t)nf * n);
            CHECK(sdrk_memcpy_d2h(0, rows2.data(), dr2, rows2.size() * 4) == SDRK_OK);
            check_rows(x.data(), 1, rows2.data(), nf, n);
            CHECK(sdrk_dev_free(0, dr2) == SDRK_OK && sdrk_plan_destroy(p2) == SDRK_OK);
        }
        // a failed hand-over: the call fails, the plan falls back for good, the same call then succeeds through the two launches
        sdrk::g_fake_fused_fail = 1;
        const int st = sdrk_exec_device(p, dx, nf, 1, static_cast<float*>(dr), nullptr);
        CHECK(st == SDRK_OK && sdrk_plan_sync(p) == SDRK_ERR_HIP);
        CHECK(sdrk_plan_fused_status(p, &launches, &fell) == SDRK_OK && launches == 3 && fell == 1);
        const int seen = sdrk::g_fake_fused_launches.load();
        CHECK(sdrk_exec_device(p, dx, nf, 1, static_cast<float*>(dr), nullptr) == SDRK_OK && sdrk_plan_sync(p) == SDRK_OK);
        CHECK(sdrk::g_fake_fused_launches.load() == seen && seen - before == 4);
        CHECK(sdrk_plan_fused_status(p, &launches, &fell) == SDRK_OK && launches == 3 && fell == 1);
        CHECK(sdrk_memcpy_d2h(0, rows.data(), dr, rows.size() * 4) == SDRK_OK);
        check_rows(x.data(), 1, rows.data(), nf, n);
    }
    CHECK(sdrk_plan_fused_status(nullptr, &launches, &fell) == SDRK_ERR_INVALID);
    CHECK(sdrk_plan_destroy(p) == SDRK_OK && sdrk_plan_destroy(forced_tiled) == SDRK_OK);
    CHECK(sdrk_dev_free(0, dx) == SDRK_OK && sdrk_dev_free(0, dr) == SDRK_OK);
    fakehip::cus() = 8;
}

int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 3;
    const int rounds = argc > 2 ? atoi(argv[2]) : 1;
    printf("sdrk %d (host side on the stand-in runtime), %d device(s), %d helper threads\n", sdrk_version(), sdrk_device_count(), sdrk_host_threads());
    std::vector<std::thread> ts;
    for (int t = 0; t < threads; ++t) ts.emplace_back(exec_host_paths, t, rounds);
    ts.emplace_back(features_paths, 0);
    ts.emplace_back(waterfall_paths, 0);
    ts.emplace_back(waterfall_paths, 1);
    ts.emplace_back([] {                                                // pinned-range table churn beside everything else
        for (int i = 0; i < 200; ++i) {
            void* h = nullptr;
            if (sdrk_host_alloc(4096 + 64 * (size_t)i, &h) == SDRK_OK) {
                CHECK(sdrk_host_is_pinned(h, 4096) == 1 && sdrk_host_is_pinned(static_cast<char*>(h) + 4096 + 64 * (size_t)i, 1) == 0);
                CHECK(sdrk_host_free(h) == SDRK_OK);
            }
        }
    });
    for (auto& t : ts) t.join();
    waterfall_companion_rows();
    placement_and_misc();
    fused_auto_policy();
    printf("bad=%d\n", g_bad.load());
    return g_bad.load() ? 1 : 0;
}
