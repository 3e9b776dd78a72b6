// sdrk_api.hip — host side of the C ABI declared in include/sdrk.h: plans,
// device staging, the pinned host pipeline, the waterfall ring, error reporting.
// All compute is in the gfx950 kernels (fft4096.hip, fft_lds.hip, fft_tiled2.hip,
// fft_small.hip, bluestein.hip, fft_fused64k.hip, row_features.hip, aux_kernels.hip);
// there is no host fallback anywhere in this file.
#include "../../include/sdrk.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "host_pool.h"
#include "kernels.h"

namespace {

thread_local std::string g_last_error = "";

int fail(int status, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int fail(int status, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess)                                                             \
            return fail(e__ == hipErrorOutOfMemory ? SDRK_ERR_NOMEM : SDRK_ERR_HIP,        \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__,  \
                        __LINE__);                                                         \
    } while (0)

int check_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(SDRK_ERR_NO_DEVICE, "no HIP device available (%s)",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    }
    if (device < 0 || device >= n)
        return fail(SDRK_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    return SDRK_OK;
}

bool is_pow2(long long v) { return v > 0 && (v & (v - 1)) == 0; }

// Pinned host ranges the library knows about (sdrk_host_alloc / sdrk_host_register): start -> (bytes, owned)
struct PinnedRanges {
    std::mutex m;
    std::map<uintptr_t, std::pair<size_t, bool>> r;
    bool covers(const void* p, size_t bytes) {
        if (!p || bytes == 0) return false;
        std::lock_guard<std::mutex> g(m);
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        auto it = r.upper_bound(a);
        if (it == r.begin()) return false;
        --it;
        return a >= it->first && a + bytes <= it->first + it->second.first;
    }
};
PinnedRanges& pinned_ranges() {
    static PinnedRanges pr;
    return pr;
}

// exp(-2 pi i m / n) in double, rounded once to float32.
float2 twiddle(double m, double n) {
    const double a = -2.0 * M_PI * m / n;
    return make_float2((float)std::cos(a), (float)std::sin(a));
}

}  // namespace

namespace {
constexpr int HOST_SLOTS = 3;
struct HostSlot {
    void *h_in = nullptr, *h_out = nullptr;   // pinned
    void *d_in = nullptr, *d_out = nullptr;
    size_t in_cap = 0, out_cap = 0;
    hipEvent_t ev_in = nullptr, ev_k = nullptr, ev_done = nullptr;
    // the chunk in flight in this slot (busy == true): where its rows go once ev_done has fired
    // (user_out == nullptr: the rows were DMA'd straight into the caller's pinned array)
    bool busy = false;
    void* user_out = nullptr;
    size_t out_bytes = 0;
};
}  // namespace

struct sdrk_plan {
    int device = 0;
    int nfft = 0;
    size_t max_batch = 0;
    float eps = 1e-12f;
    int shift = 1;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float* d_window = nullptr;     // nfft floats, or nullptr for rectangular
    float2* d_twiddle = nullptr;   // W_min(nfft,4096)^m
    float2* d_scratch = nullptr;   // large plans
    size_t scratch_frames = 0;
    void* d_in = nullptr;          // whole-stream staging (Welch PSD, waterfall append from host IQ); only grows
    size_t in_cap = 0;
    void* d_out = nullptr;
    size_t out_cap = 0;
    void* d_feat = nullptr;          // per-row feature results of sdrk_frame_features_host (only grows)
    size_t feat_cap = 0;
    float2* d_tw_2p = nullptr;       // two-pass tiled plans (fft_tiled2.hip)
    bool tiled2 = false;
    // overlapped form of the two-pass plans: row pass of chunk i on stream2 beside the col pass of chunk i + 1
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_col[2] = {nullptr, nullptr}, ev_row[2] = {nullptr, nullptr}, ev_fork = nullptr;
    int col_cus = 0, row_cus = 0;    // 0: serial form
    // sdrk_exec_host pipeline: HOST_SLOTS chunks in flight, each with pinned host and device staging
    HostSlot slot[HOST_SLOTS];
    float staging_probe_ms[HOST_SLOTS * 3] = {};   // SDRK_PLAN_TUNE_STAGING: the transform over each candidate pairing
    int staging_probe_n = 0;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    // non-power-of-two lengths (bluestein.hip): inner power-of-two plan of size blu_m
    sdrk_plan* blu_inner = nullptr;
    int blu_m = 0;
    float2* d_blu_chirp = nullptr;   // c[n] = exp(+i pi n^2 / N), n < N
    float2* d_blu_bspec = nullptr;   // FFT_M(b)
    float2* d_blu_a = nullptr;       // work buffers: blu_frames * M complex64 each
    float2* d_blu_b = nullptr;
    size_t blu_frames = 0;
    // small-call fast path of sdrk_exec_host: pinned, device-mapped staging the kernel reads and
    // writes directly over PCIe (no DMA-engine copies for a 32 KiB frame)
    void* h_small_in = nullptr;
    void* h_small_out = nullptr;
    uint32_t* h_small_flag = nullptr;   // completion word the stream writes behind a small call
    void* d_small_flag = nullptr;
    uint32_t small_seq = 0;
    // N = 65536 fused path (fft_fused64k.hip).  fused64k: every launch (SDRK_PLAN_FUSED64K); fused_auto: launches of at least
    // FUSED_AUTO_MIN_FRAMES frames (the default for nfft = 65536), until one reports a failed hand-over (fused_broken, latched).
    bool fused64k = false, fused_auto = false, fused_broken = false;
    void* d_fused_ring = nullptr;
    unsigned* d_fused_ctrl = nullptr;
    unsigned* h_fused_err = nullptr;   // pinned mailbox: error word of the last launches
    unsigned fused_launches = 0, fused_pending = 0;
};
constexpr size_t SMALL_IN_BYTES = 256 << 10;   // calls up to this much input take the zero-copy path
constexpr size_t HOST_CHUNK_BYTES = 16 << 20;  // target input bytes per pipelined chunk of sdrk_exec_host
constexpr size_t ZERO_COPY_MAX_BYTES = 32 << 20;  // calls up to this much input skip the DMA engines (see exec_host_common)
constexpr unsigned FUSED_MAILBOX = 64;   // entries of 8 words: error flag + debug record
// Below this many frames a launch of an auto plan takes the two tiled launches: the persistent launch costs about 30 us before
// its first row (control-block memset, role formation, the ramp of a set's pipeline, the mailbox copy) against 11-18 us, and
// the two forms cross between 384 and 512 frames, packed or half-overlapped (profiles/r06/fused64k_crossover.log).
constexpr size_t FUSED_AUTO_MIN_FRAMES = 512;

struct sdrk_waterfall {
    int device = 0;
    int nfft = 0;
    int maxlen = 0;
    float* d_ring = nullptr;  // maxlen * nfft float32
    size_t head = 0;          // slot the next row is written to
    size_t count = 0;         // valid rows (<= maxlen)
    hipStream_t stream = nullptr;
    // decimated read-out staging (only grows).  Slot 0 also serves the one-call form (sdrk_waterfall_read_decimated).
    void* d_dec[2] = {nullptr, nullptr};
    size_t dec_cap[2] = {0, 0};
    // two-phase decimated read-out, up to TWO in flight (a channel that enqueues batch i + 1 before it collects batch i - 1 keeps
    // the transform stream fed): the reduction AND the copy run on a second stream behind the transform that produced the rows
    // (ev_dec = "rows written", recorded on `stream`), so the transform stream goes straight on with the next batch;
    // ev_dec_done[k] = "reduction k finished with the ring" (recorded on s_copy): a later write into ring slots [dec_start[k],
    // dec_start[k] + dec_rows[k]) waits for it (wf_before_write) — in a running channel those are the newest rows and the next
    // batch lands elsewhere, so nothing waits; ev_copy_done[k] = its rows are in the caller's array.
    hipStream_t s_copy = nullptr;
    hipEvent_t ev_dec = nullptr, ev_dec_done[2] = {nullptr, nullptr}, ev_copy_done[2] = {nullptr, nullptr};
    int reads_in_flight = 0, oldest_read = 0;
    bool dec_guard[2] = {false, false};
    size_t dec_start[2] = {0, 0}, dec_rows[2] = {0, 0};
    // frame lengths whose transform can write them (sdrk::fft_tiled2_has_mip): every ring row max-hold-decimated by 16,
    // maxlen * nfft / 16 float32, written by the row pass beside the row itself; mip_ok[slot] = that slot's row came from
    // sdrk_waterfall_append_iq* (rows appended as finished rows have none)
    float* d_mip_ring = nullptr;
    std::vector<unsigned char> mip_ok;
};

namespace {

// Optional roctx ranges around every transform (SDRK_ROCTX=1; SURVEY.md §5 "tracing"): nfft, frames, stride, epilogue, so
// that a rocprofv3 --marker-trace names the calls the kernels belong to.  The library is dlopen'ed on first use — nothing is
// linked, and without the variable the cost is one load of a static.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* env = getenv("SDRK_ROCTX");
        if (!env || env[0] != '1') return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
struct RoctxRange {
    bool on = false;
    RoctxRange(const sdrk_plan* p, size_t n_frames, size_t stride, int epilogue);
    ~RoctxRange();
};
Roctx& roctx() {
    static Roctx r;
    return r;
}

// A placement probe's launch lambda maps a failed plan_launch to hipErrorUnknown; plan_launch has then already recorded the
// specific status and message on this thread — keep them instead of overwriting them with "unknown error".
thread_local int g_probe_status = SDRK_OK;
hipError_t probe_launch_result(int st) {
    if (st == SDRK_OK) return hipSuccess;
    g_probe_status = st;
    return hipErrorUnknown;
}
int probe_fail(hipError_t e, const char* what) {
    if (e == hipErrorUnknown && g_probe_status != SDRK_OK) {
        const int st = g_probe_status;
        g_probe_status = SDRK_OK;
        return st;                                     // sdrk_last_error() still holds plan_launch's own text
    }
    return fail(SDRK_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

// Does a launch of n_frames frames of this plan take the persistent N = 65536 kernel?
bool takes_fused(const sdrk_plan* p, size_t n_frames) {
    return p->fused64k || (p->fused_auto && !p->fused_broken && n_frames >= FUSED_AUTO_MIN_FRAMES);
}

// One fused N = 65536 launch at a time per device (see plan_launch_impl).  The events live for the life of the process.
struct FusedGate {
    std::mutex mu;
    hipEvent_t ev = nullptr;
    hipStream_t last_stream = nullptr;
    bool recorded = false;
};
FusedGate& fused_gate(int device) {
    static FusedGate gates[64];
    FusedGate& g = gates[device >= 0 && device < 64 ? device : 0];
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.ev && hipEventCreateWithFlags(&g.ev, hipEventDisableTiming) != hipSuccess) g.ev = nullptr;
    return g;
}

// The chirp-z path's multiplies riding on an inner (power-of-two, two-pass) transform's row-pass stores: epilogue EPI_BLU_* with
// this table, row length, and the OUTER plan's eps / shift (kernels.h).
struct EpiArgs {
    const float2* tab = nullptr;
    int n_out = 0;
    float eps = 0.0f;
    int shift = 0;
    size_t in_valid = 0;   // samples that exist per input frame (0 = all): LaunchArgs::in_valid
};

int plan_launch_impl(sdrk_plan* p, const void* d_iq, size_t n_frames, size_t frame_stride, void* d_out,
                     int epilogue, hipStream_t stream, float* d_mip, bool* mip_written, const EpiArgs* epi);

// d_mip / mip_written: see LaunchArgs (kernels.h) — *mip_written tells whether the launch wrote the by-16 companion rows.
int plan_launch(sdrk_plan* p, const void* d_iq, size_t n_frames, size_t frame_stride, void* d_out,
                int epilogue, hipStream_t stream, float* d_mip = nullptr, bool* mip_written = nullptr, const EpiArgs* epi = nullptr) {
    RoctxRange range(p, n_frames, frame_stride, epilogue);
    if (mip_written) *mip_written = false;
    return plan_launch_impl(p, d_iq, n_frames, frame_stride, d_out, epilogue, stream, d_mip, mip_written, epi);
}

int plan_launch_impl(sdrk_plan* p, const void* d_iq, size_t n_frames, size_t frame_stride, void* d_out,
                     int epilogue, hipStream_t stream, float* d_mip, bool* mip_written, const EpiArgs* epi) {
    sdrk::LaunchArgs a;
    a.d_mip = d_mip;
    a.mip_written = mip_written;
    a.d_iq = d_iq;
    a.frame_stride = frame_stride;
    a.d_out = d_out;
    a.n_frames = n_frames;
    a.nfft = p->nfft;
    a.d_window = p->d_window;
    a.d_twiddle = p->d_twiddle;
    a.eps = p->eps;
    a.shift = p->shift;
    a.epilogue = epilogue;
    a.stream = stream;
    a.num_cus = p->num_cus;
    a.d_scratch = p->d_scratch;
    a.scratch_frames = p->scratch_frames;
    a.d_twiddle_2p = p->d_tw_2p;
    if (epi) {
        if (!p->tiled2 || p->stream2 || epilogue < sdrk::EPI_BLU_MUL)
            return fail(SDRK_ERR_INVALID, "chirp-z epilogues need a serial two-pass inner plan");
        a.d_epi_tab = epi->tab;
        a.epi_n_out = epi->n_out;
        a.eps = epi->eps;
        a.shift = epi->shift;
        a.in_valid = epi->in_valid;
    } else if (epilogue >= sdrk::EPI_BLU_MUL) {
        return fail(SDRK_ERR_INVALID, "epilogue %d needs its table", epilogue);
    }
    if (p->col_cus > 0 && p->stream2) {
        a.stream2 = p->stream2;
        a.ev_fork = p->ev_fork;
        for (int h = 0; h < 2; ++h) { a.ev_col[h] = p->ev_col[h]; a.ev_row[h] = p->ev_row[h]; }
        a.col_cus = p->col_cus;
        a.row_cus = p->row_cus;
    }
    hipError_t e = hipSuccess;
    if (p->blu_inner) {
        const int N = p->nfft, M = p->blu_m;
        const size_t out_elem = epilogue == sdrk::EPI_LOGPSD ? sizeof(float) : sizeof(float2);
        for (size_t f0 = 0; f0 < n_frames; f0 += p->blu_frames) {
            const size_t nf = n_frames - f0 < p->blu_frames ? n_frames - f0 : p->blu_frames;
            if (sdrk::blu_fused_supports(M)) {   // one kernel, one pass over HBM, instead of five
                e = sdrk::launch_blu_fused(static_cast<const float2*>(d_iq) + f0 * frame_stride, frame_stride, nf, N, M,
                                           p->d_window, p->d_blu_chirp, p->d_blu_bspec, p->blu_inner->d_twiddle,
                                           p->eps, p->shift, epilogue,
                                           static_cast<char*>(d_out) + f0 * (size_t)N * out_elem, p->num_cus, stream);
                if (e != hipSuccess) break;
                continue;
            }
            const bool fused_multiplies = p->blu_inner->tiled2 && !p->blu_inner->stream2;
            // (the pre-multiply writes only the N values that exist; the first col pass reads the padding as zeros by bounds check)
            e = sdrk::launch_blu_pre(static_cast<const float2*>(d_iq) + f0 * frame_stride, frame_stride, nf, N, M,
                                     p->d_window, p->d_blu_chirp, p->d_blu_a, p->num_cus, stream, fused_multiplies);
            if (e != hipSuccess) break;
            if (fused_multiplies) {
                // five launches instead of seven (round 6): the filter multiply (+ conjugation) rides on the stores of the first
                // inner transform's row pass, the post-multiply / crop to N / fftshift / log on the second one's, which writes the
                // caller's rows directly (fft_tiled2.hip, EPI_BLU_*): 16 M + 8 M + 12 N bytes per frame fewer through the fabric
                EpiArgs mul, post;
                mul.tab = p->d_blu_bspec;
                mul.in_valid = (size_t)N;
                post.tab = p->d_blu_chirp;
                post.n_out = N;
                post.eps = p->eps;
                post.shift = p->shift;
                int st = plan_launch(p->blu_inner, p->d_blu_a, nf, (size_t)M, p->d_blu_b, sdrk::EPI_BLU_MUL, stream, nullptr, nullptr, &mul);
                if (st != SDRK_OK) return st;
                st = plan_launch(p->blu_inner, p->d_blu_b, nf, (size_t)M, static_cast<char*>(d_out) + f0 * (size_t)N * out_elem,
                                 epilogue == sdrk::EPI_LOGPSD ? sdrk::EPI_BLU_POST_LOG : sdrk::EPI_BLU_POST_C64, stream, nullptr, nullptr, &post);
                if (st != SDRK_OK) return st;
                continue;
            }
            int st = plan_launch(p->blu_inner, p->d_blu_a, nf, (size_t)M, p->d_blu_b, sdrk::EPI_COMPLEX, stream);
            if (st != SDRK_OK) return st;
            e = sdrk::launch_blu_mul(p->d_blu_b, p->d_blu_bspec, nf, M, p->d_blu_a, p->num_cus, stream);
            if (e != hipSuccess) break;
            st = plan_launch(p->blu_inner, p->d_blu_a, nf, (size_t)M, p->d_blu_b, sdrk::EPI_COMPLEX, stream);
            if (st != SDRK_OK) return st;
            e = sdrk::launch_blu_post(p->d_blu_b, p->d_blu_chirp, nf, N, M, p->eps, p->shift, epilogue,
                                      static_cast<char*>(d_out) + f0 * (size_t)N * out_elem, p->num_cus, stream);
            if (e != hipSuccess) break;
        }
    } else if (p->nfft == 4096)
        e = sdrk::launch_fft4096(a);
    else if (sdrk::fft_lds_supports(p->nfft))
        e = sdrk::launch_fft_lds(a);
    else if (p->nfft < 4096)
        e = sdrk::launch_fft_small(a);
    else if (takes_fused(p, n_frames) && !d_mip && epilogue <= sdrk::EPI_COMPLEX) {
        // The persistent grid needs every one of its workgroups resident at the same time; two such grids on two streams could
        // each hold part of the device and wait for the rest.  One at a time per device: each launch waits for the one before.
        FusedGate& gate = fused_gate(p->device);
        std::lock_guard<std::mutex> lk(gate.mu);
        if (gate.ev && gate.last_stream != stream && gate.recorded) e = hipStreamWaitEvent(stream, gate.ev, 0);
        if (e == hipSuccess) e = sdrk::launch_fused64k(a, p->d_fused_ring, p->d_fused_ctrl);
        if (e == hipSuccess)   // error word, timeout record and the number of sets formed -> pinned mailbox
            e = hipMemcpyAsync(p->h_fused_err + 16 * (p->fused_launches++ % FUSED_MAILBOX), p->d_fused_ctrl,
                               16 * sizeof(unsigned), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess && gate.ev) {
            e = hipEventRecord(gate.ev, stream);
            gate.last_stream = stream;
            gate.recorded = true;
        }
        ++p->fused_pending;
    } else if (p->tiled2)
        e = sdrk::launch_fft_tiled2(a);
    else
        return fail(SDRK_ERR_UNSUPPORTED, "no kernel for nfft=%d", p->nfft);
    if (e != hipSuccess) return fail(SDRK_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return SDRK_OK;
}

RoctxRange::RoctxRange(const sdrk_plan* p, size_t n_frames, size_t stride, int epilogue) {
    Roctx& r = roctx();
    if (!r.push) return;
    char label[160];
    snprintf(label, sizeof label, "sdrk.plan_launch nfft=%d frames=%zu stride=%zu %s dev=%d", p->nfft, n_frames, stride,
             epilogue == sdrk::EPI_LOGPSD ? "logpsd" : (epilogue == sdrk::EPI_COMPLEX ? "complex" : "chirp-z stage"), p->device);
    r.push(label);
    on = true;
}
RoctxRange::~RoctxRange() {
    if (on) roctx().pop();
}

// After a stream sync: did any fused N=65536 launch report an internal wait timeout?
int fused_check(sdrk_plan* p) {
    if (!p->h_fused_err || p->fused_pending == 0) return SDRK_OK;
    // mailbox entry: [0] sets formed + 1 (0 = unused entry), [1] error flag, [2..5] record of the first timeout
    const unsigned want = sdrk::fused64k_sets(p->num_cus);
    unsigned bad = 0, rec[16] = {0};
    const unsigned pending = p->fused_pending < FUSED_MAILBOX ? p->fused_pending : FUSED_MAILBOX;
    for (unsigned k = 0; k < pending; ++k) {            // the launches since the last check (older ones were overwritten)
        const unsigned* r = p->h_fused_err + 16 * ((p->fused_launches - 1 - k) % FUSED_MAILBOX);
        const unsigned code = r[1] ? r[1] : (r[0] != want ? 9u : 0u);
        if (code) { memcpy(rec, r, sizeof rec); bad = code; }
    }
    p->fused_pending = 0;
    if (bad && p->fused_auto) p->fused_broken = true;   // an auto plan takes the two tiled launches from here on
    if (bad)
        return fail(SDRK_ERR_HIP, "fused N=65536 kernel reported an internal synchronisation error (code %u; %u of %u sets formed; "
                    "word %u held %u, wanted %u, site %u)", bad, rec[0], want, rec[2], rec[3], rec[4], rec[5]);
    return SDRK_OK;
}

int check_exec_args(const sdrk_plan* p, const void* in, size_t n_frames, size_t frame_stride,
                    const void* out) {
    if (!p) return fail(SDRK_ERR_INVALID, "plan is NULL");
    if (n_frames == 0) return SDRK_OK;
    if (!in || !out) return fail(SDRK_ERR_INVALID, "input or output pointer is NULL");
    if (frame_stride == 0 && n_frames > 1)
        return fail(SDRK_ERR_INVALID, "frame_stride must be >= 1 for more than one frame");
    return SDRK_OK;
}

// Device staging that only ever grows (no free + malloc per call once the largest size has been seen).
int grow(int device, void** buf, size_t* cap, size_t need) {
    if (need <= *cap) return SDRK_OK;
    if (*buf) {
        HIP_TRY(hipFree(*buf));
        *buf = nullptr;
        *cap = 0;
    }
    const size_t want = need + need / 4;   // head-room: a slightly larger next call does not reallocate
    if (hipMalloc(buf, want) != hipSuccess) {
        (void)hipGetLastError();
        HIP_TRY(hipMalloc(buf, need));
        *cap = need;
    } else {
        *cap = want;
    }
    (void)device;
    return SDRK_OK;
}

int slot_reserve(sdrk_plan* p, HostSlot& s, size_t in_bytes, size_t out_bytes) {
    if (!s.ev_in) {
        HIP_TRY(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.ev_k, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
    }
    if (in_bytes > s.in_cap) {
        if (s.h_in) HIP_TRY(hipHostFree(s.h_in));
        if (s.d_in) HIP_TRY(hipFree(s.d_in));
        s.h_in = s.d_in = nullptr;
        s.in_cap = 0;
        HIP_TRY(hipHostMalloc(&s.h_in, in_bytes, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&s.d_in, in_bytes));
        s.in_cap = in_bytes;
    }
    if (out_bytes > s.out_cap) {
        if (s.h_out) HIP_TRY(hipHostFree(s.h_out));
        if (s.d_out) HIP_TRY(hipFree(s.d_out));
        s.h_out = s.d_out = nullptr;
        s.out_cap = 0;
        HIP_TRY(hipHostMalloc(&s.h_out, out_bytes, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&s.d_out, out_bytes));
        s.out_cap = out_bytes;
    }
    (void)p;
    return SDRK_OK;
}

struct HostTrace {   // SDRK_HOST_TRACE=1: where a pipelined sdrk_exec_host call spends its wall time (stderr)
    bool on = false;
    double t_in = 0, t_wait = 0, t_out = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};

// Wait for the chunk in flight in slot `s` and hand its rows to the caller's array.
int slot_retire(HostSlot& s, HostTrace& tr) {
    if (!s.busy) return SDRK_OK;
    s.busy = false;
    const double t0 = tr.on ? HostTrace::now() : 0;
    HIP_TRY(hipEventSynchronize(s.ev_done));
    const double t1 = tr.on ? HostTrace::now() : 0;
    if (s.user_out) sdrk::CopyPool::get().copy(s.user_out, s.h_out, s.out_bytes);
    if (tr.on) { tr.t_wait += t1 - t0; tr.t_out += HostTrace::now() - t1; }
    return SDRK_OK;
}

void slots_abandon(sdrk_plan* p) {   // error path: nothing may still be writing into the staging buffers
    (void)hipStreamSynchronize(p->s_h2d);
    (void)hipStreamSynchronize(p->stream);
    (void)hipStreamSynchronize(p->s_d2h);
    for (auto& s : p->slot) s.busy = false;
}

// ---- placement probes: warm first, then compare ---------------------------------------------------------------------
// An idle MI355X runs its shader clock near 1.0-1.4 GHz and needs tens of milliseconds of sustained load to reach the
// 1.85-2.0 GHz it holds afterwards (round 5, tools/cfg_steady.py: thirty back-to-back N = 2^20 transforms from idle take
// 1.53, 1.46, 1.46, 1.45, 1.42 ... 1.35 ms).  A probe that times candidate after candidate from a cold start therefore
// measures that ramp: every later candidate looks faster (round 4's sdrk_plan_tune_scratch records on config 5 were
// monotone in six runs of six, and "chose" the last candidate every time).  So every placement probe here (a) warms up BY
// TIME with the very launch it is going to time, and (b) times candidate 0 a second time after the last candidate: what a
// candidate gains is its time against that re-timed figure, and a gain under one per cent keeps what is already there.
struct PlacementReport {
    float warm_ms = 0.0f;            // wall time of the warm-up launches
    int warm_launches = 0;
    float first_ms = 0.0f;           // candidate 0 as first timed (after the warm-up)
    float retimed_first_ms = 0.0f;   // candidate 0 timed again after the last candidate
    float chosen_ms = 0.0f;          // the kept candidate's time
    int candidates = 0, chosen = 0;
};
thread_local PlacementReport g_placement;
constexpr double PLACEMENT_WARM_MS = 60.0;

template <typename Launch>
hipError_t placement_warm_up(hipStream_t s, Launch&& launch, PlacementReport& rep, int max_launches = 4000) {
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipSuccess;
    int n = 0;
    double ms = 0.0;
    while (e == hipSuccess && n < max_launches) {
        e = launch();
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        ++n;
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms >= PLACEMENT_WARM_MS && n >= 2) break;
    }
    rep.warm_ms = (float)ms;
    rep.warm_launches = n;
    return e;
}

// one untimed launch, then the median of three isolated ones (event, launch, event, wait)
template <typename Launch>
hipError_t placement_time(hipStream_t s, hipEvent_t e0, hipEvent_t e1, Launch&& launch, float* median_ms) {
    float t[4] = {0, 0, 0, 0};
    hipError_t e = hipSuccess;
    for (int r = 0; r < 4 && e == hipSuccess; ++r) {
        e = hipEventRecord(e0, s);
        if (e == hipSuccess) e = launch();
        if (e == hipSuccess) e = hipEventRecord(e1, s);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&t[r], e0, e1);
    }
    std::sort(t + 1, t + 4);
    *median_ms = t[2];
    return e;
}

// SDRK_PLAN_TUNE_STAGING: the numpy boundary's device staging (HOST_SLOTS chunk pairs of ~16 MiB in / 8 MiB out) allocated
// at plan creation, each slot's row buffer the fastest of three candidates under the plan's own transform over the
// chunk — the pairing effect of DESIGN.md §4.1 applied to the library's own buffers.  (Measured in round 4: the probe
// times of the candidates agree to the microsecond and B = 32768 does not move — a 24 MiB pair lives in the L2 /
// Infinity Cache, where placement levels do not exist, and the kernel is 1.5 % of a PCIe-bound call.  The flag stays
// for plans whose chunks are made larger.)
int tune_staging(sdrk_plan* p) {
    const size_t nfft = (size_t)p->nfft;
    if (p->blu_inner || p->max_batch * nfft * sizeof(float2) <= 2 * HOST_CHUNK_BYTES) return SDRK_OK;   // small plans: nothing staged in chunks
    size_t per = HOST_CHUNK_BYTES / (nfft * sizeof(float2));
    if (per < 1) per = 1;
    const size_t in_b = per * nfft * sizeof(float2), out_b = per * nfft * sizeof(float);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    if (hipError_t ee = hipEventCreate(&e1); ee != hipSuccess) {
        (void)hipEventDestroy(e0);
        return fail(SDRK_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(ee));
    }
    int st = SDRK_OK;
    PlacementReport rep;
    for (int i = 0; i < HOST_SLOTS && st == SDRK_OK; ++i) {
        HostSlot& s = p->slot[i];
        st = slot_reserve(p, s, in_b, 0);                        // events, pinned h_in, d_in
        if (st != SDRK_OK) break;
        if (hipHostMalloc(&s.h_out, out_b, hipHostMallocDefault) != hipSuccess) { st = fail(SDRK_ERR_NOMEM, "pinned staging"); break; }
        void* cand[3] = {nullptr, nullptr, nullptr};
        int best = 0;
        for (int c = 0; c < 3 && st == SDRK_OK; ++c) {           // earlier candidates stay allocated: each lands elsewhere
            if (hipMalloc(&cand[c], out_b) != hipSuccess) { st = fail(SDRK_ERR_NOMEM, "device staging"); break; }
            auto launch = [&]() -> hipError_t {
                return probe_launch_result(plan_launch(p, s.d_in, per, nfft, cand[c], sdrk::EPI_LOGPSD, p->stream));
            };
            hipError_t e = hipSuccess;
            if (i == 0 && c == 0) e = placement_warm_up(p->stream, launch, rep);   // (see "placement probes" above)
            float med = 0.0f;
            if (e == hipSuccess) e = placement_time(p->stream, e0, e1, launch, &med);
            if (e != hipSuccess) { st = probe_fail(e, "staging probe failed"); break; }
            p->staging_probe_ms[i * 3 + c] = med;
            if (med < p->staging_probe_ms[i * 3 + best]) best = c;
        }
        for (int c = 0; c < 3; ++c) {
            if (c == best && st == SDRK_OK) { s.d_out = cand[c]; s.out_cap = out_b; }
            else if (cand[c]) (void)hipFree(cand[c]);
        }
        if (st == SDRK_OK) p->staging_probe_n = (i + 1) * 3;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return st;
}

// The numpy boundary.  Small calls (the live app's one 4096-sample buffer, streamer.py:114-121): the
// kernel reads and writes pinned mapped host memory, no DMA copies.  Everything else: the frames go
// through in chunks of ~16 MiB, HOST_SLOTS of them in flight — helper threads copy the caller's pageable
// memory into a pinned slot, then H2D (copy stream) -> transform (plan stream) -> D2H (copy stream) run
// asynchronously, chained by events, while the host stages the next chunk and drains finished ones.
// H2D of chunk c+1 overlaps D2H of chunk c (PCIe is full duplex) and both overlap the staging memcpys.
int exec_host_common(sdrk_plan* p, const void* iq, size_t n_frames, size_t frame_stride, void* out,
                     int epilogue) {
    int st = check_exec_args(p, iq, n_frames, frame_stride, out);
    if (st != SDRK_OK || n_frames == 0) return st;
    if (n_frames > p->max_batch)
        return fail(SDRK_ERR_INVALID, "n_frames %zu exceeds the plan's max_batch %zu", n_frames,
                    p->max_batch);
    HIP_TRY(hipSetDevice(p->device));
    const size_t nfft = (size_t)p->nfft;
    const size_t in_samples = (n_frames - 1) * frame_stride + nfft;
    const size_t in_bytes = in_samples * sizeof(float2);
    const size_t out_elem = epilogue == sdrk::EPI_LOGPSD ? sizeof(float) : sizeof(float2);
    const size_t out_bytes = n_frames * nfft * out_elem;
    if (in_bytes <= SMALL_IN_BYTES && out_bytes <= SMALL_IN_BYTES && p->nfft <= 4096 && !p->blu_inner) {
        if (!p->h_small_in) {
            HIP_TRY(hipHostMalloc(&p->h_small_in, SMALL_IN_BYTES, hipHostMallocMapped));
            HIP_TRY(hipHostMalloc(&p->h_small_out, SMALL_IN_BYTES, hipHostMallocMapped));
            void* f = nullptr;
            if (hipHostMalloc(&f, 64, hipHostMallocMapped) == hipSuccess) {
                memset(f, 0, 64);
                p->h_small_flag = static_cast<uint32_t*>(f);
                if (hipHostGetDevicePointer(&p->d_small_flag, f, 0) != hipSuccess) {
                    (void)hipHostFree(f);
                    p->h_small_flag = nullptr;
                }
            }
            (void)hipGetLastError();
        }
        void *d_si = nullptr, *d_so = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&d_si, p->h_small_in, 0));
        HIP_TRY(hipHostGetDevicePointer(&d_so, p->h_small_out, 0));
        memcpy(p->h_small_in, iq, in_bytes);
        st = plan_launch(p, d_si, n_frames, frame_stride, d_so, epilogue, p->stream);
        if (st != SDRK_OK) return st;
        // Completion: the stream writes a sequence number into mapped host memory behind the kernel and the caller
        // polls it — for a 10 us job the wake-up path of hipStreamSynchronize costs as much as the job.  Falls back
        // to the synchronize after ~200 us of polling (or if the stream memory operation is not available).
        static const bool poll_ok = getenv("SDRK_SMALL_NOPOLL") == nullptr;
        bool done = false;
        if (poll_ok && p->h_small_flag) {
            const uint32_t seq = ++p->small_seq;
            if (hipStreamWriteValue32(p->stream, p->d_small_flag, seq, 0) == hipSuccess) {
                const uint32_t* flag = p->h_small_flag;
                for (int spins = 0; spins < 200000; ++spins) {
                    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) { done = true; break; }   // (a plain mov on x86: the acquire only binds the compiler)
                    __builtin_ia32_pause();
                }
            } else {
                (void)hipGetLastError();
            }
        }
        if (!done) HIP_TRY(hipStreamSynchronize(p->stream));
        memcpy(out, p->h_small_out, out_bytes);
        return SDRK_OK;
    }
    if (!p->s_h2d) {
        HIP_TRY(hipStreamCreateWithFlags(&p->s_h2d, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&p->s_d2h, hipStreamNonBlocking));
    }
    // frames per chunk: ~HOST_CHUNK_BYTES of input, but at least 4 chunks per call when the call is big
    // enough for the overlap to matter, and never less than one frame
    const size_t stride_bytes = (frame_stride ? frame_stride : 1) * sizeof(float2);
    size_t target = HOST_CHUNK_BYTES;
    if (in_bytes / 4 < target) target = in_bytes / 4 > ((size_t)1 << 20) ? in_bytes / 4 : ((size_t)1 << 20);
    size_t per = target / stride_bytes;
    if (per > target / (nfft * out_elem)) per = target / (nfft * out_elem);   // heavily overlapped frames: bound the rows too
    if (per < 1) per = 1;
    if (per > n_frames) per = n_frames;
    const size_t chunk_in = ((per - 1) * frame_stride + nfft) * sizeof(float2);
    const size_t chunk_out = per * nfft * out_elem;
    sdrk::CopyPool& pool = sdrk::CopyPool::get();
    HostTrace tr;
    { const char* env = getenv("SDRK_HOST_TRACE"); tr.on = env && env[0] == '1'; }
    // Mid-size calls of packed frames on the single-pass kernels: let the transform read the pinned chunk and
    // write the pinned rows itself over PCIe (measured on MI355X, N = 4096: B = 16 48 vs 81 us, B = 256 0.34 vs
    // 0.44 ms against the DMA form; equal at 32 MiB).  Large calls, overlapped frames (the halo would cross PCIe
    // twice) and the two-pass kernels (their 128-byte column segments read host memory at half the DMA rate:
    // 26 vs 48 GB/s at N = 65536) use the copy engines.
    // Caller arrays in pinned memory (sdrk_host_alloc / sdrk_host_register) are not staged: the copy engines read
    // and write them directly.  Decided per side.
    const bool in_pinned = pinned_ranges().covers(iq, in_bytes), out_pinned = pinned_ranges().covers(out, out_bytes);
    const bool zero_copy = p->nfft <= 16384 && !p->blu_inner && frame_stride >= nfft && in_bytes <= ZERO_COPY_MAX_BYTES &&
                           !in_pinned && !out_pinned;
    if (in_pinned && out_pinned && p->nfft <= 16384 && !p->blu_inner && frame_stride >= nfft && in_bytes <= ZERO_COPY_MAX_BYTES) {
        // both arrays pinned, a call small enough that the link's latency matters more than its last 10 %: ONE launch
        // that reads the caller's frames and writes the caller's rows over PCIe — no staging, no copy engine, no chunks
        void *d_src = nullptr, *d_dst = nullptr;
        if (hipHostGetDevicePointer(&d_src, const_cast<void*>(iq), 0) == hipSuccess &&
            hipHostGetDevicePointer(&d_dst, out, 0) == hipSuccess) {
            st = plan_launch(p, d_src, n_frames, frame_stride, d_dst, epilogue, p->stream);
            if (st != SDRK_OK) return st;
            HIP_TRY(hipStreamSynchronize(p->stream));
            return fused_check(p);
        }
        (void)hipGetLastError();   // no device view of the range: take the copy-engine path below
    }
    const double t_call = tr.on ? HostTrace::now() : 0;
    size_t c = 0;
    for (size_t f0 = 0; f0 < n_frames; f0 += per, ++c) {
        HostSlot& s = p->slot[c % HOST_SLOTS];
        const size_t nf = n_frames - f0 < per ? n_frames - f0 : per;
        const size_t cin = ((nf - 1) * frame_stride + nfft) * sizeof(float2);
        const size_t cout = nf * nfft * out_elem;
        st = slot_retire(s, tr);                               // chunk c - HOST_SLOTS: rows out, slot free
        if (st == SDRK_OK) st = slot_reserve(p, s, chunk_in, chunk_out);
        if (st != SDRK_OK) { slots_abandon(p); return st; }
        const double t0 = tr.on ? HostTrace::now() : 0;
        const void* src = static_cast<const float2*>(iq) + f0 * frame_stride;
        if (!in_pinned) {
            pool.copy(s.h_in, src, cin);
            src = s.h_in;
        }
        void* user_rows = static_cast<char*>(out) + f0 * nfft * out_elem;
        if (tr.on) tr.t_in += HostTrace::now() - t0;
        hipError_t e = hipSuccess;
        if (zero_copy) {
            // the transform reads the pinned chunk and writes the pinned rows itself, over PCIe: no DMA-engine
            // copies, two API calls per chunk
            st = plan_launch(p, s.h_in, nf, frame_stride, s.h_out, epilogue, p->stream);
            if (st != SDRK_OK) { slots_abandon(p); return st; }
            e = hipEventRecord(s.ev_done, p->stream);
        } else {
            e = hipMemcpyAsync(s.d_in, src, cin, hipMemcpyHostToDevice, p->s_h2d);
            if (e == hipSuccess) e = hipEventRecord(s.ev_in, p->s_h2d);
            if (e == hipSuccess) e = hipStreamWaitEvent(p->stream, s.ev_in, 0);
            if (e == hipSuccess) {
                st = plan_launch(p, s.d_in, nf, frame_stride, s.d_out, epilogue, p->stream);
                if (st != SDRK_OK) { slots_abandon(p); return st; }
                e = hipEventRecord(s.ev_k, p->stream);
            }
            if (e == hipSuccess) e = hipStreamWaitEvent(p->s_d2h, s.ev_k, 0);
            if (e == hipSuccess) e = hipMemcpyAsync(out_pinned ? user_rows : s.h_out, s.d_out, cout, hipMemcpyDeviceToHost, p->s_d2h);
            if (e == hipSuccess) e = hipEventRecord(s.ev_done, p->s_d2h);
        }
        if (e != hipSuccess) {
            slots_abandon(p);
            return fail(SDRK_ERR_HIP, "host pipeline failed: %s", hipGetErrorString(e));
        }
        s.busy = true;
        s.user_out = (out_pinned && !zero_copy) ? nullptr : user_rows;
        s.out_bytes = cout;
    }
    for (size_t i = 0; i < HOST_SLOTS; ++i) {                  // drain in submission order
        st = slot_retire(p->slot[(c + i) % HOST_SLOTS], tr);
        if (st != SDRK_OK) { slots_abandon(p); return st; }
    }
    if (tr.on) {
        const double t = HostTrace::now() - t_call;
        fprintf(stderr, "[sdrk host] %zu chunks of %zu frames, %.1f MiB in: total %.3f ms = stage-in %.3f + wait %.3f + "
                "stage-out %.3f + other %.3f (%.1f GB/s of input, %d helper threads)\n", c, per,
                in_bytes / 1048576.0, t * 1e3, tr.t_in * 1e3, tr.t_wait * 1e3, tr.t_out * 1e3,
                (t - tr.t_in - tr.t_wait - tr.t_out) * 1e3, in_bytes / t / 1e9, pool.helpers());
    }
    return fused_check(p);
}

}  // namespace

namespace {
// `launches` timed launches (after two untimed ones) of a probe kernel on a private stream
template <typename Launch>
int timed_probe(int device, int launches, float* each_ms, const char* what, Launch&& launch) {
    int st = check_device(device);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    for (int i = -2; i < launches && e == hipSuccess; ++i) {     // two untimed warm-ups
        if (i >= 0) e = hipEventRecord(e0, s);
        if (e == hipSuccess) e = launch(prop.multiProcessorCount, s);
        if (i >= 0 && e == hipSuccess) e = hipEventRecord(e1, s);
        if (i >= 0 && e == hipSuccess) e = hipEventSynchronize(e1);
        if (i >= 0 && e == hipSuccess) e = hipEventElapsedTime(&each_ms[i], e0, e1);
    }
    (void)hipStreamSynchronize(s);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(s);
    if (e != hipSuccess) return fail(SDRK_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    return SDRK_OK;
}
}  // namespace

extern "C" {

int sdrk_version(void) { return SDRK_VERSION; }

const char* sdrk_last_error(void) { return g_last_error.c_str(); }

int sdrk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int sdrk_device_info(int device, char* buf, size_t buf_len) {
    if (!buf || buf_len == 0) return fail(SDRK_ERR_INVALID, "buf is NULL/empty");
    int st = check_device(device);
    if (st != SDRK_OK) return st;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buf_len, "%s %s, %d CUs, %.1f GiB, %d MHz, pci %04x:%02x:%02x.0", prop.gcnArchName, prop.name,
             prop.multiProcessorCount, (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0),
             prop.clockRate / 1000, prop.pciDomainID, prop.pciBusID, prop.pciDeviceID);
    return SDRK_OK;
}

int sdrk_dev_alloc(int device, size_t bytes, void** d_ptr) {
    if (!d_ptr) return fail(SDRK_ERR_INVALID, "d_ptr is NULL");
    *d_ptr = nullptr;
    int st = check_device(device);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 1));
    return SDRK_OK;
}

int sdrk_dev_mem_info(int device, size_t* free_bytes, size_t* total_bytes) {
    if (!free_bytes || !total_bytes) return fail(SDRK_ERR_INVALID, "free_bytes / total_bytes is NULL");
    int st = check_device(device);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
    return SDRK_OK;
}

int sdrk_dev_free(int device, void* d_ptr) {
    if (!d_ptr) return SDRK_OK;
    int st = check_device(device);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(d_ptr));
    return SDRK_OK;
}

int sdrk_dev_alloc_stream_pair(int device, size_t in_bytes, size_t out_bytes, int candidates, sdrk_plan* plan,
                               void** d_in, void** d_out, float* probe_ms, int* chosen) {
    if (!d_in || !d_out) return fail(SDRK_ERR_INVALID, "d_in or d_out is NULL");
    *d_in = *d_out = nullptr;
    size_t plan_frames = 0;
    if (plan) {
        if (plan->device != device) return fail(SDRK_ERR_INVALID, "plan is on device %d, not %d", plan->device, device);
        plan_frames = in_bytes / ((size_t)plan->nfft * sizeof(float2));
        if (plan_frames == 0 || out_bytes < plan_frames * (size_t)plan->nfft * sizeof(float))
            return fail(SDRK_ERR_INVALID, "buffers do not hold whole frames of the plan's length");
    }
    if (chosen) *chosen = 0;
    g_placement = PlacementReport();                  // whatever happens below, sdrk_placement_report never describes an older call
    if (candidates < 1) candidates = 1;
    if (candidates > 16) candidates = 16;
    int st = check_device(device);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    HIP_TRY(hipMalloc(d_in, in_bytes ? in_bytes : 1));
    // probe = the no-arithmetic streaming kernel with the spectrum path's 2:1 traffic shape over the pair (its
    // first 2^20 frame-equivalents: a short prefix mispredicts the intermediate levels); below 2^13
    // frame-equivalents the levels do not separate, and nothing is tuned
    size_t pf = in_bytes / 32768 < out_bytes / 16384 ? in_bytes / 32768 : out_bytes / 16384;
    if (pf > ((size_t)1 << 20)) pf = (size_t)1 << 20;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        while (candidates > 1 && (size_t)candidates * out_bytes + ((size_t)1 << 30) > free_b) --candidates;
    }
    if (pf < ((size_t)1 << 13)) candidates = 1;
    if (plan && plan_frames > ((size_t)1 << 32) / (size_t)plan->nfft) plan_frames = ((size_t)1 << 32) / (size_t)plan->nfft;
    std::vector<void*> cand((size_t)candidates, nullptr);
    std::vector<float> ms((size_t)candidates, 0.0f);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t s = nullptr;
    hipError_t e = hipSuccess;
    if (candidates > 1) {
        e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    }
    int n_ok = 0;
    PlacementReport rep;
    auto launch_on = [&](void* out) {
        return [&, out]() -> hipError_t {
            if (plan)                                 // the plan's own transform over the pair: what will really run
                return probe_launch_result(plan_launch(plan, *d_in, plan_frames, (size_t)plan->nfft, out, sdrk::EPI_LOGPSD, s));
            return sdrk::launch_stream_mix(*d_in, out, pf, prop.multiProcessorCount, s);
        };
    };
    for (int c = 0; c < candidates && e == hipSuccess; ++c) {
        // earlier candidates stay allocated, so each new one lands somewhere else
        if (hipMalloc(&cand[(size_t)c], out_bytes ? out_bytes : 1) != hipSuccess) {
            (void)hipGetLastError();
            cand[(size_t)c] = nullptr;
            break;
        }
        ++n_ok;
        if (candidates == 1) break;
        if (c == 0) e = placement_warm_up(s, launch_on(cand[0]), rep);          // (see "placement probes" above)
        if (e == hipSuccess) e = placement_time(s, e0, e1, launch_on(cand[(size_t)c]), &ms[(size_t)c]);
    }
    if (e == hipSuccess && n_ok > 1) {                                            // candidate 0 again, after the last one
        rep.first_ms = ms[0];
        e = placement_time(s, e0, e1, launch_on(cand[0]), &rep.retimed_first_ms);
        if (e == hipSuccess) ms[0] = rep.retimed_first_ms < ms[0] ? rep.retimed_first_ms : ms[0];
    }
    if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    int best = 0;
    for (int c = 1; c < n_ok; ++c)
        if (ms[(size_t)c] < ms[(size_t)best]) best = c;
    if (e != hipSuccess || n_ok == 0) {
        for (void* p : cand) if (p) (void)hipFree(p);
        (void)hipFree(*d_in);
        *d_in = nullptr;
        if (e != hipSuccess) return probe_fail(e, "placement probe failed");
        return fail(SDRK_ERR_NOMEM, "could not allocate %zu bytes for the output buffer", out_bytes);
    }
    for (int c = 0; c < n_ok; ++c) {
        if (probe_ms) probe_ms[c] = (c == 0 && n_ok > 1) ? rep.first_ms : ms[(size_t)c];   // [0]: as first timed; re-timed: sdrk_placement_report
        if (c != best) (void)hipFree(cand[(size_t)c]);
    }
    if (probe_ms) for (int c = n_ok; c < candidates; ++c) probe_ms[c] = 0.0f;
    *d_out = cand[(size_t)best];
    if (chosen) *chosen = best;
    rep.candidates = n_ok;
    rep.chosen = best;
    rep.chosen_ms = ms[(size_t)best];
    g_placement = rep;
    return SDRK_OK;
}

int sdrk_memcpy_h2d(int device, void* d_dst, const void* h_src, size_t bytes) {
    if (bytes == 0) return SDRK_OK;
    if (!d_dst || !h_src) return fail(SDRK_ERR_INVALID, "NULL pointer");
    int st = check_device(device);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return SDRK_OK;
}

int sdrk_memcpy_d2h(int device, void* h_dst, const void* d_src, size_t bytes) {
    if (bytes == 0) return SDRK_OK;
    if (!h_dst || !d_src) return fail(SDRK_ERR_INVALID, "NULL pointer");
    int st = check_device(device);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return SDRK_OK;
}

int sdrk_host_alloc(size_t bytes, void** h_ptr) {
    if (!h_ptr) return fail(SDRK_ERR_INVALID, "h_ptr is NULL");
    *h_ptr = nullptr;
    if (bytes == 0) return fail(SDRK_ERR_INVALID, "bytes must be >= 1");
    int st = check_device(0);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipHostMalloc(h_ptr, bytes, hipHostMallocPortable));
    std::lock_guard<std::mutex> g(pinned_ranges().m);
    pinned_ranges().r[reinterpret_cast<uintptr_t>(*h_ptr)] = {bytes, true};
    return SDRK_OK;
}

int sdrk_host_free(void* h_ptr) {
    if (!h_ptr) return SDRK_OK;
    {
        std::lock_guard<std::mutex> g(pinned_ranges().m);
        auto it = pinned_ranges().r.find(reinterpret_cast<uintptr_t>(h_ptr));
        if (it == pinned_ranges().r.end() || !it->second.second)
            return fail(SDRK_ERR_INVALID, "pointer was not returned by sdrk_host_alloc");
        pinned_ranges().r.erase(it);
    }
    HIP_TRY(hipHostFree(h_ptr));
    return SDRK_OK;
}

int sdrk_host_register(void* h_ptr, size_t bytes) {
    if (!h_ptr || bytes == 0) return fail(SDRK_ERR_INVALID, "NULL pointer or zero bytes");
    int st = check_device(0);
    if (st != SDRK_OK) return st;
    {
        std::lock_guard<std::mutex> g(pinned_ranges().m);
        if (pinned_ranges().r.count(reinterpret_cast<uintptr_t>(h_ptr)))
            return fail(SDRK_ERR_INVALID, "range is already registered");
    }
    HIP_TRY(hipHostRegister(h_ptr, bytes, hipHostRegisterPortable));
    std::lock_guard<std::mutex> g(pinned_ranges().m);
    pinned_ranges().r[reinterpret_cast<uintptr_t>(h_ptr)] = {bytes, false};
    return SDRK_OK;
}

int sdrk_host_unregister(void* h_ptr) {
    if (!h_ptr) return SDRK_OK;
    {
        std::lock_guard<std::mutex> g(pinned_ranges().m);
        auto it = pinned_ranges().r.find(reinterpret_cast<uintptr_t>(h_ptr));
        if (it == pinned_ranges().r.end() || it->second.second)
            return fail(SDRK_ERR_INVALID, "pointer was not registered with sdrk_host_register");
        pinned_ranges().r.erase(it);
    }
    HIP_TRY(hipHostUnregister(h_ptr));
    return SDRK_OK;
}

int sdrk_host_is_pinned(const void* h_ptr, size_t bytes) { return pinned_ranges().covers(h_ptr, bytes) ? 1 : 0; }

int sdrk_plan_create(int device, int nfft, size_t max_batch, int window_kind, const float* window,
                     float eps, int shift, sdrk_plan** out) {
    return sdrk_plan_create_ex(device, nfft, max_batch, window_kind, window, eps, shift, 0u, out);
}

int sdrk_plan_create_ex(int device, int nfft, size_t max_batch, int window_kind, const float* window,
                        float eps, int shift, unsigned flags, sdrk_plan** out) {
    if (!out) return fail(SDRK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (flags & ~(unsigned)(SDRK_PLAN_FUSED64K | SDRK_PLAN_OVERLAP_PASSES | SDRK_PLAN_TUNE_STAGING | SDRK_PLAN_TILED64K))
        return fail(SDRK_ERR_INVALID, "unknown plan flags 0x%x", flags);
    if ((flags & SDRK_PLAN_TILED64K) && (flags & SDRK_PLAN_FUSED64K))
        return fail(SDRK_ERR_INVALID, "SDRK_PLAN_TILED64K and SDRK_PLAN_FUSED64K exclude each other");
    if ((flags & SDRK_PLAN_TILED64K) && nfft != 65536)
        return fail(SDRK_ERR_INVALID, "SDRK_PLAN_TILED64K applies to nfft = 65536 only (got %d)", nfft);
    if ((flags & SDRK_PLAN_OVERLAP_PASSES) && (!is_pow2(nfft) || nfft < (1 << 15)))
        return fail(SDRK_ERR_INVALID, "SDRK_PLAN_OVERLAP_PASSES applies to power-of-two nfft >= 32768 (got %d)", nfft);
    if ((flags & SDRK_PLAN_OVERLAP_PASSES) && (flags & SDRK_PLAN_FUSED64K))
        return fail(SDRK_ERR_INVALID, "SDRK_PLAN_OVERLAP_PASSES and SDRK_PLAN_FUSED64K exclude each other");
    if ((flags & SDRK_PLAN_FUSED64K) && nfft != 65536)
        return fail(SDRK_ERR_INVALID, "SDRK_PLAN_FUSED64K applies to nfft = 65536 only (got %d)", nfft);
    if (nfft < 2 || nfft > (1 << SDRK_MAX_LOG2_NFFT) || (!is_pow2(nfft) && nfft > (1 << (SDRK_MAX_LOG2_NFFT - 1))))
        return fail(SDRK_ERR_INVALID, "nfft=%d: must be in [2, 2^%d] (powers of two) or [2, 2^%d] (any other length)",
                    nfft, SDRK_MAX_LOG2_NFFT, SDRK_MAX_LOG2_NFFT - 1);
    if (max_batch == 0) return fail(SDRK_ERR_INVALID, "max_batch must be >= 1");
    if (window_kind < SDRK_WINDOW_RECT || window_kind > SDRK_WINDOW_CUSTOM)
        return fail(SDRK_ERR_INVALID, "unknown window_kind %d", window_kind);
    if (window_kind == SDRK_WINDOW_CUSTOM && !window)
        return fail(SDRK_ERR_INVALID, "SDRK_WINDOW_CUSTOM needs a window pointer");
    if (!(eps >= 0.0f)) return fail(SDRK_ERR_INVALID, "eps must be >= 0");
    int st = check_device(device);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SDRK_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                    device, prop.gcnArchName);

    sdrk_plan* p = new (std::nothrow) sdrk_plan();
    if (!p) return fail(SDRK_ERR_NOMEM, "out of host memory");
    p->device = device;
    p->nfft = nfft;
    p->max_batch = max_batch;
    p->eps = eps;
    p->shift = shift ? 1 : 0;
    p->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* env = getenv("SDRK_NUM_CUS")) {   // size the persistent grids as for a smaller device (a partition
        long v = atol(env);                           // mode, or the tests of the grid-smaller-than-work paths); selects no kernel
        if (v >= 1 && v < p->num_cus) p->num_cus = (int)v;
    }

#define PLAN_TRY(expr)                                                                     \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            int s__ = fail(e__ == hipErrorOutOfMemory ? SDRK_ERR_NOMEM : SDRK_ERR_HIP,     \
                           "%s failed: %s", #expr, hipGetErrorString(e__));                \
            sdrk_plan_destroy(p);                                                          \
            return s__;                                                                    \
        }                                                                                  \
    } while (0)

    PLAN_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    PLAN_TRY(hipEventCreate(&p->ev0));
    PLAN_TRY(hipEventCreate(&p->ev1));

    // window
    if (window_kind != SDRK_WINDOW_RECT) {
        std::vector<float> w(nfft);
        if (window_kind == SDRK_WINDOW_HANN) {
            // numpy.hanning(M): 0.5 - 0.5 cos(2 pi n / (M-1)); M == 1 would be [1.0]
            for (int n = 0; n < nfft; ++n)
                w[n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * (double)n / (double)(nfft - 1)));
        } else {
            memcpy(w.data(), window, sizeof(float) * nfft);
        }
        PLAN_TRY(hipMalloc((void**)&p->d_window, sizeof(float) * nfft));
        PLAN_TRY(hipMemcpy(p->d_window, w.data(), sizeof(float) * nfft, hipMemcpyHostToDevice));
    }
    if (!is_pow2(nfft)) {
        // Bluestein: inner power-of-two plan of size M >= 2N-1, chirp table, spectrum of the chirp filter
        int M = 1;
        while (M < 2 * nfft - 1) M <<= 1;
        p->blu_m = M;
        // work buffers of the five-pass form: 128 MiB each; the one-kernel form (M <= 16384) needs none and takes any batch whole
        const bool one_kernel = sdrk::blu_fused_supports(M);
        size_t frames = ((size_t)128 << 20) / ((size_t)M * sizeof(float2));
        if (frames < 1) frames = 1;
        if (frames > max_batch || one_kernel) frames = max_batch;
        p->blu_frames = frames;
        int st2 = sdrk_plan_create(device, M, frames, SDRK_WINDOW_RECT, nullptr, 0.0f, 0, &p->blu_inner);
        if (st2 != SDRK_OK) { sdrk_plan_destroy(p); return st2; }
        std::vector<float2> c(nfft), b(M, make_float2(0.f, 0.f));
        for (long long n = 0; n < nfft; ++n) {
            const long long r = (n * n) % (2LL * nfft);          // n^2 mod 2N keeps the phase exact
            const double ang = M_PI * (double)r / (double)nfft;
            c[n] = make_float2((float)std::cos(ang), (float)std::sin(ang));
            b[n] = c[n];
            if (n) b[M - n] = c[n];
        }
        PLAN_TRY(hipMalloc((void**)&p->d_blu_chirp, sizeof(float2) * nfft));
        PLAN_TRY(hipMemcpy(p->d_blu_chirp, c.data(), sizeof(float2) * nfft, hipMemcpyHostToDevice));
        PLAN_TRY(hipMalloc((void**)&p->d_blu_a, (one_kernel ? 1 : frames) * (size_t)M * sizeof(float2)));   // (also carries b[] to its transform below)
        if (!one_kernel) PLAN_TRY(hipMalloc((void**)&p->d_blu_b, frames * (size_t)M * sizeof(float2)));
        PLAN_TRY(hipMalloc((void**)&p->d_blu_bspec, (size_t)M * sizeof(float2)));
        PLAN_TRY(hipMemcpy(p->d_blu_a, b.data(), sizeof(float2) * M, hipMemcpyHostToDevice));
        st2 = plan_launch(p->blu_inner, p->d_blu_a, 1, (size_t)M, p->d_blu_bspec, sdrk::EPI_COMPLEX, p->stream);
        if (st2 != SDRK_OK) { sdrk_plan_destroy(p); return st2; }
        PLAN_TRY(hipStreamSynchronize(p->stream));
        // window (if any) was uploaded above; nothing else of the power-of-two setup applies
        *out = p;
        return SDRK_OK;
    }
    // twiddles of the in-LDS transform: W_N for N <= 16384 (fft4096.hip, fft_lds.hip, fft_small.hip); the
    // two-pass plans carry their own tables below.  (N = 32768 also fits one CU — 32 points per thread, float-plane
    // LDS exchanges — and was built and measured in round 3: no faster than the two passes)
    if (nfft <= 16384) {
        std::vector<float2> t(nfft);
        for (int m = 0; m < nfft; ++m) t[m] = twiddle(m, nfft);
        PLAN_TRY(hipMalloc((void**)&p->d_twiddle, sizeof(float2) * nfft));
        PLAN_TRY(hipMemcpy(p->d_twiddle, t.data(), sizeof(float2) * nfft, hipMemcpyHostToDevice));
    } else {
        // scratch between the two passes: up to 192 MiB of complex64 frames.  It has to stay in the 256 MiB
        // Infinity Cache between the col pass that writes it and the row pass that reads it, and per-launch
        // costs favour few large chunks: measured on config 3 (N = 65536), 96 / 192 / 288 / 384 / 576 MiB give
        // 6.15 / 5.61 / 5.99 / 6.95 / 7.23 ms (the step past 256 MiB is the cache being outrun).
        size_t scratch_mb = 192;
        if (const char* env = getenv("SDRK_SCRATCH_MB")) {  // tuning knob (developer use)
            long v = atol(env);
            if (v >= 1 && v <= 65536) scratch_mb = (size_t)v;
        }
        size_t frames = (scratch_mb << 20) / ((size_t)nfft * sizeof(float2));
        if (frames < 1) frames = 1;
        if (frames > max_batch) frames = max_batch;
        p->scratch_frames = frames;
        PLAN_TRY(hipMalloc((void**)&p->d_scratch, frames * (size_t)nfft * sizeof(float2)));
        int la = 0, lm = 0;
        if (!sdrk::fft_tiled2_split(nfft, &la, &lm)) {
            sdrk_plan_destroy(p);
            return fail(SDRK_ERR_UNSUPPORTED, "no kernel for nfft=%d", nfft);
        }
        const int A = 1 << la, M = 1 << lm, TA = A / 16;
        std::vector<float2> t((size_t)4096 + (size_t)TA * M + (size_t)M * 16);
        for (int m = 0; m < A; ++m) t[m] = twiddle(m, A);
        for (int m = 0; m < M; ++m) t[2048 + m] = twiddle(m, M);
        for (int tau = 0; tau < TA; ++tau)
            for (int m = 0; m < M; ++m) t[4096 + (size_t)tau * M + m] = twiddle((double)m * tau, (double)nfft);
        for (int m = 0; m < M; ++m)
            for (int q = 0; q < 16; ++q)
                t[4096 + (size_t)TA * M + (size_t)m * 16 + q] = twiddle((double)m * TA * q, (double)nfft);
        PLAN_TRY(hipMalloc((void**)&p->d_tw_2p, sizeof(float2) * t.size()));
        PLAN_TRY(hipMemcpy(p->d_tw_2p, t.data(), sizeof(float2) * t.size(), hipMemcpyHostToDevice));
        p->tiled2 = true;
        if (flags & SDRK_PLAN_OVERLAP_PASSES) {
            // second stream, the events that chain the two, and the split of the CUs between the roles: 3/4 of
            // the device's CUs worth of col workgroups, 1/2 worth of row workgroups (the least slow of the splits
            // tried; SDRK_OVL_COL_CUS / SDRK_OVL_ROW_CUS override it for the sweep in tools/overlap_probe.py)
            PLAN_TRY(hipStreamCreateWithFlags(&p->stream2, hipStreamNonBlocking));
            PLAN_TRY(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
            for (int h = 0; h < 2; ++h) {
                PLAN_TRY(hipEventCreateWithFlags(&p->ev_col[h], hipEventDisableTiming));
                PLAN_TRY(hipEventCreateWithFlags(&p->ev_row[h], hipEventDisableTiming));
            }
            p->col_cus = p->num_cus * 3 / 4 > 0 ? p->num_cus * 3 / 4 : 1;
            p->row_cus = p->num_cus / 2 > 0 ? p->num_cus / 2 : 1;
            if (const char* c = getenv("SDRK_OVL_COL_CUS")) { long v = atol(c); if (v >= 1 && v <= 4096) p->col_cus = (int)v; }
            if (const char* c = getenv("SDRK_OVL_ROW_CUS")) { long v = atol(c); if (v >= 1 && v <= 4096) p->row_cus = (int)v; }
        }
    }
    // Single-launch, XCD-resident form of N = 65536 (fft_fused64k.hip); shares the tables of the tiled path.  Forced by
    // SDRK_PLAN_FUSED64K; otherwise the default for launches of FUSED_AUTO_MIN_FRAMES frames or more, unless SDRK_PLAN_TILED64K
    // or SDRK_PLAN_OVERLAP_PASSES asks for the two launches or the device's CUs do not make whole sets (32 workgroups per XCD).
    // (A CU count that misdescribes the device — SDRK_NUM_CUS = 96 on eight XCDs — makes the first persistent launch fail its
    // set formation; the plan then falls back for good, which is how the suite tests the fall-back.)
    const bool fused_auto = nfft == 65536 && !(flags & (SDRK_PLAN_FUSED64K | SDRK_PLAN_TILED64K | SDRK_PLAN_OVERLAP_PASSES)) &&
                            p->num_cus >= 32 && p->num_cus % 32 == 0;
    if ((flags & SDRK_PLAN_FUSED64K) || fused_auto) {
        p->fused64k = (flags & SDRK_PLAN_FUSED64K) != 0;
        p->fused_auto = fused_auto;
        PLAN_TRY(hipMalloc(&p->d_fused_ring, sdrk::fused64k_ring_bytes()));
        PLAN_TRY(hipMalloc((void**)&p->d_fused_ctrl, sdrk::fused64k_ctrl_words() * sizeof(unsigned)));
        PLAN_TRY(hipHostMalloc((void**)&p->h_fused_err, FUSED_MAILBOX * 16 * sizeof(unsigned), hipHostMallocDefault));
        memset(p->h_fused_err, 0, FUSED_MAILBOX * 16 * sizeof(unsigned));
    }
#undef PLAN_TRY
    if (flags & SDRK_PLAN_TUNE_STAGING) {
        st = tune_staging(p);
        if (st != SDRK_OK) { sdrk_plan_destroy(p); return st; }
    }
    *out = p;
    return SDRK_OK;
}

int sdrk_plan_staging_probe(const sdrk_plan* p, float* probe_ms, int capacity, int* n) {
    if (!p || !n) return fail(SDRK_ERR_INVALID, "plan or n is NULL");
    *n = p->staging_probe_n;
    for (int i = 0; i < p->staging_probe_n && i < capacity && probe_ms; ++i) probe_ms[i] = p->staging_probe_ms[i];
    return SDRK_OK;
}

int sdrk_plan_fused_status(const sdrk_plan* p, unsigned* launches, int* fallen_back) {
    if (!p) return fail(SDRK_ERR_INVALID, "plan is NULL");
    if (launches) *launches = p->fused_launches;
    if (fallen_back) *fallen_back = p->fused_broken ? 1 : 0;
    return SDRK_OK;
}

int sdrk_plan_destroy(sdrk_plan* p) {
    if (!p) return SDRK_OK;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->stream2) { (void)hipStreamSynchronize(p->stream2); (void)hipStreamDestroy(p->stream2); }
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    for (int h = 0; h < 2; ++h) {
        if (p->ev_col[h]) (void)hipEventDestroy(p->ev_col[h]);
        if (p->ev_row[h]) (void)hipEventDestroy(p->ev_row[h]);
    }
    if (p->d_window) (void)hipFree(p->d_window);
    if (p->d_twiddle) (void)hipFree(p->d_twiddle);
    if (p->d_tw_2p) (void)hipFree(p->d_tw_2p);
    if (p->d_scratch) (void)hipFree(p->d_scratch);
    if (p->blu_inner) (void)sdrk_plan_destroy(p->blu_inner);
    if (p->d_blu_chirp) (void)hipFree(p->d_blu_chirp);
    if (p->d_blu_bspec) (void)hipFree(p->d_blu_bspec);
    if (p->d_blu_a) (void)hipFree(p->d_blu_a);
    if (p->d_blu_b) (void)hipFree(p->d_blu_b);
    if (p->h_small_in) (void)hipHostFree(p->h_small_in);
    if (p->h_small_out) (void)hipHostFree(p->h_small_out);
    if (p->h_small_flag) (void)hipHostFree(p->h_small_flag);
    if (p->d_fused_ring) (void)hipFree(p->d_fused_ring);
    if (p->d_fused_ctrl) (void)hipFree(p->d_fused_ctrl);
    if (p->h_fused_err) (void)hipHostFree(p->h_fused_err);
    if (p->d_in) (void)hipFree(p->d_in);
    if (p->d_out) (void)hipFree(p->d_out);
    if (p->d_feat) (void)hipFree(p->d_feat);
    if (p->s_h2d) (void)hipStreamSynchronize(p->s_h2d);
    if (p->s_d2h) (void)hipStreamSynchronize(p->s_d2h);
    for (auto& sl : p->slot) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.ev_in) (void)hipEventDestroy(sl.ev_in);
        if (sl.ev_k) (void)hipEventDestroy(sl.ev_k);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
    }
    if (p->s_h2d) (void)hipStreamDestroy(p->s_h2d);
    if (p->s_d2h) (void)hipStreamDestroy(p->s_d2h);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return SDRK_OK;
}

int sdrk_plan_nfft(const sdrk_plan* p) { return p ? p->nfft : fail(SDRK_ERR_INVALID, "plan is NULL"); }
int sdrk_plan_device(const sdrk_plan* p) { return p ? p->device : fail(SDRK_ERR_INVALID, "plan is NULL"); }

int sdrk_exec_host(sdrk_plan* p, const void* iq, size_t n_frames, size_t frame_stride, float* out_db) {
    return exec_host_common(p, iq, n_frames, frame_stride, out_db, sdrk::EPI_LOGPSD);
}

int sdrk_exec_fft_host(sdrk_plan* p, const void* iq, size_t n_frames, size_t frame_stride,
                       void* out_c64) {
    return exec_host_common(p, iq, n_frames, frame_stride, out_c64, sdrk::EPI_COMPLEX);
}

int sdrk_welch_psd_host(sdrk_plan* p, const void* iq, size_t n_frames, size_t frame_stride,
                        float scale, float* out_psd) {
    int st = check_exec_args(p, iq, n_frames, frame_stride, out_psd);
    if (st != SDRK_OK) return st;
    if (n_frames == 0) return fail(SDRK_ERR_INVALID, "welch needs at least one frame");
    if (n_frames > p->max_batch)
        return fail(SDRK_ERR_INVALID, "n_frames %zu exceeds the plan's max_batch %zu", n_frames, p->max_batch);
    HIP_TRY(hipSetDevice(p->device));
    const size_t nfft = (size_t)p->nfft;
    const size_t in_bytes = ((n_frames - 1) * frame_stride + nfft) * sizeof(float2);
    // spectra are produced in chunks into d_out; the column sums accumulate per chunk on the host side
    // of the call only through `scale` (each chunk adds scale * sum), so one small device row suffices.
    const size_t chunk = ((size_t)256 << 20) / (nfft * sizeof(float2)) ? ((size_t)256 << 20) / (nfft * sizeof(float2)) : 1;
    const size_t spec_frames = n_frames < chunk ? n_frames : chunk;
    st = grow(p->device, &p->d_in, &p->in_cap, in_bytes);
    if (st != SDRK_OK) return st;
    st = grow(p->device, &p->d_out, &p->out_cap, spec_frames * nfft * sizeof(float2) + nfft * sizeof(float));
    if (st != SDRK_OK) return st;
    float* d_row = reinterpret_cast<float*>(static_cast<char*>(p->d_out) + spec_frames * nfft * sizeof(float2));
    HIP_TRY(hipMemcpyAsync(p->d_in, iq, in_bytes, hipMemcpyHostToDevice, p->stream));
    std::vector<float> row(nfft), total(nfft, 0.0f);
    for (size_t f0 = 0; f0 < n_frames; f0 += spec_frames) {
        const size_t nf = n_frames - f0 < spec_frames ? n_frames - f0 : spec_frames;
        st = plan_launch(p, static_cast<const float2*>(p->d_in) + f0 * frame_stride, nf, frame_stride, p->d_out,
                         sdrk::EPI_COMPLEX, p->stream);
        if (st != SDRK_OK) return st;
        hipError_t e = sdrk::launch_power_mean(p->d_out, nf, p->nfft, scale, d_row, p->stream);
        if (e != hipSuccess) return fail(SDRK_ERR_HIP, "power_mean launch failed: %s", hipGetErrorString(e));
        HIP_TRY(hipMemcpyAsync(row.data(), d_row, nfft * sizeof(float), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        st = fused_check(p);
        if (st != SDRK_OK) return st;
        for (size_t k = 0; k < nfft; ++k) total[k] += row[k];   // <= a handful of chunks
    }
    memcpy(out_psd, total.data(), nfft * sizeof(float));
    return SDRK_OK;
}

int sdrk_exec_device(sdrk_plan* p, const void* d_iq, size_t n_frames, size_t frame_stride,
                     float* d_out_db, void* stream) {
    int st = check_exec_args(p, d_iq, n_frames, frame_stride, d_out_db);
    if (st != SDRK_OK || n_frames == 0) return st;
    HIP_TRY(hipSetDevice(p->device));
    return plan_launch(p, d_iq, n_frames, frame_stride, d_out_db, sdrk::EPI_LOGPSD,
                       stream ? static_cast<hipStream_t>(stream) : p->stream);
}

int sdrk_plan_sync(sdrk_plan* p) {
    if (!p) return fail(SDRK_ERR_INVALID, "plan is NULL");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return fused_check(p);
}

int sdrk_exec_device_timed(sdrk_plan* p, const void* d_iq, size_t n_frames, size_t frame_stride,
                           float* d_out_db, int launches, float* elapsed_ms) {
    if (!elapsed_ms || launches < 1) return fail(SDRK_ERR_INVALID, "bad launches/elapsed_ms");
    int st = check_exec_args(p, d_iq, n_frames, frame_stride, d_out_db);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventRecord(p->ev0, p->stream));
    for (int i = 0; i < launches; ++i) {
        st = plan_launch(p, d_iq, n_frames, frame_stride, d_out_db, sdrk::EPI_LOGPSD, p->stream);
        if (st != SDRK_OK) return st;
    }
    HIP_TRY(hipEventRecord(p->ev1, p->stream));
    HIP_TRY(hipEventSynchronize(p->ev1));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, p->ev0, p->ev1));
    return fused_check(p);
}

int sdrk_exec_device_timed_each(sdrk_plan* p, const void* d_iq, size_t n_frames, size_t frame_stride,
                                float* d_out_db, int launches, float* each_ms) {
    if (!each_ms || launches < 1 || launches > 4096) return fail(SDRK_ERR_INVALID, "bad launches/each_ms");
    int st = check_exec_args(p, d_iq, n_frames, frame_stride, d_out_db);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(p->device));
    std::vector<hipEvent_t> ev((size_t)launches + 1, nullptr);
    auto cleanup = [&] { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); };
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) { cleanup(); return fail(SDRK_ERR_HIP, "hipEventCreate failed"); }
    hipError_t e = hipEventRecord(ev[0], p->stream);
    for (int i = 0; i < launches && e == hipSuccess; ++i) {
        st = plan_launch(p, d_iq, n_frames, frame_stride, d_out_db, sdrk::EPI_LOGPSD, p->stream);
        if (st != SDRK_OK) { (void)hipStreamSynchronize(p->stream); cleanup(); return st; }
        e = hipEventRecord(ev[(size_t)i + 1], p->stream);
    }
    if (e == hipSuccess) e = hipEventSynchronize(ev[(size_t)launches]);
    for (int i = 0; i < launches && e == hipSuccess; ++i) e = hipEventElapsedTime(&each_ms[i], ev[i], ev[(size_t)i + 1]);
    cleanup();
    if (e != hipSuccess) return fail(SDRK_ERR_HIP, "timed launches failed: %s", hipGetErrorString(e));
    return fused_check(p);
}

int sdrk_plan_tune_scratch(sdrk_plan* p, const void* d_iq, size_t n_frames, size_t frame_stride, float* d_out_db,
                           int candidates, float* probe_ms, int* chosen) {
    if (chosen) *chosen = 0;
    int st = check_exec_args(p, d_iq, n_frames, frame_stride, d_out_db);
    if (st != SDRK_OK) return st;
    g_placement = PlacementReport();
    if (!p->d_scratch || p->scratch_frames == 0 || n_frames == 0 || takes_fused(p, n_frames)) {
        // no scratch on this workload's path (one-pass lengths; the persistent N = 65536 launch): nothing to place
        if (probe_ms) for (int c = 0; c < candidates; ++c) probe_ms[c] = 0.0f;
        return SDRK_OK;
    }
    if (candidates < 2) return SDRK_OK;
    if (candidates > 16) candidates = 16;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    const size_t bytes = p->scratch_frames * (size_t)p->nfft * sizeof(float2);
    std::vector<float2*> cand((size_t)candidates, nullptr);
    std::vector<float> ms((size_t)candidates, 0.0f);
    cand[0] = p->d_scratch;                                               // candidate 0 = the plan's present scratch
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    PlacementReport rep;
    auto launch = [&]() -> hipError_t {
        return probe_launch_result(plan_launch(p, d_iq, n_frames, frame_stride, d_out_db, sdrk::EPI_LOGPSD, p->stream));
    };
    if (e == hipSuccess) e = placement_warm_up(p->stream, launch, rep);    // (see "placement probes" above)
    int n_ok = 0;
    for (int c = 0; c < candidates && e == hipSuccess; ++c) {
        // earlier candidates stay allocated, so each new one lands somewhere else
        if (c > 0 && hipMalloc((void**)&cand[(size_t)c], bytes) != hipSuccess) {
            (void)hipGetLastError();
            cand[(size_t)c] = nullptr;
            break;
        }
        ++n_ok;
        p->d_scratch = cand[(size_t)c];
        e = placement_time(p->stream, e0, e1, launch, &ms[(size_t)c]);
    }
    if (e == hipSuccess && n_ok > 1) {                                     // candidate 0 again, after the last one
        p->d_scratch = cand[0];
        rep.first_ms = ms[0];
        e = placement_time(p->stream, e0, e1, launch, &rep.retimed_first_ms);
    }
    (void)hipStreamSynchronize(p->stream);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    // a candidate replaces the present scratch only if it beats BOTH timings of it by one per cent
    int best = 0;
    if (e == hipSuccess && n_ok > 1) {
        const float ref0 = rep.retimed_first_ms < ms[0] ? rep.retimed_first_ms : ms[0];
        float best_ms = ref0 * 0.99f;
        for (int c = 1; c < n_ok; ++c)
            if (ms[(size_t)c] < best_ms) { best = c; best_ms = ms[(size_t)c]; }
        rep.chosen_ms = best ? ms[(size_t)best] : ref0;
    }
    p->d_scratch = cand[(size_t)best];
    for (int c = 0; c < n_ok; ++c) {
        if (probe_ms) probe_ms[c] = ms[(size_t)c];
        if (c != best) (void)hipFree(cand[(size_t)c]);
    }
    if (probe_ms) for (int c = n_ok; c < candidates; ++c) probe_ms[c] = 0.0f;
    if (chosen) *chosen = best;
    rep.candidates = n_ok;
    rep.chosen = best;
    g_placement = rep;
    if (e != hipSuccess) return probe_fail(e, "scratch placement probe failed");
    return fused_check(p);
}

int sdrk_placement_report(float* warm_ms, int* warm_launches, float* first_ms, float* retimed_first_ms, float* chosen_ms) {
    const PlacementReport& r = g_placement;
    if (warm_ms) *warm_ms = r.warm_ms;
    if (warm_launches) *warm_launches = r.warm_launches;
    if (first_ms) *first_ms = r.first_ms;
    if (retimed_first_ms) *retimed_first_ms = r.retimed_first_ms;
    if (chosen_ms) *chosen_ms = r.chosen_ms;
    return r.candidates;
}

int sdrk_stream_ceiling_probe(int device, const void* d_in, void* d_out, size_t n_frames4096, int launches,
                              float* each_ms) {
    if (!d_in || !d_out || !each_ms || launches < 1 || launches > 4096 || n_frames4096 == 0)
        return fail(SDRK_ERR_INVALID, "bad argument");
    return timed_probe(device, launches, each_ms, "stream ceiling probe", [&](int cus, hipStream_t s) {
        return sdrk::launch_stream_mix(d_in, d_out, n_frames4096, cus, s);
    });
}

int sdrk_copy_probe(int device, const void* d_in, void* d_out, size_t bytes, int launches, float* each_ms) {
    if (!d_in || !d_out || !each_ms || launches < 1 || launches > 4096 || bytes < 16)
        return fail(SDRK_ERR_INVALID, "bad argument");
    // the fastest of three grid sizes (by median): a ceiling should not depend on the probe's own launch shape
    std::vector<float> t((size_t)launches), best;
    float best_med = 0.0f;
    for (int bpc : {3, 4, 16}) {
        int st = timed_probe(device, launches, t.data(), "copy probe", [&](int cus, hipStream_t s) {
            return sdrk::launch_copy_1to1(d_in, d_out, bytes, cus, bpc, s);
        });
        if (st != SDRK_OK) return st;
        std::vector<float> sorted = t;
        std::sort(sorted.begin(), sorted.end());
        const float med = sorted[sorted.size() / 2];
        if (best.empty() || med < best_med) { best = t; best_med = med; }
    }
    memcpy(each_ms, best.data(), sizeof(float) * (size_t)launches);
    return SDRK_OK;
}

int sdrk_host_link_probe(int device, size_t bytes, double* h2d_gbps, double* d2h_gbps, double* duplex_gbps) {
    if (!h2d_gbps || !d2h_gbps || !duplex_gbps || bytes < (1u << 20)) return fail(SDRK_ERR_INVALID, "bad argument");
    int st = check_device(device);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(device));
    void *h_a = nullptr, *h_b = nullptr, *d_a = nullptr, *d_b = nullptr;
    hipStream_t s0 = nullptr, s1 = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    hipError_t e = hipHostMalloc(&h_a, bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(&h_b, bytes / 2, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&d_a, bytes);
    if (e == hipSuccess) e = hipMalloc(&d_b, bytes / 2);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventCreate(&e2);
    if (e == hipSuccess) memset(h_a, 1, bytes);
    float ms = 0.f;
    auto timed = [&](bool up, bool down, double* gbps, double moved) {
        for (int rep = 0; rep < 3 && e == hipSuccess; ++rep) {       // keep the last of three
            e = hipEventRecord(e0, s0);
            if (e == hipSuccess) e = hipStreamWaitEvent(s1, e0, 0);
            if (up && e == hipSuccess) e = hipMemcpyAsync(d_a, h_a, bytes, hipMemcpyHostToDevice, s0);
            if (down && e == hipSuccess) e = hipMemcpyAsync(h_b, d_b, bytes / 2, hipMemcpyDeviceToHost, s1);
            if (e == hipSuccess) e = hipEventRecord(e2, s1);
            if (e == hipSuccess) e = hipStreamWaitEvent(s0, e2, 0);
            if (e == hipSuccess) e = hipEventRecord(e1, s0);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        }
        if (e == hipSuccess) *gbps = moved / (ms * 1e-3) / 1e9;
    };
    timed(true, false, h2d_gbps, (double)bytes);
    timed(false, true, d2h_gbps, (double)(bytes / 2));
    // the spectrum path's mix: `bytes` up while bytes/2 come down; rate quoted on the upstream bytes
    timed(true, true, duplex_gbps, (double)bytes);
    if (s0) (void)hipStreamSynchronize(s0);
    if (s1) (void)hipStreamSynchronize(s1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e2) (void)hipEventDestroy(e2);
    if (s0) (void)hipStreamDestroy(s0);
    if (s1) (void)hipStreamDestroy(s1);
    if (h_a) (void)hipHostFree(h_a);
    if (h_b) (void)hipHostFree(h_b);
    if (d_a) (void)hipFree(d_a);
    if (d_b) (void)hipFree(d_b);
    if (e != hipSuccess) return fail(SDRK_ERR_HIP, "host link probe failed: %s", hipGetErrorString(e));
    return SDRK_OK;
}

int sdrk_host_threads(void) { return sdrk::CopyPool::get().helpers(); }

int sdrk_synth_fill(int device, uint32_t seed, uint64_t first_frame, size_t n_frames, int nfft,
                    void* d_iq, void* stream) {
    if (n_frames == 0) return SDRK_OK;
    if (!d_iq) return fail(SDRK_ERR_INVALID, "d_iq is NULL");
    if (nfft < 2 || (nfft & 1)) return fail(SDRK_ERR_INVALID, "nfft must be even and >= 2");
    int st = check_device(device);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = sdrk::launch_synth_fill(seed, first_frame, n_frames, nfft, d_iq, s);
    if (e != hipSuccess) return fail(SDRK_ERR_HIP, "synth launch failed: %s", hipGetErrorString(e));
    if (!stream) HIP_TRY(hipStreamSynchronize(s));
    return SDRK_OK;
}

/* ---- per-row reductions ---------------------------------------------------- */

namespace {

// Device scratch of the plan-less row entry points: one buffer per device, only ever grown, used under the
// device's lock (calls on one device serialise; different devices run concurrently).
struct RowScratch {
    std::mutex lock;
    void* buf = nullptr;
    size_t cap = 0;
};
RowScratch g_row_scratch[64];

struct RowScratchGuard {
    RowScratch* rs;
    explicit RowScratchGuard(int device) : rs(&g_row_scratch[device & 63]) { rs->lock.lock(); }
    ~RowScratchGuard() { rs->lock.unlock(); }
    int reserve(int device, size_t bytes) { return grow(device, &rs->buf, &rs->cap, bytes); }
};

constexpr size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// layout of the packed per-row results in a scratch buffer
struct FeatLayout {
    size_t rows_off, stats_off, thr_off, idx_off, cnt_off, planes_off, freqs_off, total;
    FeatLayout(size_t n_rows, int nfft, int max_peaks, bool stage_rows, bool peaks, bool planes = false) {
        size_t o = 0;
        rows_off = o;  o += stage_rows ? align256(n_rows * (size_t)nfft * sizeof(float)) : 0;
        stats_off = o; o += align256(n_rows * 16 * sizeof(double));
        thr_off = o;   o += align256(n_rows * sizeof(double));
        idx_off = o;   o += peaks ? align256(n_rows * (size_t)max_peaks * sizeof(int)) : 0;
        cnt_off = o;   o += peaks ? align256(n_rows * sizeof(int)) : 0;
        planes_off = o; o += planes ? align256(n_rows * SDRK_FEAT_PLANES * sizeof(double)) : 0;
        freqs_off = o;  o += planes ? align256((size_t)nfft * sizeof(double)) : 0;
        total = o;
    }
};

// the packed results of a batch -> finished planes (feature_finalize_kernel) -> the caller's host arrays
int planes_to_host(char* base, const FeatLayout& L, bool peaks, size_t n_rows, int nfft, float gamma, int max_peaks,
                   const double* freqs, void* out_planes, int32_t* out_idx, hipStream_t s) {
    const double* d_freqs = nullptr;
    if (freqs) {
        HIP_TRY(hipMemcpyAsync(base + L.freqs_off, freqs, (size_t)nfft * sizeof(double), hipMemcpyHostToDevice, s));
        d_freqs = reinterpret_cast<const double*>(base + L.freqs_off);
    }
    hipError_t e = sdrk::launch_feature_finalize(reinterpret_cast<const double*>(base + L.stats_off),
                                                 reinterpret_cast<const double*>(base + L.thr_off),
                                                 peaks ? reinterpret_cast<const int*>(base + L.idx_off) : nullptr,
                                                 peaks ? reinterpret_cast<const int*>(base + L.cnt_off) : nullptr, n_rows, nfft,
                                                 gamma, max_peaks, d_freqs, reinterpret_cast<double*>(base + L.planes_off), s);
    if (e != hipSuccess) return fail(SDRK_ERR_HIP, "feature finalize launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(out_planes, base + L.planes_off, n_rows * SDRK_FEAT_PLANES * sizeof(double), hipMemcpyDeviceToHost, s));
    if (peaks)
        HIP_TRY(hipMemcpyAsync(out_idx, base + L.idx_off, n_rows * (size_t)max_peaks * sizeof(int), hipMemcpyDeviceToHost, s));
    return SDRK_OK;
}

int device_cus(int device, int* cus) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    *cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return SDRK_OK;
}

}  // namespace

namespace {
int row_features_impl(int device, const float* rows, int rows_on_device, size_t n_rows, int nfft, int rank,
                      float gamma, int min_distance, int max_peaks, double* out_stats, double* out_thr,
                      int32_t* out_idx, int32_t* out_count, const double* freqs, void* out_planes) {
    if (n_rows == 0) return SDRK_OK;
    if (!rows || (!out_stats && !out_planes)) return fail(SDRK_ERR_INVALID, "rows or the result pointer is NULL");
    if (nfft < 1) return fail(SDRK_ERR_INVALID, "nfft must be >= 1");
    const bool peaks = out_idx != nullptr || out_count != nullptr;
    if (peaks && (!out_idx || (!out_count && !out_planes) || max_peaks < 1 || min_distance < 1))
        return fail(SDRK_ERR_INVALID, "peaks need out_idx, out_count and max_peaks, min_distance >= 1");
    int st = check_device(device);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(device));
    int cus = 256;
    st = device_cus(device, &cus);
    if (st != SDRK_OK) return st;
    RowScratchGuard g(device);
    const FeatLayout L(n_rows, nfft, max_peaks, !rows_on_device, peaks, out_planes != nullptr);
    st = g.reserve(device, L.total);
    if (st != SDRK_OK) return st;
    char* base = static_cast<char*>(g.rs->buf);
    const float* d_rows = rows;
    if (!rows_on_device) {
        HIP_TRY(hipMemcpy(base + L.rows_off, rows, n_rows * (size_t)nfft * sizeof(float), hipMemcpyHostToDevice));
        d_rows = reinterpret_cast<const float*>(base + L.rows_off);
    }
    if (peaks) HIP_TRY(hipMemsetAsync(base + L.idx_off, 0xFF, n_rows * (size_t)max_peaks * sizeof(int), nullptr));   // unused slots: -1
    hipError_t e = sdrk::launch_row_features(d_rows, n_rows, nfft, rank, gamma, min_distance, max_peaks,
                                             reinterpret_cast<double*>(base + L.stats_off),
                                             reinterpret_cast<double*>(base + L.thr_off),
                                             peaks ? reinterpret_cast<int*>(base + L.idx_off) : nullptr,
                                             peaks ? reinterpret_cast<int*>(base + L.cnt_off) : nullptr, cus, nullptr);
    if (e != hipSuccess) return fail(SDRK_ERR_HIP, "row_features launch failed: %s", hipGetErrorString(e));
    if (out_planes) {
        st = planes_to_host(base, L, peaks, n_rows, nfft, gamma, max_peaks, freqs, out_planes, out_idx, nullptr);
        if (st != SDRK_OK) return st;
        HIP_TRY(hipStreamSynchronize(nullptr));
        return SDRK_OK;
    }
    HIP_TRY(hipMemcpy(out_stats, base + L.stats_off, n_rows * 16 * sizeof(double), hipMemcpyDeviceToHost));
    if (out_thr) HIP_TRY(hipMemcpy(out_thr, base + L.thr_off, n_rows * sizeof(double), hipMemcpyDeviceToHost));
    if (peaks) {
        HIP_TRY(hipMemcpy(out_idx, base + L.idx_off, n_rows * (size_t)max_peaks * sizeof(int), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_count, base + L.cnt_off, n_rows * sizeof(int), hipMemcpyDeviceToHost));
    }
    return SDRK_OK;
}
}  // namespace

int sdrk_row_features(int device, const float* rows, int rows_on_device, size_t n_rows, int nfft, int rank,
                      float gamma, int min_distance, int max_peaks, double* out_stats, double* out_thr,
                      int32_t* out_idx, int32_t* out_count) {
    return row_features_impl(device, rows, rows_on_device, n_rows, nfft, rank, gamma, min_distance, max_peaks, out_stats,
                             out_thr, out_idx, out_count, nullptr, nullptr);
}

int sdrk_row_features_planes(int device, const float* rows, int rows_on_device, size_t n_rows, int nfft, int rank,
                             float gamma, int min_distance, int max_peaks, const double* freqs, void* out_planes,
                             int32_t* out_idx) {
    if (n_rows && !out_planes) return fail(SDRK_ERR_INVALID, "out_planes is NULL");
    return row_features_impl(device, rows, rows_on_device, n_rows, nfft, rank, gamma, min_distance, max_peaks, nullptr,
                             nullptr, out_idx, nullptr, freqs, out_planes);
}

int sdrk_row_stats(int device, const float* rows, int rows_on_device, size_t n_rows, int nfft, int rank,
                   double* out) {
    return sdrk_row_features(device, rows, rows_on_device, n_rows, nfft, rank, 0.0f, 1, 1, out, nullptr, nullptr, nullptr);
}

int sdrk_row_peaks(int device, const float* rows, int rows_on_device, size_t n_rows, int nfft,
                   const double* thresholds, int min_distance, int max_peaks, int32_t* out_idx,
                   int32_t* out_count) {
    if (n_rows == 0) return SDRK_OK;
    if (!rows || !thresholds || !out_idx || !out_count) return fail(SDRK_ERR_INVALID, "NULL pointer");
    if (nfft < 1 || max_peaks < 1 || min_distance < 1)
        return fail(SDRK_ERR_INVALID, "nfft, max_peaks and min_distance must be >= 1");
    int st = check_device(device);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(device));
    RowScratchGuard g(device);
    const FeatLayout L(n_rows, nfft, max_peaks, !rows_on_device, true);
    st = g.reserve(device, L.total);
    if (st != SDRK_OK) return st;
    char* base = static_cast<char*>(g.rs->buf);
    const float* d_rows = rows;
    if (!rows_on_device) {
        HIP_TRY(hipMemcpy(base + L.rows_off, rows, n_rows * (size_t)nfft * sizeof(float), hipMemcpyHostToDevice));
        d_rows = reinterpret_cast<const float*>(base + L.rows_off);
    }
    HIP_TRY(hipMemcpy(base + L.thr_off, thresholds, n_rows * sizeof(double), hipMemcpyHostToDevice));
    hipError_t e = sdrk::launch_row_peaks(d_rows, n_rows, nfft, reinterpret_cast<const double*>(base + L.thr_off),
                                          min_distance, max_peaks, reinterpret_cast<int*>(base + L.idx_off),
                                          reinterpret_cast<int*>(base + L.cnt_off), nullptr);
    if (e != hipSuccess) return fail(SDRK_ERR_HIP, "row_peaks launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpy(out_idx, base + L.idx_off, n_rows * (size_t)max_peaks * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_count, base + L.cnt_off, n_rows * sizeof(int), hipMemcpyDeviceToHost));
    return SDRK_OK;
}

int sdrk_frame_features_device(sdrk_plan* p, const void* d_iq, size_t n_frames, size_t frame_stride,
                               float* d_out_db, int rank, float gamma, int min_distance, int max_peaks,
                               double* d_stats, double* d_thr, int32_t* d_idx, int32_t* d_count, void* stream) {
    if (!p) return fail(SDRK_ERR_INVALID, "plan is NULL");
    if (n_frames == 0) return SDRK_OK;
    if (!d_iq || !d_stats) return fail(SDRK_ERR_INVALID, "d_iq or d_stats is NULL");
    const bool peaks = d_idx != nullptr || d_count != nullptr;
    if (peaks && (!d_idx || !d_count || max_peaks < 1 || min_distance < 1))
        return fail(SDRK_ERR_INVALID, "peaks need d_idx, d_count and max_peaks, min_distance >= 1");
    if (frame_stride == 0 && n_frames > 1) return fail(SDRK_ERR_INVALID, "frame_stride must be >= 1");
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : p->stream;
    if (p->nfft == 4096) {
        // fused: the rows never leave the chip unless d_out_db asks for them (fft4096_features.hip)
        sdrk::LaunchArgs a;
        a.d_iq = d_iq; a.frame_stride = frame_stride; a.d_out = d_out_db; a.n_frames = n_frames; a.nfft = 4096;
        a.d_window = p->d_window; a.d_twiddle = p->d_twiddle; a.eps = p->eps; a.shift = p->shift;
        a.stream = s; a.num_cus = p->num_cus;
        hipError_t e = sdrk::launch_fft4096_features(a, rank, gamma, min_distance, max_peaks > 0 ? max_peaks : 1, d_stats,
                                                     d_thr, peaks ? d_idx : nullptr, peaks ? d_count : nullptr);
        if (e != hipSuccess) return fail(SDRK_ERR_HIP, "fused feature launch failed: %s", hipGetErrorString(e));
        return SDRK_OK;
    }
    // other frame lengths: the transform writes its rows (to the caller's buffer, or to plan staging in
    // chunks), then one single-read reduction launch per chunk
    const size_t nfft = (size_t)p->nfft;
    size_t per = n_frames;
    float* rows = d_out_db;
    if (!rows) {
        per = ((size_t)256 << 20) / (nfft * sizeof(float));
        if (per < 1) per = 1;
        if (per > n_frames) per = n_frames;
        int st = grow(p->device, &p->d_out, &p->out_cap, per * nfft * sizeof(float));
        if (st != SDRK_OK) return st;
        rows = static_cast<float*>(p->d_out);
    }
    for (size_t f0 = 0; f0 < n_frames; f0 += per) {
        const size_t nf = n_frames - f0 < per ? n_frames - f0 : per;
        float* dst = d_out_db ? d_out_db + f0 * nfft : rows;
        int st = plan_launch(p, static_cast<const float2*>(d_iq) + f0 * frame_stride, nf, frame_stride, dst,
                             sdrk::EPI_LOGPSD, s);
        if (st != SDRK_OK) return st;
        hipError_t e = sdrk::launch_row_features(dst, nf, p->nfft, rank, gamma, min_distance, max_peaks > 0 ? max_peaks : 1,
                                                 d_stats + f0 * 16, d_thr ? d_thr + f0 : nullptr,
                                                 peaks ? d_idx + f0 * (size_t)max_peaks : nullptr,
                                                 peaks ? d_count + f0 : nullptr, p->num_cus, s);
        if (e != hipSuccess) return fail(SDRK_ERR_HIP, "row_features launch failed: %s", hipGetErrorString(e));
    }
    return SDRK_OK;
}

namespace {
int frame_features_host_impl(sdrk_plan* p, const void* iq, size_t n_frames, size_t frame_stride, int rank,
                             float gamma, int min_distance, int max_peaks, double* out_stats, double* out_thr,
                             int32_t* out_idx, int32_t* out_count, float* out_db, const double* freqs, void* out_planes) {
    int st = check_exec_args(p, iq, n_frames, frame_stride, out_planes ? out_planes : static_cast<void*>(out_stats));
    if (st != SDRK_OK || n_frames == 0) return st;
    const bool peaks = out_idx != nullptr || out_count != nullptr;
    if (peaks && (!out_idx || (!out_count && !out_planes) || max_peaks < 1 || min_distance < 1))
        return fail(SDRK_ERR_INVALID, "peaks need out_idx, out_count and max_peaks, min_distance >= 1");
    HIP_TRY(hipSetDevice(p->device));
    const size_t nfft = (size_t)p->nfft;
    const size_t in_bytes = ((n_frames - 1) * frame_stride + nfft) * sizeof(float2);
    // results (and the rows, when the caller wants them or the frame length has no fused kernel) in a second
    // staging buffer that only grows
    const bool need_rows = out_db != nullptr;
    const FeatLayout L(n_frames, p->nfft, max_peaks, need_rows, peaks, out_planes != nullptr);
    void*& fbuf = p->d_feat;
    st = grow(p->device, &fbuf, &p->feat_cap, L.total);
    if (st != SDRK_OK) return st;
    char* base = static_cast<char*>(fbuf);
    float* d_rows = need_rows ? reinterpret_cast<float*>(base + L.rows_off) : nullptr;
    double* d_stats = reinterpret_cast<double*>(base + L.stats_off);
    double* d_thr = reinterpret_cast<double*>(base + L.thr_off);
    int32_t* d_idx = peaks ? reinterpret_cast<int32_t*>(base + L.idx_off) : nullptr;
    int32_t* d_cnt = peaks ? reinterpret_cast<int32_t*>(base + L.cnt_off) : nullptr;
    if (peaks) HIP_TRY(hipMemsetAsync(d_idx, 0xFF, n_frames * (size_t)max_peaks * sizeof(int), p->stream));        // unused slots: -1
    if (in_bytes <= 2 * HOST_CHUNK_BYTES) {
        st = grow(p->device, &p->d_in, &p->in_cap, in_bytes);
        if (st != SDRK_OK) return st;
        HIP_TRY(hipMemcpyAsync(p->d_in, iq, in_bytes, hipMemcpyHostToDevice, p->stream));
        st = sdrk_frame_features_device(p, p->d_in, n_frames, frame_stride, d_rows, rank, gamma, min_distance, max_peaks,
                                        d_stats, d_thr, d_idx, d_cnt, nullptr);
        if (st != SDRK_OK) return st;
    } else {
        // Large batches: the frames go through the pinned slots of the sdrk_exec_host pipeline in ~16 MiB chunks —
        // helper threads stage chunk c+1 (or the copy engine reads the caller's pinned array directly) while chunk
        // c crosses PCIe and chunk c-1 is measured.  The per-row results stay on the device until the end (they are
        // ~1 % of the input).
        if (!p->s_h2d) {
            HIP_TRY(hipStreamCreateWithFlags(&p->s_h2d, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&p->s_d2h, hipStreamNonBlocking));
        }
        const size_t stride_bytes = (frame_stride ? frame_stride : 1) * sizeof(float2);
        size_t per = HOST_CHUNK_BYTES / stride_bytes;
        if (per < 1) per = 1;
        const size_t chunk_in = ((per - 1) * frame_stride + nfft) * sizeof(float2);
        const bool in_pinned = pinned_ranges().covers(iq, in_bytes);
        sdrk::CopyPool& pool = sdrk::CopyPool::get();
        size_t c = 0;
        for (size_t f0 = 0; f0 < n_frames; f0 += per, ++c) {
            HostSlot& s = p->slot[c % HOST_SLOTS];
            const size_t nf = n_frames - f0 < per ? n_frames - f0 : per;
            const size_t cin = ((nf - 1) * frame_stride + nfft) * sizeof(float2);
            hipError_t e = hipSuccess;
            if (s.busy) {                                         // chunk c - HOST_SLOTS: measured, its staging is free
                e = hipEventSynchronize(s.ev_k);
                s.busy = false;
            }
            if (e == hipSuccess) {
                st = slot_reserve(p, s, chunk_in, 0);
                if (st != SDRK_OK) { slots_abandon(p); return st; }
                const void* src = static_cast<const float2*>(iq) + f0 * frame_stride;
                if (!in_pinned) {
                    pool.copy(s.h_in, src, cin);
                    src = s.h_in;
                }
                e = hipMemcpyAsync(s.d_in, src, cin, hipMemcpyHostToDevice, p->s_h2d);
            }
            if (e == hipSuccess) e = hipEventRecord(s.ev_in, p->s_h2d);
            if (e == hipSuccess) e = hipStreamWaitEvent(p->stream, s.ev_in, 0);
            if (e != hipSuccess) {
                slots_abandon(p);
                return fail(SDRK_ERR_HIP, "feature pipeline failed: %s", hipGetErrorString(e));
            }
            st = sdrk_frame_features_device(p, s.d_in, nf, frame_stride, d_rows ? d_rows + f0 * nfft : nullptr, rank, gamma,
                                            min_distance, max_peaks, d_stats + f0 * 16, d_thr + f0,
                                            d_idx ? d_idx + f0 * (size_t)max_peaks : nullptr, d_cnt ? d_cnt + f0 : nullptr, nullptr);
            if (st != SDRK_OK) { slots_abandon(p); return st; }
            e = hipEventRecord(s.ev_k, p->stream);
            if (e != hipSuccess) {
                slots_abandon(p);
                return fail(SDRK_ERR_HIP, "feature pipeline failed: %s", hipGetErrorString(e));
            }
            s.busy = true;
            s.user_out = nullptr;
        }
    }
    // the results come back through ONE exit: whatever fails from here on, no chunk of the pipelined form may still be
    // using its staging slot when the call returns (a later sdrk_exec_host would restage it under the copy engine)
    auto results = [&]() -> int {
        if (out_planes) {
            int r = planes_to_host(base, L, peaks, n_frames, p->nfft, gamma, max_peaks, freqs, out_planes, out_idx, p->stream);
            if (r != SDRK_OK) return r;
        } else {
            HIP_TRY(hipMemcpyAsync(out_stats, base + L.stats_off, n_frames * 16 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
            if (out_thr) HIP_TRY(hipMemcpyAsync(out_thr, base + L.thr_off, n_frames * sizeof(double), hipMemcpyDeviceToHost, p->stream));
            if (peaks) {
                HIP_TRY(hipMemcpyAsync(out_idx, base + L.idx_off, n_frames * (size_t)max_peaks * sizeof(int), hipMemcpyDeviceToHost, p->stream));
                HIP_TRY(hipMemcpyAsync(out_count, base + L.cnt_off, n_frames * sizeof(int), hipMemcpyDeviceToHost, p->stream));
            }
        }
        if (need_rows)
            HIP_TRY(hipMemcpyAsync(out_db, base + L.rows_off, n_frames * nfft * sizeof(float), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        return SDRK_OK;
    };
    st = results();
    if (st != SDRK_OK) {
        if (p->s_h2d) slots_abandon(p);
        else (void)hipStreamSynchronize(p->stream);
        for (auto& s : p->slot) s.busy = false;
        return st;
    }
    for (auto& s : p->slot) s.busy = false;                      // the pipelined form's chunks are all through
    return fused_check(p);
}
}  // namespace

int sdrk_frame_features_host(sdrk_plan* p, const void* iq, size_t n_frames, size_t frame_stride, int rank,
                             float gamma, int min_distance, int max_peaks, double* out_stats, double* out_thr,
                             int32_t* out_idx, int32_t* out_count, float* out_db) {
    if (p && n_frames && !out_stats) return fail(SDRK_ERR_INVALID, "input or output pointer is NULL");
    return frame_features_host_impl(p, iq, n_frames, frame_stride, rank, gamma, min_distance, max_peaks, out_stats, out_thr,
                                    out_idx, out_count, out_db, nullptr, nullptr);
}

int sdrk_frame_features_host_planes(sdrk_plan* p, const void* iq, size_t n_frames, size_t frame_stride, int rank,
                                    float gamma, int min_distance, int max_peaks, const double* freqs, void* out_planes,
                                    int32_t* out_idx, float* out_db) {
    if (p && n_frames && !out_planes) return fail(SDRK_ERR_INVALID, "input or output pointer is NULL");
    return frame_features_host_impl(p, iq, n_frames, frame_stride, rank, gamma, min_distance, max_peaks, nullptr, nullptr,
                                    out_idx, nullptr, out_db, freqs, out_planes);
}

/* ---- waterfall ring ------------------------------------------------------ */

int sdrk_waterfall_create(int device, int nfft, int maxlen, sdrk_waterfall** out) {
    if (!out) return fail(SDRK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (nfft < 1) return fail(SDRK_ERR_INVALID, "nfft must be >= 1");
    if (maxlen < 1) return fail(SDRK_ERR_INVALID, "maxlen must be >= 1");
    int st = check_device(device);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipSetDevice(device));
    sdrk_waterfall* wf = new (std::nothrow) sdrk_waterfall();
    if (!wf) return fail(SDRK_ERR_NOMEM, "out of host memory");
    wf->device = device;
    wf->nfft = nfft;
    wf->maxlen = maxlen;
    hipError_t e = hipMalloc((void**)&wf->d_ring, (size_t)maxlen * nfft * sizeof(float));
    if (e == hipSuccess && sdrk::fft_tiled2_has_mip(nfft, sdrk::EPI_LOGPSD)) {
        // 1/16 of the ring again: the rows max-hold-decimated by 16, written by the transform beside the rows (N >= 2^20)
        e = hipMalloc((void**)&wf->d_mip_ring, (size_t)maxlen * (nfft / 16) * sizeof(float));
        // -inf everywhere: should a slot ever be read before the transform has written it, a maximum over it is harmless
        if (e == hipSuccess) e = hipMemsetD32(wf->d_mip_ring, (int)0xFF800000u, (size_t)maxlen * (size_t)(nfft / 16));
        wf->mip_ok.assign((size_t)maxlen, 0);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&wf->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        int s = fail(e == hipErrorOutOfMemory ? SDRK_ERR_NOMEM : SDRK_ERR_HIP,
                     "waterfall allocation failed: %s", hipGetErrorString(e));
        sdrk_waterfall_destroy(wf);
        return s;
    }
    *out = wf;
    return SDRK_OK;
}

int sdrk_waterfall_destroy(sdrk_waterfall* wf) {
    if (!wf) return SDRK_OK;
    (void)hipSetDevice(wf->device);
    if (wf->s_copy) {
        (void)hipStreamSynchronize(wf->s_copy);
        (void)hipStreamDestroy(wf->s_copy);
    }
    if (wf->ev_dec) (void)hipEventDestroy(wf->ev_dec);
    for (int k = 0; k < 2; ++k) {
        if (wf->ev_dec_done[k]) (void)hipEventDestroy(wf->ev_dec_done[k]);
        if (wf->ev_copy_done[k]) (void)hipEventDestroy(wf->ev_copy_done[k]);
    }
    if (wf->stream) {
        (void)hipStreamSynchronize(wf->stream);
        (void)hipStreamDestroy(wf->stream);
    }
    if (wf->d_ring) (void)hipFree(wf->d_ring);
    if (wf->d_mip_ring) (void)hipFree(wf->d_mip_ring);
    for (int k = 0; k < 2; ++k)
        if (wf->d_dec[k]) (void)hipFree(wf->d_dec[k]);
    delete wf;
    return SDRK_OK;
}

int sdrk_waterfall_rows(const sdrk_waterfall* wf) {
    return wf ? (int)wf->count : fail(SDRK_ERR_INVALID, "waterfall is NULL");
}

int sdrk_waterfall_maxhold16_rows(const sdrk_waterfall* wf) {
    if (!wf) return fail(SDRK_ERR_INVALID, "waterfall is NULL");
    if (wf->mip_ok.empty()) return 0;
    int n = 0;
    const size_t L = (size_t)wf->maxlen, start = (wf->head + L - wf->count % L) % L;
    for (size_t r = 0; r < wf->count; ++r) n += wf->mip_ok[(start + r) % L] ? 1 : 0;
    return n;
}

int sdrk_waterfall_clear(sdrk_waterfall* wf) {
    if (!wf) return fail(SDRK_ERR_INVALID, "waterfall is NULL");
    wf->head = 0;
    wf->count = 0;
    return SDRK_OK;
}

// Before `run` ring slots from wf->head are overwritten on wf->stream: if the second stream's reduction may still be reading
// any of them, the write waits for it.
static hipError_t wf_before_write(sdrk_waterfall* wf, size_t run) {
    if (run == 0) return hipSuccess;
    const size_t L = (size_t)wf->maxlen;
    for (int k = 0; k < 2; ++k) {
        if (!wf->dec_guard[k]) continue;
        const size_t a0 = wf->head, b0 = wf->dec_start[k];    // both ranges may wrap: compare slot by modular distance
        const bool overlap = ((b0 + L - a0) % L) < run || ((a0 + L - b0) % L) < wf->dec_rows[k];
        if (!overlap) continue;
        wf->dec_guard[k] = false;                             // (a stream waits for an event once; later writes are behind it)
        const hipError_t e = hipStreamWaitEvent(wf->stream, wf->ev_dec_done[k], 0);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

static void wf_advance(sdrk_waterfall* wf, size_t rows) {
    wf->head = (wf->head + rows) % (size_t)wf->maxlen;
    wf->count = wf->count + rows > (size_t)wf->maxlen ? (size_t)wf->maxlen : wf->count + rows;
}

int sdrk_waterfall_append_rows(sdrk_waterfall* wf, const float* rows, size_t n_rows) {
    if (!wf) return fail(SDRK_ERR_INVALID, "waterfall is NULL");
    if (n_rows == 0) return SDRK_OK;
    if (!rows) return fail(SDRK_ERR_INVALID, "rows is NULL");
    HIP_TRY(hipSetDevice(wf->device));
    // deque(maxlen) semantics (dashboard/callbacks.py:19,176): only the newest maxlen survive.
    size_t skip = n_rows > (size_t)wf->maxlen ? n_rows - (size_t)wf->maxlen : 0;
    if (skip) wf_advance(wf, skip);
    const size_t row_bytes = (size_t)wf->nfft * sizeof(float);
    size_t done = skip;
    while (done < n_rows) {
        size_t run = (size_t)wf->maxlen - wf->head;
        if (run > n_rows - done) run = n_rows - done;
        HIP_TRY(wf_before_write(wf, run));
        HIP_TRY(hipMemcpyAsync(wf->d_ring + wf->head * (size_t)wf->nfft, rows + done * (size_t)wf->nfft,
                               run * row_bytes, hipMemcpyHostToDevice, wf->stream));
        if (!wf->mip_ok.empty()) std::fill(wf->mip_ok.begin() + (long)wf->head, wf->mip_ok.begin() + (long)(wf->head + run), 0);
        wf_advance(wf, run);
        done += run;
    }
    HIP_TRY(hipStreamSynchronize(wf->stream));
    return SDRK_OK;
}

int sdrk_waterfall_append_iq_device_async(sdrk_waterfall* wf, sdrk_plan* p, const void* d_iq,
                                          size_t n_frames, size_t frame_stride) {
    if (!wf || !p) return fail(SDRK_ERR_INVALID, "waterfall or plan is NULL");
    if (p->nfft != wf->nfft || p->device != wf->device)
        return fail(SDRK_ERR_INVALID, "plan (nfft %d, device %d) does not match waterfall (nfft %d, device %d)",
                    p->nfft, p->device, wf->nfft, wf->device);
    if (n_frames == 0) return SDRK_OK;
    if (!d_iq) return fail(SDRK_ERR_INVALID, "d_iq is NULL");
    if (frame_stride == 0 && n_frames > 1) return fail(SDRK_ERR_INVALID, "frame_stride must be >= 1 for more than one frame");
    HIP_TRY(hipSetDevice(wf->device));
    size_t skip = n_frames > (size_t)wf->maxlen ? n_frames - (size_t)wf->maxlen : 0;
    if (skip) wf_advance(wf, skip);
    size_t done = skip;
    while (done < n_frames) {
        size_t run = (size_t)wf->maxlen - wf->head;
        if (run > n_frames - done) run = n_frames - done;
        // the transform writes its rows straight into the ring slots
        // ... and, where the row pass can, the by-16 max-hold of each row beside it.  Whether it did is reported by the
        // launcher itself (with_mip); a plan that cannot — chirp-z, N = 65536, the pair-kernel builds — leaves the slots marked
        // as having none, and max-mode read-outs of those slots reduce the rows themselves.
        bool with_mip = false;
        HIP_TRY(wf_before_write(wf, run));
        int st = plan_launch(p, static_cast<const float2*>(d_iq) + done * frame_stride, run, frame_stride,
                             wf->d_ring + wf->head * (size_t)wf->nfft, sdrk::EPI_LOGPSD, wf->stream,
                             wf->d_mip_ring ? wf->d_mip_ring + wf->head * (size_t)(wf->nfft / 16) : nullptr, &with_mip);
        if (st != SDRK_OK) return st;
        if (!wf->mip_ok.empty()) std::fill(wf->mip_ok.begin() + (long)wf->head, wf->mip_ok.begin() + (long)(wf->head + run), with_mip ? 1 : 0);
        wf_advance(wf, run);
        done += run;
    }
    return SDRK_OK;
}

int sdrk_waterfall_sync(sdrk_waterfall* wf, sdrk_plan* p) {
    if (!wf) return fail(SDRK_ERR_INVALID, "waterfall is NULL");
    HIP_TRY(hipSetDevice(wf->device));
    HIP_TRY(hipStreamSynchronize(wf->stream));
    return p ? fused_check(p) : SDRK_OK;
}

int sdrk_waterfall_append_iq_device(sdrk_waterfall* wf, sdrk_plan* p, const void* d_iq,
                                    size_t n_frames, size_t frame_stride) {
    int st = sdrk_waterfall_append_iq_device_async(wf, p, d_iq, n_frames, frame_stride);
    if (st != SDRK_OK || n_frames == 0) return st;
    return sdrk_waterfall_sync(wf, p);
}

int sdrk_waterfall_append_iq(sdrk_waterfall* wf, sdrk_plan* p, const void* iq, size_t n_frames,
                             size_t frame_stride) {
    if (!wf || !p) return fail(SDRK_ERR_INVALID, "waterfall or plan is NULL");
    if (n_frames == 0) return SDRK_OK;
    if (!iq) return fail(SDRK_ERR_INVALID, "iq is NULL");
    if (frame_stride == 0 && n_frames > 1) return fail(SDRK_ERR_INVALID, "frame_stride must be >= 1");
    if (n_frames > p->max_batch)
        return fail(SDRK_ERR_INVALID, "n_frames %zu exceeds the plan's max_batch %zu", n_frames, p->max_batch);
    HIP_TRY(hipSetDevice(p->device));
    const size_t in_bytes = ((n_frames - 1) * frame_stride + (size_t)p->nfft) * sizeof(float2);
    int st = grow(p->device, &p->d_in, &p->in_cap, in_bytes);
    if (st != SDRK_OK) return st;
    HIP_TRY(hipMemcpyAsync(p->d_in, iq, in_bytes, hipMemcpyHostToDevice, wf->stream));
    return sdrk_waterfall_append_iq_device(wf, p, p->d_in, n_frames, frame_stride);
}

int sdrk_waterfall_read(sdrk_waterfall* wf, float* out, size_t max_rows, size_t* n_rows) {
    if (!wf || !n_rows) return fail(SDRK_ERR_INVALID, "waterfall or n_rows is NULL");
    *n_rows = 0;
    size_t rows = wf->count < max_rows ? wf->count : max_rows;
    if (rows == 0) return SDRK_OK;
    if (!out) return fail(SDRK_ERR_INVALID, "out is NULL");
    HIP_TRY(hipSetDevice(wf->device));
    const size_t L = (size_t)wf->maxlen, nf = (size_t)wf->nfft;
    // newest row is at head-1; the `rows` newest start at head-rows (mod L)
    size_t start = (wf->head + L - rows % L) % L;
    size_t first = L - start < rows ? L - start : rows;
    HIP_TRY(hipMemcpyAsync(out, wf->d_ring + start * nf, first * nf * sizeof(float),
                           hipMemcpyDeviceToHost, wf->stream));
    if (first < rows)
        HIP_TRY(hipMemcpyAsync(out + first * nf, wf->d_ring, (rows - first) * nf * sizeof(float),
                               hipMemcpyDeviceToHost, wf->stream));
    HIP_TRY(hipStreamSynchronize(wf->stream));
    *n_rows = rows;
    return SDRK_OK;
}

// the reduction of `rows` ring rows starting at slot `start` to nfft / factor bins each, into wf->d_dec[slot]: from the by-16
// rows when every requested slot has one (max mode, factor a multiple of 16) — 1/16 of the bytes —, else from the rows
static hipError_t wf_launch_decimate(sdrk_waterfall* wf, size_t start, size_t rows, int factor, int mode, hipStream_t stream,
                                     int slot) {
    bool mip = wf->d_mip_ring && mode == 0 && factor % 16 == 0;
    for (size_t r = 0; r < rows && mip; ++r) mip = wf->mip_ok[(start + r) % (size_t)wf->maxlen] != 0;
    if (mip)
        return sdrk::launch_decimate_mip(wf->d_mip_ring, wf->nfft, wf->maxlen, (int)start, (int)rows, factor,
                                         static_cast<float*>(wf->d_dec[slot]), stream);
    return sdrk::launch_decimate_rows(wf->d_ring, wf->nfft, wf->maxlen, (int)start, (int)rows, factor, mode,
                                      static_cast<float*>(wf->d_dec[slot]), stream);
}

int sdrk_waterfall_read_decimated(sdrk_waterfall* wf, float* out, size_t max_rows, int factor, int mode,
                                  size_t* n_rows) {
    if (!wf || !n_rows) return fail(SDRK_ERR_INVALID, "waterfall or n_rows is NULL");
    *n_rows = 0;
    if (factor < 1 || wf->nfft % factor != 0) return fail(SDRK_ERR_INVALID, "factor %d must divide nfft %d", factor, wf->nfft);
    if (mode != 0 && mode != 1) return fail(SDRK_ERR_INVALID, "mode must be 0 (max) or 1 (mean)");
    if (wf->reads_in_flight) return fail(SDRK_ERR_INVALID, "a two-phase decimated read is in flight (call _end first)");
    size_t rows = wf->count < max_rows ? wf->count : max_rows;
    if (rows == 0) return SDRK_OK;
    if (!out) return fail(SDRK_ERR_INVALID, "out is NULL");
    HIP_TRY(hipSetDevice(wf->device));
    const size_t L = (size_t)wf->maxlen;
    const size_t start = (wf->head + L - rows % L) % L;
    const size_t bins = (size_t)(wf->nfft / factor);
    int st = grow(wf->device, &wf->d_dec[0], &wf->dec_cap[0], rows * bins * sizeof(float));
    if (st != SDRK_OK) return st;
    hipError_t e = wf_launch_decimate(wf, start, rows, factor, mode, wf->stream, 0);
    if (e != hipSuccess) return fail(SDRK_ERR_HIP, "decimate launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(out, wf->d_dec[0], rows * bins * sizeof(float), hipMemcpyDeviceToHost, wf->stream));
    HIP_TRY(hipStreamSynchronize(wf->stream));
    *n_rows = rows;
    return SDRK_OK;
}

int sdrk_waterfall_read_decimated_begin(sdrk_waterfall* wf, float* out, size_t max_rows, int factor, int mode,
                                        size_t* n_rows) {
    if (!wf || !n_rows) return fail(SDRK_ERR_INVALID, "waterfall or n_rows is NULL");
    *n_rows = 0;
    if (wf->reads_in_flight >= 2) return fail(SDRK_ERR_INVALID, "two decimated reads are already in flight (call _end first)");
    if (factor < 1 || wf->nfft % factor != 0) return fail(SDRK_ERR_INVALID, "factor %d must divide nfft %d", factor, wf->nfft);
    if (mode != 0 && mode != 1) return fail(SDRK_ERR_INVALID, "mode must be 0 (max) or 1 (mean)");
    size_t rows = wf->count < max_rows ? wf->count : max_rows;
    if (rows == 0) return SDRK_OK;
    if (!out) return fail(SDRK_ERR_INVALID, "out is NULL");
    HIP_TRY(hipSetDevice(wf->device));
    if (!wf->s_copy) {
        HIP_TRY(hipStreamCreateWithFlags(&wf->s_copy, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&wf->ev_dec, hipEventDisableTiming));
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(hipEventCreateWithFlags(&wf->ev_dec_done[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&wf->ev_copy_done[k], hipEventDisableTiming));
        }
    }
    const int k = (wf->oldest_read + wf->reads_in_flight) & 1;
    const size_t L = (size_t)wf->maxlen;
    const size_t start = (wf->head + L - rows % L) % L;
    const size_t bins = (size_t)(wf->nfft / factor);
    if (rows * bins * sizeof(float) > wf->dec_cap[k]) HIP_TRY(hipStreamSynchronize(wf->s_copy));   // the staging is about to move
    int st = grow(wf->device, &wf->d_dec[k], &wf->dec_cap[k], rows * bins * sizeof(float));
    if (st != SDRK_OK) return st;
    // the rows are complete once everything enqueued on the transform stream so far has run; from there on the second stream
    HIP_TRY(hipEventRecord(wf->ev_dec, wf->stream));
    HIP_TRY(hipStreamWaitEvent(wf->s_copy, wf->ev_dec, 0));
    hipError_t e = wf_launch_decimate(wf, start, rows, factor, mode, wf->s_copy, k);
    if (e != hipSuccess) return fail(SDRK_ERR_HIP, "decimate launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipEventRecord(wf->ev_dec_done[k], wf->s_copy));
    wf->dec_start[k] = start;
    wf->dec_rows[k] = rows;
    wf->dec_guard[k] = true;
    HIP_TRY(hipMemcpyAsync(out, wf->d_dec[k], rows * bins * sizeof(float), hipMemcpyDeviceToHost, wf->s_copy));
    HIP_TRY(hipEventRecord(wf->ev_copy_done[k], wf->s_copy));
    ++wf->reads_in_flight;
    *n_rows = rows;
    return SDRK_OK;
}

// Waits for the OLDEST read in flight (its rows are then in the caller's array); no-op when none is.
int sdrk_waterfall_read_decimated_end(sdrk_waterfall* wf) {
    if (!wf) return fail(SDRK_ERR_INVALID, "waterfall is NULL");
    if (!wf->reads_in_flight) return SDRK_OK;
    HIP_TRY(hipSetDevice(wf->device));
    const int k = wf->oldest_read;
    wf->oldest_read ^= 1;
    --wf->reads_in_flight;
    wf->dec_guard[k] = false;                                  // (its copy is behind its reduction on the same stream)
    HIP_TRY(hipEventSynchronize(wf->ev_copy_done[k]));
    return SDRK_OK;
}

}  // extern "C"
