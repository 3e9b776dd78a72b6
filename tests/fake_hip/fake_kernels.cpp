// tests/fake_hip/fake_kernels.cpp — stand-ins for the gfx950 kernels behind csrc/kernels.h, for the host-only sanitizer
// build of csrc/sdrk_api.hip (see hip/hip_runtime.h beside this file).  Each "launch" enqueues a host function on the
// stream it was given, so it runs asynchronously on that stream's thread like a kernel would.  The "transform" is NOT a
// spectrum: row[f][k] = 3 re(x[f][k]) - im(x[f][k]) + k (log epilogue) or (re + 1, im - 1) (complex epilogue) — a function of
// the input that lets the driver check, element by element, that the host pipeline moved the right bytes to the right place.
#include "../../sdr-iq-visualizer_amd/csrc/kernels.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <vector>

namespace sdrk {

static hipError_t fake_transform(const LaunchArgs& a) {
    LaunchArgs c = a;
    fakehip::of(a.stream).push([c] {
        const float2* x = static_cast<const float2*>(c.d_iq);
        for (size_t f = 0; f < c.n_frames; ++f)
            for (int k = 0; k < c.nfft; ++k) {
                const float2 v = x[f * c.frame_stride + (size_t)k];
                if (c.epilogue == EPI_LOGPSD) static_cast<float*>(c.d_out)[f * (size_t)c.nfft + k] = 3.0f * v.x - v.y + (float)(k & 1023);
                else if (c.epilogue == EPI_BLU_POST_LOG) { if (k < c.epi_n_out) static_cast<float*>(c.d_out)[f * (size_t)c.epi_n_out + k] = v.x; }
                else if (c.epilogue == EPI_BLU_POST_C64) { if (k < c.epi_n_out) static_cast<float2*>(c.d_out)[f * (size_t)c.epi_n_out + k] = v; }
                else static_cast<float2*>(c.d_out)[f * (size_t)c.nfft + k] = make_float2(v.x + 1.0f, v.y - 1.0f);   // EPI_COMPLEX, EPI_BLU_MUL
            }
        if (c.d_mip && fft_tiled2_has_mip(c.nfft, c.epilogue)) {                 // by-16 max-hold, [band][km] as the row pass writes it
            const int M = 2048, A = c.nfft / M, bands = A / 16;
            for (size_t f = 0; f < c.n_frames; ++f)
                for (int band = 0; band < bands; ++band)
                    for (int km = 0; km < M; ++km) {
                        const float* r = static_cast<const float*>(c.d_out) + f * (size_t)c.nfft + (size_t)km * A + 16 * band;
                        c.d_mip[f * (size_t)(c.nfft / 16) + (size_t)band * M + km] = *std::max_element(r, r + 16);
                    }
        }
    });
    return hipSuccess;
}

hipError_t launch_fft4096(const LaunchArgs& a) { return fake_transform(a); }
hipError_t launch_fft_small(const LaunchArgs& a) { return fake_transform(a); }
hipError_t launch_fft_lds(const LaunchArgs& a) { return fake_transform(a); }
hipError_t launch_fft_tiled2(const LaunchArgs& a) {
    if (a.mip_written) *a.mip_written = a.d_mip && fft_tiled2_has_mip(a.nfft, a.epilogue);   // as the real launcher reports it
    return fake_transform(a);
}
bool fft_lds_supports(int nfft) { return nfft >= 16 && nfft <= 16384 && nfft != 4096 && (nfft & (nfft - 1)) == 0; }
bool fft_tiled2_split(int nfft, int* la, int* lm) {
    int lg = 0;
    while ((1 << lg) < nfft) ++lg;
    if ((1 << lg) != nfft || lg < 15 || lg > 22) return false;
    *la = lg == 20 ? 9 : (lg / 2 < 7 ? 7 : lg / 2);
    *lm = lg - *la;
    return true;
}
bool fft_tiled2_has_mip(int nfft, int epilogue) { return epilogue == EPI_LOGPSD && nfft >= (1 << 20) && nfft <= (1 << 22); }

hipError_t launch_synth_fill(uint32_t seed, uint64_t first_frame, size_t n_frames, int nfft, void* d_iq, hipStream_t s) {
    fakehip::of(s).push([=] {
        float2* o = static_cast<float2*>(d_iq);
        for (size_t i = 0; i < n_frames * (size_t)nfft; ++i) o[i] = make_float2((float)((seed + first_frame + i) & 0xFFF) - 2048.0f, (float)(i & 0xFFF) - 2048.0f);
    });
    return hipSuccess;
}

size_t fused64k_ring_bytes() { return 4096; }
size_t fused64k_ctrl_words() { return 64; }
unsigned fused64k_sets(int) { return 1; }
std::atomic<int> g_fake_fused_fail{0};   // the driver sets it: the next persistent launch reports a failed hand-over
std::atomic<int> g_fake_fused_launches{0};
hipError_t launch_fused64k(const LaunchArgs& a, void*, unsigned* d_ctrl) {
    hipError_t e = fake_transform(a);
    const unsigned err = g_fake_fused_fail.exchange(0) ? 1u : 0u;
    ++g_fake_fused_launches;
    fakehip::of(a.stream).push([d_ctrl, err] { memset(d_ctrl, 0, 64 * sizeof(unsigned)); d_ctrl[0] = 1; d_ctrl[1] = err; });   // "1 set formed"
    return e;
}
bool blu_fused_supports(int) { return false; }
hipError_t launch_blu_fused(const void*, size_t, size_t, int, int, const float*, const void*, const void*, const void*, float,
                            int, int, void*, int, hipStream_t) { return hipErrorUnknown; }
hipError_t launch_blu_pre(const void* d_iq, size_t stride, size_t nf, int N, int M, const float*, const void*, void* d_a, int, hipStream_t s, bool) {
    fakehip::of(s).push([=] {
        for (size_t f = 0; f < nf; ++f) {
            float2* o = static_cast<float2*>(d_a) + f * (size_t)M;
            memset(o, 0, (size_t)M * sizeof(float2));
            memcpy(o, static_cast<const float2*>(d_iq) + f * stride, (size_t)N * sizeof(float2));
        }
    });
    return hipSuccess;
}
hipError_t launch_blu_mul(const void* d_A, const void*, size_t nf, int M, void* d_out, int, hipStream_t s) {
    fakehip::of(s).push([=] { memcpy(d_out, d_A, nf * (size_t)M * sizeof(float2)); });
    return hipSuccess;
}
hipError_t launch_blu_post(const void* d_Y, const void*, size_t nf, int N, int M, float, int, int epilogue, void* d_out, int, hipStream_t s) {
    fakehip::of(s).push([=] {
        for (size_t f = 0; f < nf; ++f)
            for (int k = 0; k < N; ++k) {
                const float2 v = static_cast<const float2*>(d_Y)[f * (size_t)M + k];
                if (epilogue == EPI_LOGPSD) static_cast<float*>(d_out)[f * (size_t)N + k] = v.x;
                else static_cast<float2*>(d_out)[f * (size_t)N + k] = v;
            }
    });
    return hipSuccess;
}

// per-row results: stats[r][0] = max of the row, stats[r][15] = nfft, everything else zero; no peaks
static void fake_stats(const float* rows, size_t n_rows, int nfft, double* st, double* thr, int* cnt) {
    for (size_t r = 0; r < n_rows; ++r) {
        std::fill(st + r * 16, st + r * 16 + 16, 0.0);
        st[r * 16] = *std::max_element(rows + r * (size_t)nfft, rows + (r + 1) * (size_t)nfft);
        st[r * 16 + 15] = nfft;
        if (thr) thr[r] = st[r * 16] - 1.0;
        if (cnt) cnt[r] = 0;
    }
}
hipError_t launch_row_features(const float* d_rows, size_t n_rows, int nfft, int, float, int, int, double* d_stats, double* d_thr,
                               int*, int* d_cnt, int, hipStream_t s) {
    fakehip::of(s).push([=] { fake_stats(d_rows, n_rows, nfft, d_stats, d_thr, d_cnt); });
    return hipSuccess;
}
hipError_t launch_feature_finalize(const double* d_stats, const double*, const int*, const int*, size_t n_rows, int, float, int,
                                   const double*, double* d_out, hipStream_t s) {
    fakehip::of(s).push([=] {
        for (size_t p = 0; p < 19; ++p)
            for (size_t r = 0; r < n_rows; ++r) d_out[p * n_rows + r] = p == 0 ? d_stats[r * 16] : 0.0;
    });
    return hipSuccess;
}
hipError_t launch_fft4096_features(const LaunchArgs& a, int, float, int, int, double* d_stats, double* d_thr, int*, int* d_cnt) {
    // rows are optional here: transform into a private buffer when the caller wants none
    LaunchArgs c = a;
    fakehip::of(a.stream).push([c, d_stats, d_thr, d_cnt] {
        std::vector<float> row((size_t)c.nfft);
        const float2* x = static_cast<const float2*>(c.d_iq);
        for (size_t f = 0; f < c.n_frames; ++f) {
            for (int k = 0; k < c.nfft; ++k) {
                const float2 v = x[f * c.frame_stride + (size_t)k];
                row[(size_t)k] = 3.0f * v.x - v.y + (float)(k & 1023);
            }
            if (c.d_out) memcpy(static_cast<float*>(c.d_out) + f * (size_t)c.nfft, row.data(), row.size() * sizeof(float));
            fake_stats(row.data(), 1, c.nfft, d_stats + f * 16, d_thr ? d_thr + f : nullptr, d_cnt ? d_cnt + f : nullptr);
        }
    });
    return hipSuccess;
}
hipError_t launch_row_peaks(const float*, size_t n_rows, int, const double*, int, int, int*, int* d_count, hipStream_t s) {
    fakehip::of(s).push([=] { std::fill(d_count, d_count + n_rows, 0); });
    return hipSuccess;
}
hipError_t launch_decimate_rows(const float* d_ring, int nfft, int maxlen, int start, int n_rows, int factor, int mode, float* d_out,
                                hipStream_t s) {
    fakehip::of(s).push([=] {
        const int bins = nfft / factor;
        for (int r = 0; r < n_rows; ++r)
            for (int b = 0; b < bins; ++b) {
                const float* src = d_ring + (size_t)((start + r) % maxlen) * nfft + (size_t)b * factor;
                float acc = mode == 0 ? -INFINITY : 0.0f;
                for (int i = 0; i < factor; ++i) acc = mode == 0 ? std::max(acc, src[i]) : acc + src[i];
                d_out[(size_t)r * bins + b] = mode == 0 ? acc : acc / (float)factor;
            }
    });
    return hipSuccess;
}
hipError_t launch_decimate_mip(const float* d_mip, int nfft, int maxlen, int start, int n_rows, int factor, float* d_out, hipStream_t s) {
    fakehip::of(s).push([=] {
        const int n16 = nfft / 16, G = factor / 16, bands = n16 / 2048, M = 2048, bins = n16 / G;
        for (int r = 0; r < n_rows; ++r)
            for (int b = 0; b < bins; ++b) {
                const float* src = d_mip + (size_t)((start + r) % maxlen) * n16;
                float acc = -INFINITY;
                for (int i = 0; i < G; ++i) {
                    const int j = b * G + i, km = j / bands, band = j - km * bands;
                    acc = std::max(acc, src[(size_t)band * M + km]);
                }
                d_out[(size_t)r * bins + b] = acc;
            }
    });
    return hipSuccess;
}
hipError_t launch_stream_mix(const void*, void*, size_t, int, hipStream_t s) { fakehip::of(s).push([] {}); return hipSuccess; }
hipError_t launch_copy_1to1(const void* d_in, void* d_out, size_t bytes, int, int, hipStream_t s) {
    fakehip::of(s).push([=] { memcpy(d_out, d_in, bytes); });
    return hipSuccess;
}
hipError_t launch_power_mean(const void* d_spec, size_t n_frames, int nfft, float scale, float* d_out, hipStream_t s) {
    fakehip::of(s).push([=] {
        for (int k = 0; k < nfft; ++k) {
            float acc = 0.0f;
            for (size_t f = 0; f < n_frames; ++f) {
                const float2 z = static_cast<const float2*>(d_spec)[f * (size_t)nfft + k];
                acc += z.x * z.x + z.y * z.y;
            }
            d_out[k] = acc * scale;
        }
    });
    return hipSuccess;
}

}  // namespace sdrk
