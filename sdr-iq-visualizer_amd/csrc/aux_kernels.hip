// aux_kernels.hip — the synthetic IQ generator used by the bench harness and the
// full-size parity tests (declared in include/sdrk.h, sdrk_synth_fill).
//
// Values imitate what the reference reads at app/sdr/streamer.py:114
// (self.sdr.rx() on a PlutoSDR: 12-bit ADC codes on I and Q), generated from an
// integer hash so that the numpy mirror in the host package (synth.py) produces
// the same float32 bits with no device round trip.
#include "kernels.h"

namespace sdrk {

__host__ __device__ inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// One thread -> two consecutive samples (one 16-byte store).
__global__ __launch_bounds__(256) void synth_fill_kernel(uint32_t seed, uint64_t first_frame,
                                                         size_t n_frames, int nfft,
                                                         float4* __restrict__ out) {
    const size_t pairs_per_frame = (size_t)nfft / 2;
    const size_t total = n_frames * pairs_per_frame;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / pairs_per_frame;
        const uint32_t n = (uint32_t)(i - f * pairs_per_frame) * 2u;
        const uint64_t F = first_frame + f;
        const uint32_t base = fmix32(seed ^ (uint32_t)F) ^ fmix32((uint32_t)(F >> 32) + 0x9E3779B1u);
        const uint32_t h0 = fmix32(base ^ n), h1 = fmix32(base ^ (n + 1u));
        float4 v;
        v.x = (float)((int)(h0 & 0xFFFu) - 2048);
        v.y = (float)((int)((h0 >> 12) & 0xFFFu) - 2048);
        v.z = (float)((int)(h1 & 0xFFFu) - 2048);
        v.w = (float)((int)((h1 >> 12) & 0xFFFu) - 2048);
        out[i] = v;
    }
}

hipError_t launch_synth_fill(uint32_t seed, uint64_t first_frame, size_t n_frames, int nfft,
                             void* d_iq, hipStream_t stream) {
    if (n_frames == 0) return hipSuccess;
    const size_t total = n_frames * (size_t)(nfft / 2);
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(synth_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, seed,
                       first_frame, n_frames, nfft, static_cast<float4*>(d_iq));
    return hipGetLastError();
}

// Welch-style averaged periodogram: out[k] = scale * sum_f |X[f][k]|^2 over the
// complex spectra of n_frames frames (column sums; thread k walks down its column,
// consecutive threads read consecutive bins).  Used by the offline PSD of the SigMF
// CLI (reference: plt.psd at scripts/process_sigmf_data.py:188-189 = matplotlib
// mlab.psd: mean of |FFT(w x)|^2 over segments / (Fs * sum w^2)).
__global__ __launch_bounds__(256) void power_mean_kernel(const float2* __restrict__ X, size_t n_frames,
                                                         int nfft, float scale, float* __restrict__ out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nfft) return;
    float acc = 0.0f, comp = 0.0f;  // Kahan: thousands of segments of equal magnitude
    for (size_t f = 0; f < n_frames; ++f) {
        const float2 z = X[f * (size_t)nfft + k];
        const float p = fmaf(z.x, z.x, z.y * z.y) - comp;
        const float t = acc + p;
        comp = (t - acc) - p;
        acc = t;
    }
    out[k] = acc * scale;
}

// Display decimation of waterfall rows: out[r][b] = max (mode 0) or mean (mode 1) of the
// `factor` consecutive dB values in[r][b*factor .. (b+1)*factor).  One wave per output bin
// group: lanes stride the factor-long run (coalesced), then a wave reduction.  No reference
// counterpart (the reference draws full rows, app/dashboard/callbacks.py:182-190); it makes
// N = 2^20 rows drawable.  `row_of` maps output row r to its ring slot.
__global__ __launch_bounds__(256) void decimate_rows_kernel(const float* __restrict__ ring, int nfft, int maxlen,
                                                            int start_slot, int n_rows, int factor, int mode,
                                                            float* __restrict__ out) {
    const int bins = nfft / factor;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t total = (size_t)n_rows * bins;
    for (size_t item = (size_t)blockIdx.x * 4 + wave; item < total; item += (size_t)gridDim.x * 4) {
        const int r = (int)(item / bins), b = (int)(item - (size_t)r * bins);
        const float* __restrict__ src = ring + (size_t)((start_slot + r) % maxlen) * nfft + (size_t)b * factor;
        float acc = mode == 0 ? -INFINITY : 0.0f;
        for (int i = lane; i < factor; i += 64) {
            const float v = src[i];
            acc = mode == 0 ? fmaxf(acc, v) : acc + v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float other = __shfl_down(acc, o, 64);
            acc = mode == 0 ? fmaxf(acc, other) : acc + other;
        }
        if (lane == 0) out[item] = mode == 0 ? acc : acc / (float)factor;
    }
}

hipError_t launch_decimate_rows(const float* d_ring, int nfft, int maxlen, int start_slot, int n_rows, int factor,
                                int mode, float* d_out, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    size_t total = (size_t)n_rows * (nfft / factor);
    size_t blocks = (total + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(decimate_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_ring, nfft, maxlen,
                       start_slot, n_rows, factor, mode, d_out);
    return hipGetLastError();
}

// Max-hold read-out from the by-16 rows that the M = 2048 row pass writes beside the ring (fft_tiled2.hip,
// row_pass_wave_kernel<…, MIP>): mip[slot][band][km] = max of output bins [16 (bands km + band), + 16), bands = A / 16 =
// nfft / 16 / 2048.  out[r][b] = max over by-16 index j in [b G, (b + 1) G), G = factor / 16, j = bands km + band.  One
// thread per output bin; neighbouring lanes read neighbouring km of the same band.  Max is exact, so the result is
// bit-identical to decimate_rows_kernel's over the full rows.
__global__ __launch_bounds__(256) void decimate_mip_kernel(const float* __restrict__ mip, int n16, int bands, int maxlen,
                                                           int start_slot, int n_rows, int G, float* __restrict__ out) {
    const int bins = n16 / G, M = n16 / bands;
    const size_t total = (size_t)n_rows * bins;
    for (size_t item = (size_t)blockIdx.x * 256 + threadIdx.x; item < total; item += (size_t)gridDim.x * 256) {
        const int r = (int)(item / bins), b = (int)(item - (size_t)r * bins);
        const float* __restrict__ src = mip + (size_t)((start_slot + r) % maxlen) * n16;
        float acc = -INFINITY;
        for (int i = 0; i < G; ++i) {
            const int j = b * G + i, km = j / bands, band = j - km * bands;
            acc = fmaxf(acc, src[(size_t)band * M + km]);
        }
        out[item] = acc;
    }
}

hipError_t launch_decimate_mip(const float* d_mip_ring, int nfft, int maxlen, int start_slot, int n_rows, int factor,
                               float* d_out, hipStream_t stream) {
    if (n_rows == 0) return hipSuccess;
    const int n16 = nfft / 16, G = factor / 16, bands = n16 / 2048;
    size_t blocks = ((size_t)n_rows * (n16 / G) + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(decimate_mip_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_mip_ring, n16, bands, maxlen,
                       start_slot, n_rows, G, d_out);
    return hipGetLastError();
}

// Streaming-copy ceiling with the spectrum path's traffic shape and nothing else: per 4096-sample
// frame 32 KiB read (8 x 16 B per thread), 16 KiB written (4 x 16 B per thread), no arithmetic beyond one
// add per output vector, non-temporal both ways, frames interleaved over a persistent grid of
// 3 workgroups per CU (the flagship kernel's launch shape).  bench.py quotes the flagship's achieved GB/s
// as a fraction of what THIS kernel reaches in the same process on the same buffers
// ("report both nominal and measured-copy fractions").
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_mix_kernel(const v4f* __restrict__ in, v4f* __restrict__ out,
                                                         size_t n_frames) {
    for (size_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const v4f* x = in + f * 2048;
        v4f* o = out + f * 1024;
        v4f v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(&x[threadIdx.x + 256 * j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(v[2 * j] + v[2 * j + 1], &o[threadIdx.x + 256 * j]);
    }
}

hipError_t launch_stream_mix(const void* d_in, void* d_out, size_t n_frames4096, int num_cus, hipStream_t stream) {
    if (n_frames4096 == 0) return hipSuccess;
    const size_t cap = (size_t)num_cus * 3;
    hipLaunchKernelGGL(stream_mix_kernel, dim3((unsigned)(n_frames4096 < cap ? n_frames4096 : cap)), dim3(256), 0,
                       stream, static_cast<const v4f*>(d_in), static_cast<v4f*>(d_out), n_frames4096);
    return hipGetLastError();
}

// The guide's reference shape for "achievable HBM bandwidth": a 1:1 copy, 16 bytes per lane each way
// (MI355X_MICROARCH.md quotes 6.29 TB/s for it).  Grid-stride over a fixed grid of a few workgroups per CU, four
// independent 16-byte loads per thread in flight.  bench.py runs it on the timed run's own buffers so that the
// 2:1 probe above can be anchored to a number somebody else measured.  The grid size matters by ~6 % (round 1's
// streambench: 4 or 16 workgroups per CU 5.5 TB/s, 8 per CU 5.2 on 32 + 16 GiB), so the probe tries several.
__global__ __launch_bounds__(256) void copy_1to1_kernel(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n_vec) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n_vec; i += stride) {
        v4f v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i + 256 * j < n_vec) v[j] = __builtin_nontemporal_load(&in[i + 256 * j]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i + 256 * j < n_vec) __builtin_nontemporal_store(v[j], &out[i + 256 * j]);
    }
}

hipError_t launch_copy_1to1(const void* d_in, void* d_out, size_t bytes, int num_cus, int blocks_per_cu, hipStream_t stream) {
    const size_t n_vec = bytes / 16;
    if (n_vec == 0) return hipSuccess;
    size_t blocks = (n_vec + 1023) / 1024;
    if (blocks > (size_t)num_cus * blocks_per_cu) blocks = (size_t)num_cus * blocks_per_cu;
    hipLaunchKernelGGL(copy_1to1_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<const v4f*>(d_in),
                       static_cast<v4f*>(d_out), n_vec);
    return hipGetLastError();
}

hipError_t launch_power_mean(const void* d_spec, size_t n_frames, int nfft, float scale, float* d_out,
                             hipStream_t stream) {
    hipLaunchKernelGGL(power_mean_kernel, dim3((nfft + 255) / 256), dim3(256), 0, stream,
                       static_cast<const float2*>(d_spec), n_frames, nfft, scale, d_out);
    return hipGetLastError();
}

}  // namespace sdrk
