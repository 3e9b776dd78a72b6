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

hipError_t launch_power_mean(const void* d_spec, size_t n_frames, int nfft, float scale, float* d_out,
                             hipStream_t stream) {
    hipLaunchKernelGGL(power_mean_kernel, dim3((nfft + 255) / 256), dim3(256), 0, stream,
                       static_cast<const float2*>(d_spec), n_frames, nfft, scale, d_out);
    return hipGetLastError();
}

}  // namespace sdrk
