/*
 * sdrk.h — C ABI of the MI355X-native IQ spectrum path ("sdrk" = SDR kernels).
 *
 * This is the drop-in boundary of the build.  The reference (a pure-Python Dash
 * app) has no FFI for this path: the work is three inline numpy expressions in
 * its SDR reader thread and a deque in its dashboard callback.  Each entry point
 * below names the reference expression it replaces (paths relative to the
 * reference checkout):
 *
 *   app/sdr/streamer.py:119   fft_data = np.fft.fftshift(np.fft.fft(samples))
 *   app/sdr/streamer.py:121   power_db = 20 * np.log10(np.abs(fft_data) + 1e-12)
 *   app/dashboard/callbacks.py:19    waterfall_data = deque(maxlen=100)
 *   app/dashboard/callbacks.py:176   waterfall_data.append(power_db)
 *   app/dashboard/callbacks.py:182   waterfall_array = np.array(waterfall_data)
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - every function returns SDRK_OK (0) or a negative sdrk_status; nothing
 *     throws; sdrk_last_error() returns a thread-local message for the last
 *     failure on the calling thread;
 *   - the caller owns every host buffer; the library owns plans, rings, device
 *     scratch, pinned staging and streams, released by the matching _destroy;
 *   - a plan or a waterfall is used by one thread at a time; distinct handles
 *     (e.g. one per GPU) may be used concurrently from different threads;
 *   - there is NO CPU fallback behind any of these symbols: with no usable
 *     gfx950 device they fail with SDRK_ERR_NO_DEVICE.
 */
#ifndef SDRK_H
#define SDRK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDRK_VERSION 500 /* 0.5.0: every addition to this ABI bumps it; _ffi.py refuses a library of another version */

typedef enum sdrk_status {
    SDRK_OK = 0,
    SDRK_ERR_INVALID = -1,     /* bad argument (NULL, size, unsupported nfft …)   */
    SDRK_ERR_NO_DEVICE = -2,   /* no HIP device / device index out of range        */
    SDRK_ERR_HIP = -3,         /* a HIP runtime call failed; see sdrk_last_error() */
    SDRK_ERR_NOMEM = -4,       /* host or device allocation failed                 */
    SDRK_ERR_UNSUPPORTED = -5  /* valid request this build has no kernel for       */
} sdrk_status;

/* Window applied to each frame before the transform.  The reference applies
 * none (streamer.py:119): SDRK_WINDOW_RECT reproduces it.  SDRK_WINDOW_HANN is
 * numpy.hanning(nfft) (symmetric; w[0]=w[nfft-1]=0), the window matplotlib's
 * psd() uses in scripts/process_sigmf_data.py:188.  SDRK_WINDOW_CUSTOM takes
 * nfft float32 coefficients from the caller. */
typedef enum sdrk_window {
    SDRK_WINDOW_RECT = 0,
    SDRK_WINDOW_HANN = 1,
    SDRK_WINDOW_CUSTOM = 2
} sdrk_window;

typedef struct sdrk_plan sdrk_plan;           /* opaque */
typedef struct sdrk_waterfall sdrk_waterfall; /* opaque */

/* ---- library / device ---------------------------------------------------- */

int sdrk_version(void);
/* Thread-local, never NULL; valid until the next failing call on this thread. */
const char* sdrk_last_error(void);
/* Number of HIP devices visible to the process (0 if none). */
int sdrk_device_count(void);
/* Short device description ("gfx950 … 256 CUs …") into buf. */
int sdrk_device_info(int device, char* buf, size_t buf_len);

/* ---- device memory helpers (so callers need no other GPU runtime) -------- */

int sdrk_dev_alloc(int device, size_t bytes, void** d_ptr);
int sdrk_dev_free(int device, void* d_ptr);
/* Free and total device memory in bytes (what the runtime reports for `device` right now). */
int sdrk_dev_mem_info(int device, size_t* free_bytes, size_t* total_bytes);
/* A long-lived input/output pair for the device-resident path, with the output placed where the two streams
 * interfere least.  On MI355X the achieved rate of a kernel that streams one buffer in and another out (the
 * spectrum path: 8 B in, 4 B out per sample) has two or three discrete levels ~6 % apart that depend on WHICH
 * two allocations are paired — read-only and (with rare exceptions) write-only rates do not depend on the buffer, and the level is
 * stable for the life of the pair (experiments/probes/placeprobe.hip).  This call allocates the input,
 * then up to `candidates` outputs (earlier ones stay allocated meanwhile, so each lands elsewhere), times a
 * probe over each pairing (after a warm-up by time; candidate 0 is timed again at the end and counts with the better of
 * its two timings: sdrk_placement_report) and keeps the fastest.  The probe is `plan`'s own transform over the pair (packed
 * frames; the input need not be initialised) or, with plan = NULL, a no-arithmetic kernel with the 2:1 traffic
 * shape.  probe_ms (may be NULL): `candidates` floats, the median probe time of each candidate (0 = not tried);
 * chosen (may be NULL): index kept.  Pairs too small for the effect to show (< 2^13 frame-equivalents of 4096
 * samples) are allocated without probing.  Free both with sdrk_dev_free. */
int sdrk_dev_alloc_stream_pair(int device, size_t in_bytes, size_t out_bytes, int candidates,
                               sdrk_plan* plan, void** d_in, void** d_out, float* probe_ms, int* chosen);
int sdrk_memcpy_h2d(int device, void* d_dst, const void* h_src, size_t bytes);
int sdrk_memcpy_d2h(int device, void* h_dst, const void* d_src, size_t bytes);

/* ---- pinned host memory for the numpy boundary ---------------------------------
 * sdrk_exec_host / sdrk_exec_fft_host stage pageable caller arrays through pinned slots with a pool of copy
 * threads (host_pool.h) — about 100 GB/s per process however many GPUs it drives.  Arrays that live in PINNED
 * host memory skip that: their chunks are DMA'd straight from / to the caller's memory, no host copy and no
 * host thread involved, so the boundary scales with the number of GPUs ("host gather via per-GPU
 * D2H into slices of one pinned array").  Either let the library allocate (sdrk_host_alloc: page-locked, visible
 * to every device) or register memory of your own (sdrk_host_register: the range must stay mapped until
 * sdrk_host_unregister — a numpy array must outlive its registration).  exec_host recognises any range that lies
 * inside such an allocation; input and output are decided independently. */
int sdrk_host_alloc(size_t bytes, void** h_ptr);
int sdrk_host_free(void* h_ptr);
int sdrk_host_register(void* h_ptr, size_t bytes);
int sdrk_host_unregister(void* h_ptr);
/* 1 if [h_ptr, h_ptr + bytes) lies inside memory made known by the calls above, else 0. */
int sdrk_host_is_pinned(const void* h_ptr, size_t bytes);

/* ---- spectrum plan -------------------------------------------------------
 * Replaces streamer.py:119,121 for frames of nfft complex64 samples:
 *     out_db[k] = 20*log10( | fftshift( fft( w * x ) ) |[k] + eps )     (float32)
 * nfft: 2 <= nfft <= 2^22 (2^SDRK_MAX_LOG2_NFFT) for powers of two (direct kernels);
 *       any other length 2 <= nfft <= 2^21 goes through a chirp-z (Bluestein) convolution
 *       built from the power-of-two kernels — the reference transforms whatever
 *       len(samples) is (streamer.py:119).
 * max_batch: largest n_frames a single sdrk_exec_host() call will be given
 *            (sizes the plan's device staging; exec_device has no such limit).
 * window_kind / window: see sdrk_window; `window` is read only for CUSTOM.
 * eps: additive floor on |X| (reference: 1e-12; legacy script 1e-10; 0 allowed).
 * shift: non-zero = fftshift order (DC at index nfft/2), as the reference. */
#define SDRK_MAX_LOG2_NFFT 22
int sdrk_plan_create(int device, int nfft, size_t max_batch, int window_kind,
                     const float* window, float eps, int shift, sdrk_plan** out);
/* Same with option flags.  nfft = 65536 has two forms: ONE persistent launch whose per-frame intermediate stays in each
 * XCD's L2 (fft_fused64k.hip), and the two tiled launches through a scratch buffer (fft_tiled2.hip); bit-identical rows.
 * Without a flag a plan takes the persistent launch for calls of 512 frames or more (BASELINE config 3: 3.9 ms against
 * 5.0 ms) and the two launches below that (they are faster there: the persistent grid costs ~30 us to set up); if a persistent launch ever reports a failed hand-over (its
 * workgroups must all be resident at once — another process's kernels can prevent that), the call returns SDRK_ERR_HIP and
 * the plan takes the two launches from then on.  SDRK_PLAN_FUSED64K: the persistent launch for every call (A/B work, tests).
 * SDRK_PLAN_TILED64K: never the persistent launch (a device shared with other processes' long-running kernels). */
#define SDRK_PLAN_FUSED64K 0x1u
#define SDRK_PLAN_TILED64K 0x8u
/* SDRK_PLAN_OVERLAP_PASSES (power-of-two nfft >= 2^15): run the row pass of chunk i on a second stream beside the
 * col pass of chunk i + 1, each on its own half of the scratch and on its own share of the CUs.  Bit-identical
 * rows; kept for A/B work like the flag above — measured 20-35 % SLOWER than the serial two-launch form on
 * BASELINE configs 3 and 5 (profiles/r03/overlap_probe.log). */
#define SDRK_PLAN_OVERLAP_PASSES 0x2u
/* SDRK_PLAN_TUNE_STAGING: allocate the device staging of the numpy boundary (sdrk_exec_host's chunk slots) at plan
 * creation and place each slot's row buffer as sdrk_dev_alloc_stream_pair places a resident pair — the fastest of three
 * candidates under the plan's own transform over one chunk.  Plans whose max_batch needs no chunking are unaffected.
 * sdrk_plan_staging_probe returns the probe times (3 per slot; n = 0 when nothing was tuned).  Measured on MI355X:
 * no effect at the shipped chunk size (the 24 MiB pairs are cache resident and the call is PCIe-bound). */
#define SDRK_PLAN_TUNE_STAGING 0x4u
int sdrk_plan_create_ex(int device, int nfft, size_t max_batch, int window_kind,
                        const float* window, float eps, int shift, unsigned flags, sdrk_plan** out);
int sdrk_plan_staging_probe(const sdrk_plan* plan, float* probe_ms, int capacity, int* n);
/* nfft = 65536 plans: how many persistent (fused) launches the plan has made so far, and whether one of them reported a
 * failed hand-over so that the plan now takes the two tiled launches (see SDRK_PLAN_FUSED64K above).  Either pointer may be
 * NULL.  Other plans: 0 / 0. */
int sdrk_plan_fused_status(const sdrk_plan* plan, unsigned* launches, int* fallen_back);
/* Large-frame plans (nfft >= 2^15) keep their two-pass intermediate in a scratch buffer, and — like the resident
 * input / output pair, see sdrk_dev_alloc_stream_pair — their speed depends a few per cent on WHERE that buffer
 * landed relative to the data (N = 65536 STFT over the same buffers with six different scratch allocations:
 * 5.02 ... 5.34 ms, stable per allocation).  This call times the plan's own transform of (d_iq, n_frames,
 * frame_stride) -> d_out_db with the present scratch and with up to `candidates` - 1 freshly allocated ones,
 * keeps the fastest and frees the others.  probe_ms (or NULL) receives `candidates` times (0 = not tried);
 * chosen (or NULL) the index kept (0 = the original).  d_out_db is overwritten with the transform's result.
 * Plans without a scratch return at once.  Not to be called while the plan is in use by another thread.
 * The probe warms up by time first (>= 60 ms of the plan's own launches: an idle MI355X needs tens of milliseconds of load
 * to reach its sustained shader clock, and a probe that starts cold measures that ramp instead of the placement), times
 * candidate 0 AGAIN after the last candidate, and replaces the present scratch only by a candidate that beats both of its
 * timings by one per cent (sdrk_placement_report has the re-timed figure). */
int sdrk_plan_tune_scratch(sdrk_plan* plan, const void* d_iq_c64, size_t n_frames, size_t frame_stride_samples,
                           float* d_out_db, int candidates, float* probe_ms, int* chosen);
/* What the LAST placement probe on the calling thread did (sdrk_dev_alloc_stream_pair, sdrk_plan_tune_scratch): the wall
 * time and number of its warm-up launches, candidate 0 as first timed and as timed again after the last candidate, and the
 * kept candidate's time, all in milliseconds (any pointer may be NULL).  Returns the number of candidates it tried (0: no
 * probe has run on this thread, or the last one had nothing to place).  A first / re-timed pair that differs by more than
 * the candidates do says the probe measured drift, not placement. */
int sdrk_placement_report(float* warm_ms, int* warm_launches, float* first_ms, float* retimed_first_ms, float* chosen_ms);
int sdrk_plan_destroy(sdrk_plan* plan);
int sdrk_plan_nfft(const sdrk_plan* plan);
int sdrk_plan_device(const sdrk_plan* plan);

/* Host in / host out (the numpy boundary).  iq_c64: interleaved float32 I,Q,
 * frame f starts at sample f*frame_stride (frame_stride == nfft for packed
 * frames, == hop for an overlapped STFT over one contiguous stream; the buffer
 * must hold (n_frames-1)*frame_stride + nfft samples).  out_db: n_frames*nfft
 * float32, row-major.  Blocks until out_db is complete.
 * Calls of up to 256 KiB (the reference's live shape: one 4096-sample buffer,
 * streamer.py:114-121) are served by a kernel that reads and writes pinned host
 * memory directly; larger calls stream through pinned staging in ~16 MiB chunks,
 * three in flight (staging memcpys by a small helper-thread pool, H2D, transform
 * and D2H overlapped).  The caller's arrays may be ordinary pageable memory. */
int sdrk_exec_host(sdrk_plan* plan, const void* iq_c64, size_t n_frames,
                   size_t frame_stride, float* out_db);

/* Device in / device out, asynchronous on `stream` (a hipStream_t, or NULL for
 * the plan's own stream).  Same layout rules as sdrk_exec_host. */
int sdrk_exec_device(sdrk_plan* plan, const void* d_iq_c64, size_t n_frames,
                     size_t frame_stride, float* d_out_db, void* stream);

/* Complex spectrum (no |.|, no log): out_c64[f][k] = fft(w*x)[k], optionally
 * fftshifted as the plan says.  Replaces np.fft.fft at streamer.py:119 alone;
 * used by the parity tests to check the transform before the log epilogue. */
int sdrk_exec_fft_host(sdrk_plan* plan, const void* iq_c64, size_t n_frames,
                       size_t frame_stride, void* out_c64);

/* Averaged periodogram (Welch, no detrend): out_psd[k] = scale * sum_f |fft(w*x_f)|^2[k]
 * over n_frames frames cut from one host stream at spacing frame_stride, in the plan's
 * shift order.  With scale = 1/(n_frames * Fs * sum(w^2)) this is matplotlib's
 * mlab.psd(..., window=hanning, noverlap=nfft-frame_stride) as plotted by the reference's
 * offline script (scripts/process_sigmf_data.py:188-189).  out_psd: nfft float32. */
int sdrk_welch_psd_host(sdrk_plan* plan, const void* iq_c64, size_t n_frames, size_t frame_stride,
                        float scale, float* out_psd);

/* Block until everything queued on the plan's own stream has finished. */
int sdrk_plan_sync(sdrk_plan* plan);

/* Timed replay for the bench harness: runs `launches` back-to-back
 * sdrk_exec_device() calls on the plan's stream bracketed by HIP events on that
 * stream and returns the elapsed milliseconds (all launches together). */
int sdrk_exec_device_timed(sdrk_plan* plan, const void* d_iq_c64, size_t n_frames,
                           size_t frame_stride, float* d_out_db, int launches,
                           float* elapsed_ms);

/* The same, returning the elapsed milliseconds of each of the `launches` launches
 * (events between consecutive launches; each_ms holds `launches` floats). */
int sdrk_exec_device_timed_each(sdrk_plan* plan, const void* d_iq_c64, size_t n_frames,
                                size_t frame_stride, float* d_out_db, int launches, float* each_ms);

/* ---- measurement probes (bench harness; no reference counterpart) ----------
 * sdrk_stream_ceiling_probe: a plain streaming kernel with the spectrum path's traffic
 *   shape at N = 4096 (32 KiB read + 16 KiB written per frame, no arithmetic), timed per
 *   launch on device buffers the caller provides (d_in: n*32 KiB, d_out: n*16 KiB) — the
 *   "measured-copy" ceiling SURVEY.md §8(d) asks to be reported next to the nominal 8 TB/s.
 * sdrk_copy_probe: a plain 1:1 copy of `bytes` (16 bytes per lane each way, non-temporal) from d_in to
 *   d_out, timed per launch — the shape MI355X_MICROARCH.md quotes 6.29 TB/s for; run on the same buffers it
 *   anchors the 2:1 probe above to a published figure (fastest of three grid sizes).  Bandwidth = 2 * bytes / time.
 * sdrk_host_link_probe: pinned-memory DMA rates in GB/s — `bytes` host-to-device, bytes/2
 *   device-to-host, and both at once (quoted on the upstream bytes): what the numpy
 *   boundary could reach at best.
 * sdrk_host_threads: helper threads of the host staging pool (SDRK_HOST_THREADS overrides). */
int sdrk_stream_ceiling_probe(int device, const void* d_in, void* d_out, size_t n_frames4096,
                              int launches, float* each_ms);
int sdrk_copy_probe(int device, const void* d_in, void* d_out, size_t bytes, int launches, float* each_ms);
int sdrk_host_link_probe(int device, size_t bytes, double* h2d_gbps, double* d2h_gbps,
                         double* duplex_gbps);
int sdrk_host_threads(void);

/* ---- synthetic IQ generator (bench / parity input, device resident) ------
 * Sample n of frame F (F = first_frame + f, 64-bit) is
 *     h = fmix32( fmix32(seed ^ lo32(F)) ^ fmix32(hi32(F) + 0x9E3779B1) ^ n )
 *     I = (h & 0xFFF) - 2048,  Q = ((h >> 12) & 0xFFF) - 2048      (as float32)
 * i.e. 12-bit integers like the AD9363 behind streamer.py:114, bit-identical
 * to the numpy generator in the host package (synth.py). */
int sdrk_synth_fill(int device, uint32_t seed, uint64_t first_frame, size_t n_frames,
                    int nfft, void* d_iq_c64, void* stream);

/* ---- per-row reductions for the spectrum's first consumer ------------------
 * The O(N) measurements of the reference's classifier helpers
 * (app/processing/classifier.py:163-212) computed next to the rows, so that a
 * consumer needs ~20 scalars per row instead of the row.  `rows` is n_rows*nfft
 * float32, on the host (rows_on_device = 0) or already on `device` (non-zero, e.g.
 * the output of sdrk_exec_device).
 *
 * stats: out[r*16 + i] (double), i =
 *   0 max | 1 sorted[rank] | 2 sorted[rank+1] (order statistics of the row, for the
 *   percentile noise floor, classifier.py:179-181) | 3 mean | 4 mean (x-mean)^2 |
 *   5 mean (x-mean)^4 (:191-198) | 6 mean ln p | 7 mean p, p = max(10^(x/10), 1e-15)
 *   (:183-189) | 8,9 first,last index with x >= max-3 | 10,11 ... max-10 | 12,13 ...
 *   max-20 (:163-170) | 14 argmax | 15 nfft.
 * thr: the adaptive peak threshold of classifier.py:55, max(noise_floor + 5,
 *   max - 0.9*snr + 5), with the percentile interpolated as numpy.percentile does on a
 *   float32 row: noise_floor = sorted[rank] + (sorted[rank+1]-sorted[rank])*gamma
 *   (`rank`, `gamma` = floor and fraction of float32(nfft-1)*float32(q/100)).
 * peaks: strict local maxima above the threshold, accepted left to right when
 *   >= min_distance bins after the previous accepted one (:200-212).
 *   out_idx: n_rows*max_peaks int32 (first max_peaks peaks of each row; the host entry points
 *   sdrk_row_features / sdrk_frame_features_host fill the unused slots with -1),
 *   out_count: n_rows int32 (may exceed max_peaks: the total found).
 *
 * sdrk_row_features: everything in ONE launch that reads each row from HBM once (rows up
 *   to 32768 bins are staged in LDS; longer rows are scanned in place); out_thr, out_idx
 *   and out_count may be NULL (then only the stats are produced).
 * sdrk_row_stats / sdrk_row_peaks: the two halves separately (sdrk_row_peaks takes the
 *   thresholds from the caller). */
int sdrk_row_features(int device, const float* rows, int rows_on_device, size_t n_rows, int nfft,
                      int rank, float gamma, int min_distance, int max_peaks, double* out_stats,
                      double* out_thr, int32_t* out_idx, int32_t* out_count);
int sdrk_row_stats(int device, const float* rows, int rows_on_device, size_t n_rows, int nfft,
                   int rank, double* out);
int sdrk_row_peaks(int device, const float* rows, int rows_on_device, size_t n_rows, int nfft,
                   const double* thresholds, int min_distance, int max_peaks, int32_t* out_idx,
                   int32_t* out_count);

/* IQ frames -> those measurements, the rows staying on the device (streamer.py:119,121
 * followed by classifier.py:163-212).  For nfft = 4096 the reductions run as the epilogue
 * of the transform kernel itself: the row exists only in LDS unless the caller passes a
 * buffer for it.  _device: every pointer is device memory, asynchronous on `stream` (or
 * the plan's stream); d_out_db, d_thr, d_idx/d_count may be NULL.  _host: host pointers,
 * blocking; out_db (the rows) may be NULL; batches of more than 32 MiB of IQ stream through the
 * pinned staging of sdrk_exec_host in chunks (pinned caller arrays are read in place). */
int sdrk_frame_features_device(sdrk_plan* plan, const void* d_iq_c64, size_t n_frames,
                               size_t frame_stride, float* d_out_db, int rank, float gamma,
                               int min_distance, int max_peaks, double* d_stats, double* d_thr,
                               int32_t* d_idx, int32_t* d_count, void* stream);
int sdrk_frame_features_host(sdrk_plan* plan, const void* iq_c64, size_t n_frames,
                             size_t frame_stride, int rank, float gamma, int min_distance,
                             int max_peaks, double* out_stats, double* out_thr, int32_t* out_idx,
                             int32_t* out_count, float* out_db);

/* The same measurements FINISHED on the device, for a whole batch: what classify_signal_advanced forms from its helpers'
 * results before the rule ladder (classifier.py:45-58) as planes of n_rows 8-byte words — the host receives arrays, not
 * packed per-row records that it then has to post-process row by row.  out_planes: SDRK_FEAT_PLANES * n_rows words;
 * plane P, row r at out_planes[P * n_rows + r]:
 *   double planes   SDRK_FEAT_MAX_DB                np.max(power_db)                                    (:46)
 *                   SDRK_FEAT_NOISE_FLOOR_DB        np.percentile(power_db, q) as numpy forms it on a float32 row (:179-181)
 *                   SDRK_FEAT_SNR_DB                float32(max - noise floor)                          (:46)
 *                   SDRK_FEAT_FLATNESS              clip(exp(mean ln p) / mean p, 0, 1)                 (:183-189)
 *                   SDRK_FEAT_KURTOSIS              0 if sigma < 1e-9 else m4 / m2^2                    (:191-198)
 *                   SDRK_FEAT_THRESHOLD_DB          the adaptive peak threshold                         (:55)
 *                   SDRK_FEAT_PEAK_SPACING_STD_HZ   np.std(np.diff(freqs[peaks])) over the kept peaks, 0 for < 3 (:214-219)
 *                   SDRK_FEAT_PEAK_DENSITY          peak_count / nfft                                   (:58)
 *                   SDRK_FEAT_BANDWIDTH_HZ + j      freqs[last] - freqs[first] of the bins within 3 / 10 / 20 dB (j = 0, 1, 2)
 *                                                   of the maximum, 0.0 when no bin qualifies           (:163-170)
 *   int64 planes    SDRK_FEAT_ARGMAX, SDRK_FEAT_PEAK_COUNT (total found; may exceed max_peaks)
 *                   SDRK_FEAT_OCCUPIED_BINS + 2 j   the (first, last) bin pairs behind bandwidth j: 2 * n_rows words,
 *                                                   row r at [2 r], [2 r + 1]
 * freqs: the nfft float64 bin frequencies (streamer.py:120), host memory; NULL: the three Hz quantities are 0.
 * out_idx (n_rows * max_peaks int32, unused slots -1) may be NULL: then no peaks are looked for (count 0). */
enum {
    SDRK_FEAT_MAX_DB = 0, SDRK_FEAT_NOISE_FLOOR_DB, SDRK_FEAT_SNR_DB, SDRK_FEAT_FLATNESS, SDRK_FEAT_KURTOSIS,
    SDRK_FEAT_THRESHOLD_DB, SDRK_FEAT_PEAK_SPACING_STD_HZ, SDRK_FEAT_PEAK_DENSITY, SDRK_FEAT_BANDWIDTH_HZ /* 3 planes */,
    SDRK_FEAT_ARGMAX = 11, SDRK_FEAT_PEAK_COUNT, SDRK_FEAT_OCCUPIED_BINS /* 6 planes */, SDRK_FEAT_PLANES = 19
};
int sdrk_row_features_planes(int device, const float* rows, int rows_on_device, size_t n_rows, int nfft,
                             int rank, float gamma, int min_distance, int max_peaks, const double* freqs,
                             void* out_planes, int32_t* out_idx);
int sdrk_frame_features_host_planes(sdrk_plan* plan, const void* iq_c64, size_t n_frames, size_t frame_stride,
                                    int rank, float gamma, int min_distance, int max_peaks, const double* freqs,
                                    void* out_planes, int32_t* out_idx, float* out_db);

/* ---- waterfall ring -------------------------------------------------------
 * Replaces deque(maxlen=100) / append / np.array(deque) at
 * dashboard/callbacks.py:19,176,182: a device-resident ring of the last
 * `maxlen` rows of nfft float32, read out oldest row first. */
int sdrk_waterfall_create(int device, int nfft, int maxlen, sdrk_waterfall** out);
int sdrk_waterfall_destroy(sdrk_waterfall* wf);
/* Append n_rows precomputed rows (host float32, n_rows*nfft). */
int sdrk_waterfall_append_rows(sdrk_waterfall* wf, const float* rows, size_t n_rows);
/* Transform n_frames host IQ frames with `plan` and append the resulting rows
 * without a host round trip of the rows (plan nfft/device must match). */
int sdrk_waterfall_append_iq(sdrk_waterfall* wf, sdrk_plan* plan, const void* iq_c64,
                             size_t n_frames, size_t frame_stride);
/* Same, IQ already on the device. */
int sdrk_waterfall_append_iq_device(sdrk_waterfall* wf, sdrk_plan* plan,
                                    const void* d_iq_c64, size_t n_frames,
                                    size_t frame_stride);
/* The same without waiting: the transforms are only enqueued on the ring's stream.  Every later call on this
 * waterfall (append, read, read_decimated, sync, destroy) is ordered behind them; d_iq_c64 must stay valid and
 * unmodified until one of those has returned after waiting (read, read_decimated, _end, sync). */
int sdrk_waterfall_append_iq_device_async(sdrk_waterfall* wf, sdrk_plan* plan, const void* d_iq_c64,
                                          size_t n_frames, size_t frame_stride);
/* Wait for everything enqueued on the ring's stream (`plan`, if not NULL: also report its launches' errors). */
int sdrk_waterfall_sync(sdrk_waterfall* wf, sdrk_plan* plan);
/* Number of valid rows (<= maxlen). */
int sdrk_waterfall_rows(const sdrk_waterfall* wf);
/* Copy the valid rows, oldest first, into out (capacity max_rows rows); the
 * number written is returned through n_rows.  If fewer than rows() fit, the
 * NEWEST max_rows are returned (still oldest-of-those first). */
int sdrk_waterfall_read(sdrk_waterfall* wf, float* out, size_t max_rows, size_t* n_rows);
/* Decimated read-out for display: like sdrk_waterfall_read, but every run of `factor`
 * consecutive bins is reduced on the device to one value — mode 0 = max (peak hold),
 * mode 1 = mean of the dB values — so `out` holds rows of nfft/factor float32.  factor
 * must divide nfft.  (Build-side extension: the reference plots full rows,
 * dashboard/callbacks.py:182-190, which is unusable at nfft = 2^20.) */
int sdrk_waterfall_read_decimated(sdrk_waterfall* wf, float* out, size_t max_rows, int factor, int mode,
                                  size_t* n_rows);
/* For nfft = 2^20 ... 2^22 the transform behind sdrk_waterfall_append_iq* also leaves every row max-hold-decimated by 16
 * beside the ring (1/16 of its size; the row pass has the sixteen neighbouring bins in LDS anyway), and a max-mode read-out
 * whose factor is a multiple of 16 is served from those — it reads 1/16 of the bytes instead of every 4 MiB row again, with
 * bit-identical results (a maximum does not depend on the order).  Rows appended as finished rows carry no such companion;
 * a read-out that touches one falls back to the full rows.  Returns how many of the valid rows carry one. */
int sdrk_waterfall_maxhold16_rows(const sdrk_waterfall* wf);
/* sdrk_waterfall_read_decimated in two halves, for a continuous channel (BASELINE config 5): _begin enqueues the
 * reduction behind everything appended so far and the device-to-host copy of its result on a second stream, and
 * returns; _end waits for that copy.  Between the two the caller can enqueue the next batch of frames
 * (sdrk_waterfall_append_iq_device_async), whose transform then runs while the previous batch's rows cross PCIe.
 * `out` must stay valid until _end (pinned memory makes the copy truly asynchronous).  The reduction itself runs on that
 * second stream too (behind an event), so the transform stream goes straight on; a later append waits for it only if it
 * overwrites ring rows still being reduced.  Up to TWO reads may be in flight per waterfall — _end completes the OLDEST —
 * so that a channel can enqueue batch i + 1 and begin its read-out before it collects batch i - 1 (the GPU then never
 * waits for the host); a third _begin is refused.  n_rows is known at _begin. */
int sdrk_waterfall_read_decimated_begin(sdrk_waterfall* wf, float* out, size_t max_rows, int factor, int mode,
                                        size_t* n_rows);
int sdrk_waterfall_read_decimated_end(sdrk_waterfall* wf);
int sdrk_waterfall_clear(sdrk_waterfall* wf);

#ifdef __cplusplus
}
#endif
#endif /* SDRK_H */
