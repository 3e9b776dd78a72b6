"""Developer helper (GPU box): every C entry point with arguments it must refuse (NULL pointers, devices that do not
exist, lengths of zero, mismatched handles, unknown flags).  Each call has to come back with a negative status and a
message — not a crash, not a silent success — and the handles involved must still work afterwards.
python tools/abi_invalid_probe.py    (prints one line per call; tests/test_parity_gpu.py asserts the same list)"""
import ctypes
import sys
from ctypes import POINTER, byref, c_float, c_int, c_size_t, c_void_p

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sdr_iq_visualizer_amd import _ffi


def cases():
    """Yield (description, status) for every refused call; leaves nothing allocated."""
    lib = _ffi.lib()
    NULL = c_void_p()
    plan, plan1k, wf = c_void_p(), c_void_p(), c_void_p()
    assert lib.sdrk_plan_create(0, 4096, 64, _ffi.WINDOW_RECT, None, c_float(1e-12), 1, byref(plan)) == 0
    assert lib.sdrk_plan_create(0, 1024, 64, _ffi.WINDOW_RECT, None, c_float(1e-12), 1, byref(plan1k)) == 0
    assert lib.sdrk_waterfall_create(0, 4096, 10, byref(wf)) == 0
    assert lib.sdrk_waterfall_append_rows(wf, np.zeros((2, 4096), dtype=np.float32).ctypes.data_as(c_void_p), 2) == 0
    x = np.zeros((2, 4096), dtype=np.complex64)
    rows = np.zeros((2, 4096), dtype=np.float32)
    stats = np.zeros((2, 16)); thr = np.zeros(2); idx = np.zeros((2, 8), dtype=np.int32); cnt = np.zeros(2, dtype=np.int32)
    xp, rp = x.ctypes.data_as(c_void_p), rows.ctypes.data_as(c_void_p)
    sp, tp, ip, cp = (a.ctypes.data_as(c_void_p) for a in (stats, thr, idx, cnt))
    d = c_void_p()
    assert lib.sdrk_dev_alloc(0, x.nbytes, byref(d)) == 0
    got = c_size_t(0)
    buf = ctypes.create_string_buffer(64)
    f3 = (c_float * 3)()
    out_plan, out_wf, out_ptr = c_void_p(), c_void_p(), c_void_p()

    yield "device_info: no such device", lib.sdrk_device_info(99, buf, 64)
    yield "dev_alloc: no such device", lib.sdrk_dev_alloc(99, 16, byref(out_ptr))
    yield "dev_alloc: NULL result pointer", lib.sdrk_dev_alloc(0, 16, None)
    yield "dev_mem_info: no such device", lib.sdrk_dev_mem_info(99, byref(got), byref(got))
    yield "memcpy_h2d: NULL destination", lib.sdrk_memcpy_h2d(0, NULL, xp, 16)
    yield "memcpy_d2h: NULL source", lib.sdrk_memcpy_d2h(0, xp, NULL, 16)
    yield "host_alloc: NULL result pointer", lib.sdrk_host_alloc(16, None)
    yield "host_free: not from host_alloc", lib.sdrk_host_free(xp)
    yield "host_register: NULL", lib.sdrk_host_register(NULL, 16)
    yield "host_register: zero bytes", lib.sdrk_host_register(xp, 0)
    yield "host_unregister: never registered", lib.sdrk_host_unregister(xp)
    yield "plan_create: nfft 0", lib.sdrk_plan_create(0, 0, 1, _ffi.WINDOW_RECT, None, c_float(1e-12), 1, byref(out_plan))
    yield "plan_create: nfft 1", lib.sdrk_plan_create(0, 1, 1, _ffi.WINDOW_RECT, None, c_float(1e-12), 1, byref(out_plan))
    yield "plan_create: nfft -4096", lib.sdrk_plan_create(0, -4096, 1, _ffi.WINDOW_RECT, None, c_float(1e-12), 1, byref(out_plan))
    yield "plan_create: nfft 2^23", lib.sdrk_plan_create(0, 1 << 23, 1, _ffi.WINDOW_RECT, None, c_float(1e-12), 1, byref(out_plan))
    yield "plan_create: no such device", lib.sdrk_plan_create(99, 4096, 1, _ffi.WINDOW_RECT, None, c_float(1e-12), 1, byref(out_plan))
    yield "plan_create: unknown window kind", lib.sdrk_plan_create(0, 4096, 1, 7, None, c_float(1e-12), 1, byref(out_plan))
    yield "plan_create: custom window without coefficients", lib.sdrk_plan_create(0, 4096, 1, _ffi.WINDOW_CUSTOM, None, c_float(1e-12), 1, byref(out_plan))
    yield "plan_create: negative eps", lib.sdrk_plan_create(0, 4096, 1, _ffi.WINDOW_RECT, None, c_float(-1.0), 1, byref(out_plan))
    yield "plan_create: NaN eps", lib.sdrk_plan_create(0, 4096, 1, _ffi.WINDOW_RECT, None, c_float(float("nan")), 1, byref(out_plan))
    yield "plan_create: NULL result pointer", lib.sdrk_plan_create(0, 4096, 1, _ffi.WINDOW_RECT, None, c_float(1e-12), 1, None)
    yield "plan_create_ex: unknown flag", lib.sdrk_plan_create_ex(0, 4096, 1, _ffi.WINDOW_RECT, None, c_float(1e-12), 1, 0x80, byref(out_plan))
    yield "plan_nfft: NULL plan", lib.sdrk_plan_nfft(None)
    yield "plan_sync: NULL plan", lib.sdrk_plan_sync(None)
    yield "exec_host: NULL plan", lib.sdrk_exec_host(None, xp, 1, 4096, rp)
    yield "exec_host: NULL input", lib.sdrk_exec_host(plan, NULL, 1, 4096, rp)
    yield "exec_host: NULL output", lib.sdrk_exec_host(plan, xp, 1, 4096, NULL)
    yield "exec_host: more frames than max_batch", lib.sdrk_exec_host(plan, xp, 65, 4096, rp)
    yield "exec_host: stride 0 with two frames", lib.sdrk_exec_host(plan, xp, 2, 0, rp)
    yield "exec_fft_host: NULL output", lib.sdrk_exec_fft_host(plan, xp, 1, 4096, NULL)
    yield "exec_device: NULL output", lib.sdrk_exec_device(plan, d, 1, 4096, NULL, None)
    yield "exec_device: NULL input", lib.sdrk_exec_device(plan, NULL, 1, 4096, d, None)
    yield "exec_device_timed: zero launches", lib.sdrk_exec_device_timed(plan, d, 1, 4096, d, 0, f3)
    yield "welch_psd_host: NULL output", lib.sdrk_welch_psd_host(plan, xp, 1, 4096, c_float(1.0), NULL)
    yield "welch_psd_host: zero frames", lib.sdrk_welch_psd_host(plan, xp, 0, 4096, c_float(1.0), rp)
    yield "synth_fill: nfft 0", lib.sdrk_synth_fill(0, 1, 0, 1, 0, d, None)
    yield "synth_fill: NULL buffer", lib.sdrk_synth_fill(0, 1, 0, 1, 4096, NULL, None)
    yield "row_features: no such device", lib.sdrk_row_features(99, rp, 0, 2, 4096, 800, c_float(0.5), 13, 8, sp, tp, ip, cp)
    yield "row_features: NULL rows", lib.sdrk_row_features(0, NULL, 0, 2, 4096, 800, c_float(0.5), 13, 8, sp, tp, ip, cp)
    yield "row_features: NULL stats", lib.sdrk_row_features(0, rp, 0, 2, 4096, 800, c_float(0.5), 13, 8, NULL, tp, ip, cp)
    yield "row_features: nfft 0", lib.sdrk_row_features(0, rp, 0, 2, 0, 0, c_float(0.5), 13, 8, sp, tp, ip, cp)
    yield "row_features: peak list without its counts", lib.sdrk_row_features(0, rp, 0, 2, 4096, 800, c_float(0.5), 13, 8, sp, tp, ip, NULL)
    yield "row_features: min_distance 0", lib.sdrk_row_features(0, rp, 0, 2, 4096, 800, c_float(0.5), 0, 8, sp, tp, ip, cp)
    yield "row_features: max_peaks 0", lib.sdrk_row_features(0, rp, 0, 2, 4096, 800, c_float(0.5), 13, 0, sp, tp, ip, cp)
    yield "row_stats: NULL output", lib.sdrk_row_stats(0, rp, 0, 2, 4096, 800, NULL)
    yield "row_peaks: NULL thresholds", lib.sdrk_row_peaks(0, rp, 0, 2, 4096, NULL, 13, 8, ip, cp)
    yield "row_peaks: max_peaks 0", lib.sdrk_row_peaks(0, rp, 0, 2, 4096, tp, 13, 0, ip, cp)
    yield "frame_features_host: NULL plan", lib.sdrk_frame_features_host(None, xp, 1, 4096, 800, c_float(0.5), 13, 8, sp, tp, ip, cp, NULL)
    yield "frame_features_host: NULL stats", lib.sdrk_frame_features_host(plan, xp, 1, 4096, 800, c_float(0.5), 13, 8, NULL, tp, ip, cp, NULL)
    yield "frame_features_host: counts without the list", lib.sdrk_frame_features_host(plan, xp, 1, 4096, 800, c_float(0.5), 13, 8, sp, tp, NULL, cp, NULL)
    planes = np.zeros((_ffi.FEAT_PLANES, 2)); freqs = np.zeros(4096)
    pp, fp = planes.ctypes.data_as(c_void_p), freqs.ctypes.data_as(c_void_p)
    yield "frame_features_host_planes: NULL plan", lib.sdrk_frame_features_host_planes(None, xp, 1, 4096, 800, c_float(0.5), 13, 8, fp, pp, ip, NULL)
    yield "frame_features_host_planes: NULL planes", lib.sdrk_frame_features_host_planes(plan, xp, 1, 4096, 800, c_float(0.5), 13, 8, fp, NULL, ip, NULL)
    yield "frame_features_host_planes: NULL frames", lib.sdrk_frame_features_host_planes(plan, NULL, 1, 4096, 800, c_float(0.5), 13, 8, fp, pp, ip, NULL)
    yield "frame_features_host_planes: peak list with max_peaks 0", lib.sdrk_frame_features_host_planes(plan, xp, 1, 4096, 800, c_float(0.5), 13, 0, fp, pp, ip, NULL)
    yield "row_features_planes: NULL planes", lib.sdrk_row_features_planes(0, rp, 0, 2, 4096, 800, c_float(0.5), 13, 8, fp, NULL, ip)
    yield "row_features_planes: NULL rows", lib.sdrk_row_features_planes(0, NULL, 0, 2, 4096, 800, c_float(0.5), 13, 8, fp, pp, ip)
    yield "row_features_planes: no such device", lib.sdrk_row_features_planes(99, rp, 0, 2, 4096, 800, c_float(0.5), 13, 8, fp, pp, ip)
    yield "row_features_planes: min_distance 0", lib.sdrk_row_features_planes(0, rp, 0, 2, 4096, 800, c_float(0.5), 0, 8, fp, pp, ip)
    yield "plan_staging_probe: NULL plan", lib.sdrk_plan_staging_probe(None, f3, 3, byref(c_int(0)))
    yield "plan_staging_probe: NULL count", lib.sdrk_plan_staging_probe(plan, f3, 3, None)
    yield "frame_features_device: NULL stats", lib.sdrk_frame_features_device(plan, d, 1, 4096, NULL, 800, c_float(0.5), 13, 8, NULL, NULL, NULL, NULL, None)
    yield "waterfall_create: nfft 0", lib.sdrk_waterfall_create(0, 0, 10, byref(out_wf))
    yield "waterfall_create: maxlen 0", lib.sdrk_waterfall_create(0, 4096, 0, byref(out_wf))
    yield "waterfall_create: no such device", lib.sdrk_waterfall_create(99, 4096, 10, byref(out_wf))
    yield "waterfall_create: NULL result pointer", lib.sdrk_waterfall_create(0, 4096, 10, None)
    yield "waterfall_rows: NULL ring", lib.sdrk_waterfall_rows(None)
    yield "waterfall_clear: NULL ring", lib.sdrk_waterfall_clear(None)
    yield "waterfall_append_rows: NULL ring", lib.sdrk_waterfall_append_rows(None, rp, 1)
    yield "waterfall_append_rows: NULL rows", lib.sdrk_waterfall_append_rows(wf, NULL, 1)
    yield "waterfall_append_iq: NULL plan", lib.sdrk_waterfall_append_iq(wf, None, xp, 1, 4096)
    yield "waterfall_append_iq: plan of another length", lib.sdrk_waterfall_append_iq(wf, plan1k, xp, 1, 1024)
    yield "waterfall_append_iq: NULL frames", lib.sdrk_waterfall_append_iq(wf, plan, NULL, 1, 4096)
    yield "waterfall_append_iq_device: plan of another length", lib.sdrk_waterfall_append_iq_device(wf, plan1k, d, 1, 1024)
    yield "waterfall_append_iq_device_async: NULL buffer", lib.sdrk_waterfall_append_iq_device_async(wf, plan, NULL, 1, 4096)
    yield "waterfall_append_iq_device: stride 0 with two frames", lib.sdrk_waterfall_append_iq_device(wf, plan, d, 2, 0)
    yield "waterfall_read: NULL output", lib.sdrk_waterfall_read(wf, NULL, 1, byref(got))
    yield "waterfall_read_decimated: factor that does not divide", lib.sdrk_waterfall_read_decimated(wf, rp, 1, 3, 0, byref(got))
    yield "waterfall_read_decimated: factor 0", lib.sdrk_waterfall_read_decimated(wf, rp, 1, 0, 0, byref(got))
    yield "waterfall_read_decimated: unknown mode", lib.sdrk_waterfall_read_decimated(wf, rp, 1, 4, 7, byref(got))
    yield "waterfall_read_decimated_begin: NULL output", lib.sdrk_waterfall_read_decimated_begin(wf, NULL, 1, 4, 0, byref(got))
    yield "waterfall_sync: NULL ring", lib.sdrk_waterfall_sync(None, None)
    yield "stream_ceiling_probe: NULL buffers", lib.sdrk_stream_ceiling_probe(0, NULL, NULL, 16, 1, f3)
    yield "copy_probe: zero launches", lib.sdrk_copy_probe(0, d, d, 1024, 0, f3)
    yield "host_link_probe: no such device", lib.sdrk_host_link_probe(99, 1 << 20, None, None, None)

    # the handles still work (and the planes form with neither a frequency axis nor a peak table is a valid call)
    assert lib.sdrk_frame_features_host_planes(plan, xp, 2, 4096, 800, c_float(0.5), 13, 8, NULL, pp, NULL, NULL) == 0
    assert np.all(planes[_ffi.FEAT_MAX_DB] == np.float32(-240.00002)) and np.all(planes.view(np.int64)[_ffi.FEAT_PEAK_COUNT] == 0)
    assert lib.sdrk_exec_host(plan, xp, 2, 4096, rp) == 0 and np.all(rows == np.float32(-240.00002))
    assert lib.sdrk_waterfall_append_rows(wf, rp, 2) == 0 and lib.sdrk_waterfall_rows(wf) == 4
    assert lib.sdrk_waterfall_destroy(wf) == 0 and lib.sdrk_plan_destroy(plan) == 0 and lib.sdrk_plan_destroy(plan1k) == 0
    assert lib.sdrk_dev_free(0, d) == 0
    for leaked in (out_plan, out_wf, out_ptr):
        assert not leaked.value, "a refused call left a handle behind"


if __name__ == "__main__":
    bad = 0
    for what, status in cases():
        msg = _ffi.lib().sdrk_last_error().decode() if status < 0 else ""
        print(f"{'ok ' if status < 0 else 'ACCEPTED'} {status:4d}  {what}  {msg}", flush=True)
        bad += status >= 0
    print("refused everything" if not bad else f"{bad} call(s) were not refused")
