"""Error metrics shared by the parity tests.

Tolerance (BASELINE.json north_star): outputs match numpy.fft within 1e-5
relative error.  Per-bin relative error is ill-posed at spectral nulls (an exact
tone leaves rounding noise in every other bin), so "relative" is taken against
the frame's peak magnitude, as SURVEY.md §7.3 defines it:

    max_k | |X_gpu[k]| - |X_ref[k]| |  <=  REL_TOL * max_k |X_ref[k]|

evaluated in float64 on magnitudes recovered from the dB rows (the additive
floor eps is part of both sides).  On bins within 20 dB of the peak the same
bound implies |delta dB| <= 8.7e-4; STRONG_DB_TOL checks that directly.

The peak-relative bound says little about WEAK bins (a tone 60 dB under the peak is
held to 1 % of its own magnitude by it), so the reference-produced fixtures are also
checked in dB down to DEEP_DEPTH_DB under the frame peak (assert_db_parity_deep): the
expression of streamer.py:121 evaluated in float32 by numpy differs from its float64
evaluation by < 2e-5 dB on every such bin of every fixture, so DEEP_DB_TOL = 0.01 dB
leaves the GPU's 1-ulp sqrt / log2 and its own twiddle rounding ample room and still
catches a wrong twiddle, a lost low-order term or a mis-scaled floor.
"""
import numpy as np

REL_TOL = 1e-5
STRONG_DB_TOL = 1e-3
DEEP_DEPTH_DB = 70.0
DEEP_DB_TOL = 1e-2


def mag_from_db(db):
    return np.power(10.0, np.asarray(db, dtype=np.float64) / 20.0)


def peak_rel_err(got_db, ref_db):
    """Per-frame max |mag_got - mag_ref| / max mag_ref."""
    mg, mr = mag_from_db(got_db), mag_from_db(ref_db)
    scale = mr.max(axis=-1, keepdims=True)
    return (np.abs(mg - mr) / scale).max(axis=-1)


def assert_db_parity(got_db, ref_db, rel=REL_TOL, what=""):
    got_db, ref_db = np.asarray(got_db), np.asarray(ref_db)
    assert got_db.shape == ref_db.shape, (got_db.shape, ref_db.shape)
    assert got_db.dtype == np.float32, got_db.dtype
    assert np.all(np.isfinite(got_db) == np.isfinite(ref_db)), f"{what}: finite mask differs"
    fin = np.isfinite(ref_db)
    g = np.where(fin, got_db, 0.0)
    r = np.where(fin, ref_db, 0.0)
    err = peak_rel_err(g, r)
    assert np.all(err <= rel), f"{what}: peak-relative magnitude error {err.max():.3e} > {rel:g}"
    mr = mag_from_db(r)
    strong = fin & (mr >= 0.1 * mr.max(axis=-1, keepdims=True))
    ddb = np.abs(g.astype(np.float64) - r.astype(np.float64))[strong]
    if ddb.size:
        assert ddb.max() <= STRONG_DB_TOL, f"{what}: |delta dB| {ddb.max():.3e} on strong bins"


def assert_db_parity_deep(got_db, ref_db, depth_db=DEEP_DEPTH_DB, tol_db=DEEP_DB_TOL, what=""):
    """|delta dB| <= tol_db on every bin within depth_db of its frame's peak (reference rows).  Returns the
    largest difference seen, for the test log."""
    g = np.asarray(got_db, dtype=np.float64)
    r = np.asarray(ref_db, dtype=np.float64)
    assert g.shape == r.shape, (g.shape, r.shape)
    fin = np.isfinite(r)
    peak = np.where(fin, r, -np.inf).max(axis=-1, keepdims=True)
    sel = fin & (r >= peak - depth_db)
    assert np.all(np.isfinite(g[sel])), f"{what}: non-finite value on a bin within {depth_db:g} dB of the peak"
    d = np.abs(g - r)[sel]
    worst = float(d.max()) if d.size else 0.0
    assert worst <= tol_db, f"{what}: |delta dB| {worst:.3e} > {tol_db:g} within {depth_db:g} dB of the peak ({int(sel.sum())} bins)"
    return worst


def assert_complex_parity(got, ref, rel=REL_TOL, what=""):
    got = np.asarray(got).astype(np.complex128)
    ref = np.asarray(ref).astype(np.complex128)
    assert got.shape == ref.shape
    scale = np.abs(ref).max(axis=-1, keepdims=True)
    scale = np.where(scale == 0, 1.0, scale)
    err = (np.abs(got - ref) / scale).max()
    assert err <= rel, f"{what}: complex spectrum error {err:.3e} > {rel:g}"
