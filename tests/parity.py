"""Error metrics shared by the parity tests.

Tolerance (BASELINE.json north_star): outputs match numpy.fft within 1e-5
relative error.  Per-bin relative error is ill-posed at spectral nulls (an exact
tone leaves rounding noise in every other bin), so "relative" is taken against
the frame's peak magnitude, as SURVEY.md §7.3 defines it:

    max_k | |X_gpu[k]| - |X_ref[k]| |  <=  REL_TOL * max_k |X_ref[k]|

evaluated in float64 on magnitudes recovered from the dB rows (the additive
floor eps is part of both sides).  On bins within 20 dB of the peak the same
bound implies |delta dB| <= 8.7e-4; STRONG_DB_TOL checks that directly.
"""
import numpy as np

REL_TOL = 1e-5
STRONG_DB_TOL = 1e-3


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


def assert_complex_parity(got, ref, rel=REL_TOL, what=""):
    got = np.asarray(got).astype(np.complex128)
    ref = np.asarray(ref).astype(np.complex128)
    assert got.shape == ref.shape
    scale = np.abs(ref).max(axis=-1, keepdims=True)
    scale = np.where(scale == 0, 1.0, scale)
    err = (np.abs(got - ref) / scale).max()
    assert err <= rel, f"{what}: complex spectrum error {err:.3e} > {rel:g}"
