"""Minimal SigMF reader/writer for ``cf32_le`` recordings (BASELINE.json config 1).

Format = what the reference's dashboard exports (app/dashboard/callbacks.py:285-319):
a ``.sigmf-data`` file of interleaved little-endian float32 I,Q and a ``.sigmf-meta``
JSON with ``global.core:datatype = "cf32_le"``, ``core:sample_rate``, ``core:version`` and
``captures[0].core:frequency`` / ``core:sample_start`` / ``core:datetime``, optionally
zipped together with a README.  The reference's offline reader
(scripts/process_sigmf_data.py:20-62) goes through the ``sigmf`` package, which is not a
dependency here: the format is simple enough to read directly.  File I/O only — no
signal processing in this module.
"""
from __future__ import annotations

import io
import json
import os
import zipfile
from datetime import datetime, timezone
from typing import Optional, Tuple

import numpy as np

_DTYPES = {"cf32_le": np.dtype("<c8"), "cf64_le": np.dtype("<c16"),
           "ci16_le": np.dtype("<i2"), "ci8": np.dtype("i1")}


def make_metadata(sample_rate: float, center_freq: float, *, description: str = "IQ recording",
                  author: str = "sdr_iq_visualizer_amd", hw: str = "", when: Optional[datetime] = None) -> dict:
    """Metadata dict with the keys of app/dashboard/callbacks.py:285-304."""
    when = when or datetime.now(timezone.utc)
    return {
        "global": {
            "core:datatype": "cf32_le",
            "core:sample_rate": int(sample_rate),
            "core:version": "1.0.0",
            "core:description": description,
            "core:author": author,
            "core:hw": hw,
            "core:license": "CC0-1.0",
        },
        "captures": [{
            "core:sample_start": 0,
            "core:frequency": int(center_freq),
            "core:datetime": when.strftime("%Y-%m-%dT%H:%M:%S.%f") + "Z",
        }],
        "annotations": [],
    }


def write_sigmf(base_path: str, samples, sample_rate: float, center_freq: float, **meta_kw) -> Tuple[str, str]:
    """Write ``<base>.sigmf-data`` (cf32_le) and ``<base>.sigmf-meta``; returns both paths."""
    x = np.asarray(samples)
    if x.dtype != np.complex64:                       # callbacks.py:307-308
        x = x.astype(np.complex64)
    data_path, meta_path = base_path + ".sigmf-data", base_path + ".sigmf-meta"
    x.astype("<c8", copy=False).tofile(data_path)     # callbacks.py:310 (tobytes)
    with open(meta_path, "w") as fh:
        json.dump(make_metadata(sample_rate, center_freq, **meta_kw), fh, indent=2)
    return data_path, meta_path


def _decode(raw: bytes, datatype: str) -> np.ndarray:
    if datatype not in _DTYPES:
        raise ValueError(f"unsupported core:datatype {datatype!r} (supported: {sorted(_DTYPES)})")
    a = np.frombuffer(raw, dtype=_DTYPES[datatype])
    if datatype.startswith("ci"):                     # interleaved integers -> complex64
        a = a[: a.size // 2 * 2].astype(np.float32).view(np.complex64)
    return a.astype(np.complex64, copy=False)


def read_sigmf(path: str, max_samples: Optional[int] = None) -> Tuple[np.ndarray, dict]:
    """Read a recording given its ``.sigmf-meta``, ``.sigmf-data``, base name, or the
    ``.zip`` the dashboard's download button produces.  Returns ``(samples complex64, meta)``;
    ``meta['sample_rate']`` and ``meta['center_freq']`` are lifted out for convenience."""
    if path.endswith(".zip"):
        with zipfile.ZipFile(path) as z:
            names = z.namelist()
            meta_name = next(n for n in names if n.endswith(".sigmf-meta"))
            data_name = next(n for n in names if n.endswith(".sigmf-data"))
            meta = json.loads(z.read(meta_name))
            raw = z.read(data_name)
    else:
        base = path
        for ext in (".sigmf-meta", ".sigmf-data"):
            if base.endswith(ext):
                base = base[: -len(ext)]
        with open(base + ".sigmf-meta") as fh:
            meta = json.load(fh)
        itemsize = _DTYPES.get(meta.get("global", {}).get("core:datatype", "cf32_le"), np.dtype("<c8")).itemsize
        with open(base + ".sigmf-data", "rb") as fh:
            raw = fh.read() if max_samples is None else fh.read(int(max_samples) * itemsize * 2)
    g = meta.get("global", {})
    samples = _decode(raw, g.get("core:datatype", "cf32_le"))
    if max_samples is not None:
        samples = samples[: int(max_samples)]
    caps = meta.get("captures") or [{}]
    out = dict(meta)
    out["sample_rate"] = float(g.get("core:sample_rate", 1.0))
    out["center_freq"] = float(caps[0].get("core:frequency", 0.0))
    return samples, out


def to_zip_bytes(samples, sample_rate: float, center_freq: float, base_name: str = "sdr_sample") -> bytes:
    """The dashboard's download payload (callbacks.py:313-341): a zip of data + meta."""
    x = np.asarray(samples).astype(np.complex64)
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, "w", zipfile.ZIP_DEFLATED) as z:
        z.writestr(f"{base_name}.sigmf-data", x.tobytes())
        z.writestr(f"{base_name}.sigmf-meta", json.dumps(make_metadata(sample_rate, center_freq), indent=2))
    return buf.getvalue()
