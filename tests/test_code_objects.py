"""What the compiler made of the kernels, read from the gfx950 code objects inside the built libsdrk.so (no GPU needed): register
budgets that the occupancy the kernels are designed for depends on, and no scratch — a kernel that silently starts spilling (as the
complex-output row pass did when complex values became aligned register pairs, round 5) loses its fourth wave per SIMD or pays scratch
traffic, and nothing else in the CPU suite would notice."""
import os
import re
import shutil
import subprocess

import pytest

from sdr_iq_visualizer_amd import _ffi

LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels(tmp_path):
    lib = _ffi.library_path()
    if not (os.path.exists(lib) and os.path.exists(os.path.join(LLVM, "llvm-objdump")) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(lib, work / "libsdrk.so")
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "libsdrk.so"], cwd=work, check=True, capture_output=True)
    rows, cur = [], None
    for co in sorted(work.glob("libsdrk.so.*gfx950*")):
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        for ln in notes.splitlines():
            m = re.match(r"\s*\.(name|private_segment_fixed_size|vgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|max_flat_workgroup_size):\s*(\S+)", ln)
            if not m:
                continue
            if m.group(1) == "name":
                if not m.group(2).startswith("_Z"):      # argument names share the key
                    continue
                cur = {"name": m.group(2)}
                rows.append(cur)
            elif cur is not None:
                cur[m.group(1)] = int(m.group(2))
    return [r for r in rows if "vgpr_count" in r]


def test_kernels_fit_their_register_budgets_and_do_not_spill(tmp_path):
    ks = _kernels(tmp_path)
    assert len(ks) >= 130                                           # every template instantiation the plans can pick
    by = {k["name"]: k for k in ks}
    spilling = {n: (k["private_segment_fixed_size"], k["vgpr_spill_count"]) for n, k in by.items()
                if k["private_segment_fixed_size"] or k["vgpr_spill_count"]}   # (SGPRs spilled into VGPR lanes cost no memory)
    assert not spilling, spilling

    def one(fragment):
        hits = [k for n, k in by.items() if fragment in n]
        assert hits, fragment
        return hits

    # flagship: three workgroups of 256 threads per CU = three waves per SIMD -> at most 512 / 3 = 170 registers, 160 KiB / 3 of LDS
    for k in one("fft4096_kernelILb"):
        assert k["vgpr_count"] <= 128 and k["group_segment_fixed_size"] <= 160 * 1024 // 3   # (126 since the pass-1 twiddles are powers of one base)
    # the feature kernel: four per CU -> 128 registers, 40 KiB
    for k in one("fft4096_features_kernel"):
        assert k["vgpr_count"] <= 128 and k["group_segment_fixed_size"] <= 40 * 1024
    # row passes that count on four waves per SIMD (512-thread workgroups, two per CU), both epilogues
    for k in one("row_pass_kernelILi9E"):
        assert k["vgpr_count"] <= 128
    for k in one("row_pass_pair_kernel"):
        assert k["vgpr_count"] <= 128
    # two waves per SIMD: the staged col pass and the one-wave-per-row row pass
    for k in one("col_pass_staged_kernel") + one("row_pass_wave_kernel"):
        assert k["vgpr_count"] <= 256


def test_transform_kernels_use_packed_complex_arithmetic_and_the_feature_kernel_does_not(tmp_path):
    """cplx.h: packed register pairs everywhere except the issue-bound, four-waves-per-SIMD feature kernel (DESIGN_APPENDIX.md A.13)."""
    lib = _ffi.library_path()
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")) or not os.path.exists(lib):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(lib, work / "libsdrk.so")
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "libsdrk.so"], cwd=work, check=True, capture_output=True)
    counts = {}
    for co in sorted(work.glob("libsdrk.so.*gfx950*")):
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", str(co)], check=True, capture_output=True, text=True).stdout
        cur = None
        for ln in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
            if m:
                cur = m.group(1)
                counts.setdefault(cur, [0, 0])
            elif cur and "\tv_" in ln or (cur and " v_" in ln):
                counts[cur][1] += 1
                if "v_pk_add_f32" in ln or "v_pk_fma_f32" in ln or "v_pk_mul_f32" in ln:
                    counts[cur][0] += 1
    flag = next(v for n, v in counts.items() if "fft4096_kernelILb1ELi0E" in n)
    feat = next(v for n, v in counts.items() if "fft4096_features_kernelILb1E" in n)
    assert flag[0] >= 300 and flag[1] <= 520, flag                  # 327 packed of 473 VALU instructions (scalar form: ~800)
    assert feat[0] == 0, feat
