"""CPU-side checks: the C-ABI library loads and exports every symbol include/mpqr.h declares; host-only
helpers (reader, log, flop models, partition arithmetic, generator) agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mpqr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpqr_[a-z0-9_]+)\s*\(", text)) - {"mpqr_bcast_fn"})


def test_library_exports_every_declared_symbol():
    from mixedprecisionblockqr_amd import _lib
    L = C.CDLL(_lib.build())
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/mpqr.h but not exported by libmpqr.so"
    assert sorted(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ set(names)


def test_no_device_is_reported_not_faked():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import mixedprecisionblockqr_amd as mp
    with pytest.raises(mp.MpqrError) as e:
        mp.Handle(0)
    assert e.value.code == mp._lib.ERR_NO_DEVICE


def test_host_helpers_match_oracle(po, tmp_path):
    import mixedprecisionblockqr_amd as mp
    assert np.array_equal(mp.generate_matrix(37, 19, seed=99), po.generate(37, 19, seed=99))
    assert mp.h_qr_flops_per_second(3.5, 600, 400) == po.lib().orc_qr_flops_per_second(3.5, 600, 400)
    f = mp.flops(16384, 16384, 128)
    assert abs(f["geqrf"] - 5.864e12) < 1e9 and abs(f["trailing"] - 5.881e12) < 1e9 and abs(f["panel"] - 3.46e10) < 1e8
    f = mp.flops(2048, 2048, 64)
    assert abs(f["geqrf"] - 1.145e10) < 1e7 and abs(f["trailing"] - 1.157e10) < 1e7
    p = tmp_path / "A_000000100.txt"
    p.write_text("4 3\n  0 0 1.5\n1 2 -2.25e-3\n   3 1 7\n1 2 4.0\n\n")
    assert np.array_equal(mp.read_euroc_jacobian(p), po.read_euroc_jacobian(str(p)))
    M = (np.random.default_rng(0).standard_normal((30, 20)) * (np.random.default_rng(1).random((30, 20)) < .2)).astype(np.float32)
    mp.write_euroc_jacobian(tmp_path / "B.txt", M)
    assert np.array_equal(po.read_euroc_jacobian(str(tmp_path / "B.txt")), M)
    with pytest.raises(mp.MpqrError):
        mp.read_euroc_jacobian(tmp_path / "nope.txt")
    assert mp.error_passes(2 ** -11 * 10, 10, 11) and not mp.error_passes(2 ** -11 * 10.01, 10, 11)


def test_results_log_format(tmp_path):
    """CSV schema of h_write_results_to_log (Cuda/qr.cu:58-83), readable by Cuda/performance/util.py."""
    import mixedprecisionblockqr_amd as mp
    d = tmp_path / "log"
    mp.h_write_results_to_log(600, 400, 12.5, 33.25, 154.0, "gpu_block", log_dir=d)
    mp.h_write_results_to_log(240, 160, 1.5, 3.0, 90.0, "gpu_block", log_dir=d)
    lines = (d / "gpu_block.txt").read_text().splitlines()
    assert lines[0] == "rows,cols,runtime,flops,error"
    assert lines[1] == "600.000000,400.000000,12.500000,33.250000,154.000000" and len(lines) == 3


def test_block_cyclic_partition():
    import mixedprecisionblockqr_amd as mp
    L = mp._lib.lib()
    n, b, G = 1000, 128, 3
    seen = []
    for rank in range(G):
        k = L.mpqr_part_local_cols(n, b, G, rank)
        cols = [L.mpqr_part_global_index(lc, b, G, rank) for lc in range(k)]
        assert all(L.mpqr_part_owner(c, b, G) == rank for c in cols)
        assert [L.mpqr_part_local_index(c, b, G) for c in cols] == list(range(k))
        seen += cols
    assert sorted(seen) == list(range(n))


def test_struct_layouts_match_the_header():
    """The ctypes mirrors of mpqr_opts / mpqr_metrics / mpqr_timings: same size as compiled into the library and the same
    field names, in order, as include/mpqr.h declares."""
    import ctypes as C, os, re
    from mixedprecisionblockqr_amd import _lib as L
    out = (C.c_int * 3)()
    L.lib().mpqr_abi_sizes(out)
    assert list(out) == [C.sizeof(L.MpqrOpts), C.sizeof(L.MpqrMetrics), C.sizeof(L.MpqrTimings)]
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mpqr.h")).read()
    for name, cls in (("mpqr_opts", L.MpqrOpts), ("mpqr_metrics", L.MpqrMetrics), ("mpqr_timings", L.MpqrTimings)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = [re.sub(r"\[.*", "", d.split()[-1]) for d in body.split(";") if d.strip()]
        assert fields == [f for f, _ in cls._fields_], (name, fields)
