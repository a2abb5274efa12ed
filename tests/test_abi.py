"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/arcnerf_hip.h
declares, validates arguments before touching the device, and its host-only helpers (pcg32) match the oracle."""
import ctypes as C
import os
import subprocess

import pytest

from arcnerf_amd import _native as N


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(N.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return N.lib()


def test_header_declares_and_library_exports_every_entry_point(lib):
    protos = N.parse_header()
    assert len(protos) >= 30
    for name in protos:
        assert hasattr(lib, name), name
    # and nothing undeclared leaks out of the library (only arcn_* symbols are visible)
    out = subprocess.check_output(['nm', '-D', '--defined-only', N.LIB_PATH]).decode()
    exported = {ln.split()[-1] for ln in out.splitlines() if ' T ' in ln}
    assert {e for e in exported if e.startswith('arcn_')} == set(protos)


def test_version_and_error_string(lib):
    assert lib.arcn_version() == 100
    rc = lib.arcn_hashgrid_fwd(None, None, None, None, None, 10, None, None)
    assert rc == -1
    assert b'hashgrid_fwd' in lib.arcn_last_error()


def test_argument_validation_happens_before_any_launch(lib):
    d = N.make_hashgrid_desc([15, 22], [0, 4096, 16263], 3, [-1] * 3, [1] * 3)  # n_feat 3 unsupported
    one = C.c_void_p(16)
    assert lib.arcn_hashgrid_fwd(one, one, C.addressof(d), one, None, 4, None, None) == -1
    m = N.make_mlp_desc([32, 256, 3])  # width > 128
    assert lib.arcn_mlp_fwd(one, one, None, C.addressof(m), one, None, 4, 4, None, None) == -1
    m = N.make_mlp_desc([32, 64, 3], has_bias=True)
    assert lib.arcn_mlp_fwd(one, one, None, C.addressof(m), one, None, 4, 4, None, None) == -1  # bias missing
    assert lib.arcn_sh_fwd(one, 6, 0, one, 4, None) == -1
    assert lib.arcn_ray_marching_fwd(None, None, None, one, None, None, 0, 4, 8, 0, 0, None, None, None, None, None, None,
                                     None, None) == -1
    assert lib.arcn_sparse_volume_sampling(one, one, one, one, 8, 0.0, one, 8, one, 0.0, 1, 1, one, one, None, 4, None) == -1
    # empty inputs are a no-op success (reference: linear_kernel returns on n_elements <= 0, include/common.h:41-44)
    assert lib.arcn_hashgrid_fwd(None, None, None, None, None, 0, None, None) == 0
    assert lib.arcn_ray_marching_fwd(None, None, None, None, None, None, 0, 0, 8, 0, 0, None, None, None, None, None, None,
                                     None, None) == 0


def _ngp_levels():
    from arcnerf_amd.pipeline import hashgrid_level_table
    res, offs = hashgrid_level_table(16, 19, 16, 2048)
    return res, offs, 2, [-1.5] * 3, [1.5] * 3


def test_workspace_queries(lib):
    m = N.make_mlp_desc([32, 64, 64, 3])
    # hidden activations, rows padded to a multiple of 16 (the level-major / concat entry points keep them in 16-sample tiles)
    assert lib.arcn_mlp_acts_floats(C.addressof(m), 1000) == 1008 * 128
    assert lib.arcn_mlp_acts_floats(C.addressof(m), 1024) == 1024 * 128
    # dpre of every layer + per-workgroup partial dW tiles (2 slabs) of the 3 (layer, 64x64 quadrant) pairs
    assert lib.arcn_mlp_scratch_floats(C.addressof(m), 1000) == 1000 * 131 + 2 * 3 * (4096 + 64)
    # hash-grid scatter: bin counters + 16-byte records, never less than 8 floats per (level, sample)
    h = N.make_hashgrid_desc(*_ngp_levels())
    n = 1 << 16
    w = lib.arcn_hashgrid_bwd_workspace_floats(C.addressof(h), n)
    assert n * 16 * 8 <= w <= n * 16 * 64
    assert lib.arcn_hashgrid_bwd_workspace_floats(C.addressof(h), 0) == 0
    # too small a workspace is an argument error, not a memory fault (validated before any device work)
    one = C.c_void_p(16)
    assert lib.arcn_hashgrid_bwd(one, one, one, C.addressof(h), one, None, one, 64, n, None, None) == -1
    assert b'workspace' in lib.arcn_last_error()


def test_host_pcg32_matches_oracle(lib, oracle):
    si = (C.c_uint64 * 2)()
    lib.arcn_pcg32_seed(9121, 1, C.addressof(si))
    ref = oracle.Pcg32(9121)
    assert (int(si[0]), int(si[1])) == (ref.state, ref.inc)
    for _ in range(3):
        lib.arcn_pcg32_advance(C.addressof(si), 1 << 32)
        ref.advance()
        assert int(si[0]) == ref.state


def test_ops_refuse_cpu_tensors():
    import torch
    from arcnerf_amd.ops import functional as F
    with pytest.raises(RuntimeError):
        F.freq_fwd(torch.zeros(4, 3), 4)
