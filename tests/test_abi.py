"""CPU: the C-ABI shared library loads without a GPU and exports exactly the symbols include/spgan_hip.h declares;
the ctypes table in spgan/_lib.py covers all of them; the product refuses to run without the library / on CPU tensors."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "spgan_hip.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(spgan_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_typed():
    from spgan import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    syms = declared_symbols()
    assert len(syms) >= 40
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "libspgan_hip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES.keys()) == syms, set(_lib.SIGNATURES.keys()) ^ set(syms)
    _lib.load()
    assert _lib.load().spgan_version() >= 1 and _lib.load().spgan_arch() == b"gfx950"


def test_argument_validation_without_gpu():
    """Status codes instead of exit(-1): bad sizes are rejected before any launch (no GPU needed)."""
    from spgan import _lib
    lib = _lib.load()
    assert lib.spgan_knn(None, 1, 16, 3, 4, 0, None, None) == -22
    assert lib.spgan_knn(1, 1, 16, 3, 40, 0, 1, None) == -22          # k > 32
    assert lib.spgan_gemm_tn_ws_bytes(0, 4, 4) == 0
    with pytest.raises(RuntimeError):
        _lib.check(-22, "knn", B=1)
    # the grouped launches (round 5): count outside 1 .. SPGAN_GROUP_MAX, problems of different geometry, missing pointers
    one = (_lib.GemmDualArgs * 1)()
    assert lib.spgan_gemm_dual_multi(one, 0, None) == -22 and lib.spgan_gemm_dual_multi(one, 5, None) == -22
    assert lib.spgan_gemm_dual_multi(one, 1, None) == -22                 # an empty argument block
    assert lib.spgan_colstats_finalize_multi((_lib.ColFinalizeArgs * 1)(), 1, None) == -22
    assert lib.spgan_pool_bwd_stats_prep_multi((_lib.PoolBwdArgs * 1)(), 1, None) == -22
    assert lib.spgan_wgrad_collapse_multi((_lib.WgradCollapseArgs * 2)(), 2, None) == -22
    assert lib.spgan_gemm_tn_skinny_multi((_lib.GemmTNArgs * 1)(), 1, None) == -22
    assert lib.spgan_multi_addn(_lib.MultiAddNArgs(), None) == -22
    assert lib.spgan_collapse_prep(_lib.CollapsePrepArgs(), None) == -22
    assert lib.spgan_gp_penalty_fwd_bwd(None, 1, 1, 1.0, 1.0, None, None, None, None, None, None) == -22
    assert _lib.GROUP_MAX == 4 and _lib.MULTI_ADDN_MAX == 32


def test_cpu_tensors_are_refused():
    import spgan

    class O:
        np = 64; nk = 20; nz = 8; softmax = True; off = False; attn = False; use_head = False; eql = False; z_norm = False; small_d = False
    D = spgan.Discriminator(O)
    with pytest.raises(RuntimeError, match="no CPU"):
        D(torch.zeros(2, 3, 64))
    with pytest.raises(RuntimeError, match="GPU"):
        spgan.ops.knn(torch.zeros(32, 3), 1, 32, 4)
    O.attn = True; O.eql = True                           # non-default variants construct on the CPU, but run only on the GPU
    G = spgan.Generator(O)
    assert "attn.gamma" in G.state_dict() and "head.0.conv.weight_orig" in G.state_dict()
    with pytest.raises(RuntimeError, match="no CPU"):
        G(torch.zeros(2, 64, 3), torch.zeros(2, 64, 8))
