"""GPU: fp16-operand mode of the MFMA contractions (BASELINE configs[4] "fp16 MFMA MLPs"): every gemm_nt flavour against its
fp32 model at fp16-operand accuracy, and one full train step against the fp32 step."""
import numpy as np
import pytest
import torch

import kernel_model as km
from spgan import fixture_rng as fr
from oracle import spgan_oracle as orc
from test_kernels_gpu import close, ops, rnd  # noqa: F401  (ops is a fixture)
from test_parity_gpu import Opts, _load, sp   # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture()
def f16(ops):
    ops.set_mfma_operands("f16")
    yield ops
    ops.set_mfma_operands("f32")


@pytest.mark.parametrize("hint", [0, 2])
@pytest.mark.parametrize("M,N,K", [(2048, 256, 128), (4096, 128, 1280), (1024, 1024, 256), (700, 64, 64), (512, 40, 36)])
def test_gemm_nt_f16(f16, M, N, K, hint):
    """hint 2: the same checks through the 256 x 256-tile kernel (csrc/gemm_wide.hip, fp16-operand instantiation) where the shape is eligible"""
    if hint == 2 and (M % 256 or N % 256 or K % 32):
        pytest.skip("not a 256 x 256-tile shape")
    with f16.nt_tile_hint(hint):
        _gemm_nt_f16(f16, M, N, K)


def _gemm_nt_f16(ops, M, N, K):
    A, W, b = rnd("h.A%d" % K, (M, K)), rnd("h.W%d.%d" % (N, K), (N, K), 0.1), rnd("h.b%d" % N, (N,))
    ref = km.gemm_nt(A, W, b)
    close(ops.gemm_nt(A, W, b), ref, rtol=1e-3, what="plain")
    exact = km.gemm_nt(A.half().float(), W.half().float(), b)            # fp16-rounded operands, exact products, fp32 sums
    close(ops.gemm_nt(A, W, b), exact, rtol=2e-5, atol=1e-5, what="fp16-operand arithmetic")
    sc, sh = rnd("h.sc%d" % K, (K,)).abs() + 0.5, rnd("h.sh%d" % K, (K,), 0.3)
    y, m, v = ops.gemm_nt(A, W, b, pro=(sc, sh, 0.01), stats=True)
    y2, m2, v2 = km.gemm_nt(A, W, b, pro=(sc, sh, 0.01), stats=True)
    close(y, y2, rtol=1e-3, what="affine"); close(m, m2, rtol=1e-3, atol=1e-4); close(v, v2, rtol=2e-3)
    refm = rnd("h.ref%d.%d" % (M, N), (M, N))
    close(ops.gemm_nt_maskout(A, W, refm, 0.01), km.gemm_nt_maskout(A, W, refm, 0.01), rtol=1e-3, what="maskout")
    bsc, bsh, mu, inv = rnd("h.bsc%d" % N, (N,)), rnd("h.bsh%d" % N, (N,), 0.3), rnd("h.mu%d" % N, (N,), 0.2), rnd("h.inv%d" % N, (N,)).abs() + 0.5
    for a_, b_ in zip(ops.gemm_nt_bnbwd(A, W, refm, bsc, bsh, mu, inv, 0.01), km.gemm_nt_bnbwd(A, W, refm, bsc, bsh, mu, inv, 0.01)):
        close(a_, b_, rtol=2e-3, atol=2e-3, what="bnbwd")


@pytest.mark.parametrize("M,Na,Nb", [(16384, 256, 256), (65536, 128, 1280), (20480, 128, 64), (9000, 64, 32), (8192, 320, 64), (12288, 132, 36)])
def test_gemm_tn_bf16_operands(f16, M, Na, Nb):
    """Weight gradients in the "f16" operand mode: both operands rounded to bfloat16 at the LDS staging (after the fp32 prologue),
    fp32 accumulation / split partials / reduction.  Against the exact product of bf16-rounded operands (tight: only the fp32
    summation order differs) and against the fp32 model at bf16-operand accuracy; tiny per-point gradients (1e-7) must survive
    (fp16 would flush them); exact=True and short reductions keep fp32 operands."""
    ops = f16
    A, Bm = rnd("tb.A%d.%d" % (M, Na), (M, Na)) * 1e-7, rnd("tb.B%d.%d" % (M, Nb), (M, Nb))
    sc, sh = rnd("tb.sc%d" % Nb, (Nb,)).abs() + 0.5, rnd("tb.sh%d" % Nb, (Nb,), 0.3)
    r = lambda t: t.bfloat16().float()
    for pro in (None, (sc, sh, 0.01)):
        got = ops.gemm_tn(A, Bm, pro=pro)
        b = Bm if pro is None else torch.where(Bm * sc + sh > 0, Bm * sc + sh, (Bm * sc + sh) * 0.01)
        exact = (r(A).double().t() @ r(b).double()).float()
        close(got, exact, rtol=3e-5, atol=1e-12, what="bf16-operand arithmetic")
        close(got, km.gemm_tn(A, Bm, pro=pro), rtol=6e-3 * (1 + (pro is not None)), atol=1e-12, what="vs fp32 model")
        close(ops.gemm_tn(A, Bm, pro=pro, exact=True), km.gemm_tn(A, Bm, pro=pro), rtol=2e-5, atol=1e-12, what="exact=True keeps fp32 operands")
    out = rnd("tb.out%d.%d" % (Na, Nb), (Na, Nb)) * 1e-4
    want = 0.5 * out + (r(A).double().t() @ r(Bm).double()).float()
    close(ops.gemm_tn(A, Bm, out=out.clone(), beta=0.5), want, rtol=3e-5, atol=1e-12, what="beta accumulate")
    # by-products and operand modes of this round on the bf16 kernel: fp32 column sums of A, the lazy two-tensor A operand, A-side LeakyReLU
    o2, cs = ops.gemm_tn(A, Bm, with_colsum=True)
    close(o2, (r(A).double().t() @ r(Bm).double()).float(), rtol=3e-5, atol=1e-12, what="with_colsum: product"); close(cs, A.double().sum(0).float(), rtol=3e-6, atol=1e-10, what="with_colsum: fp32 column sums")
    if Na % 4 == 0:
        y = rnd("tb.y%d.%d" % (M, Na), (M, Na)) * 1e-7
        coef = torch.stack([rnd("tb.p%d" % Na, (Na,)).abs() + 0.5, rnd("tb.q%d" % Na, (Na,), 0.3), rnd("tb.r%d" % Na, (Na,), 1e-8)])
        lazy = ops.Affine2(A, y, coef)
        d = lazy.dense()
        close(ops.gemm_tn(lazy, Bm), (r(d).double().t() @ r(Bm).double()).float(), rtol=5e-5, atol=1e-12, what="lazy A operand")
        asc, ash = rnd("tb.asc%d" % Na, (Na,)).abs() + 0.5, rnd("tb.ash%d" % Na, (Na,), 1e-8)
        a_act = torch.where(A * asc + ash > 0, A * asc + ash, (A * asc + ash) * 0.01)
        close(ops.gemm_tn(A, Bm, a_pro=(asc, ash, 0.01)), (r(a_act).double().t() @ r(Bm).double()).float(), rtol=5e-5, atol=1e-12, what="A-side LeakyReLU")
    short = ops.gemm_tn(A[:4096], Bm[:4096])                                     # below TN_LP_MIN_ROWS: fp32 operands
    close(short, km.gemm_tn(A[:4096], Bm[:4096]), rtol=2e-5, atol=1e-12, what="short reduction stays fp32")


# ------------------------------------------------------------------ 16-bit storage of the EdgeBlock's GEMM-only tensors
@pytest.mark.parametrize("M,N,K", [(4096, 128, 1280), (2048, 64, 640), (700, 64, 64), (1024, 256, 256)])
def test_storage16_gemm_nt_half_operand(f16, M, N, K):
    """A float16-stored A operand (the EdgeBlock's T) goes to LDS as it lies in memory: bit-identical to the same values handed over
    as float32 (which the 128-row fp16 kernel rounds to the same halfs), bias / statistics epilogues included."""
    ops = f16
    A, W, b = rnd("s16.A%d" % K, (M, K)), rnd("s16.W%d.%d" % (N, K), (N, K), 0.1), rnd("s16.b%d" % N, (N,))
    Ah = A.half()
    with ops.nt_tile_hint(1):       # the float32-stored twin through the 128-row kernel too (the 256-wide one sums in another order)
        want = ops.gemm_nt(Ah.float(), W, b)
        want_s = ops.gemm_nt(Ah.float(), W, b, stats=True)
    assert torch.equal(ops.gemm_nt(Ah, W, b), want)
    for g, w in zip(ops.gemm_nt(Ah, W, b, stats=True), want_s):
        assert torch.equal(g, w)
    close(ops.gemm_nt(Ah, W, b), km.gemm_nt(Ah, W, b), rtol=1e-3, what="vs model")
    ops.set_mfma_operands("f32")
    with pytest.raises(ValueError):
        ops.gemm_nt(Ah, W, b)       # fp32 operand mode has no 16-bit storage
    ops.set_mfma_operands("f16")
    with pytest.raises(ValueError):
        ops.gemm_nt(Ah, W, b, pro=(rnd("s16.sc", (K,)), rnd("s16.sh", (K,)), 0.01))


@pytest.mark.parametrize("hint", [0, 2])
@pytest.mark.parametrize("M,N,K", [(4096, 1280, 128), (2048, 640, 64), (700, 640, 64), (65536, 1280, 128)])
def test_storage16_gemm_nt_bf16_result(f16, M, N, K, hint):
    """out_bf16=True: the float32 result of the same kernel rounded to bfloat16 (round to nearest even) at the store -- both the
    128-row and the 256 x 256-tile kernel (hint 2; (65536, 1280, 128) is G.EdgeConv2's dT and takes the wide kernel by itself)."""
    ops = f16
    if hint == 2 and (M % 256 or N % 256 or K % 32):
        pytest.skip("not a 256 x 256-tile shape")
    A, W = rnd("s16.dA%d.%d" % (M, K), (M, K)) * 1e-6, rnd("s16.dW%d.%d" % (N, K), (N, K), 0.1)     # gradient-sized values: bf16 keeps them, fp16 would not
    with ops.nt_tile_hint(hint):
        y32 = ops.gemm_nt(A, W)
        y16 = ops.gemm_nt(A, W, out_bf16=True)
    assert y16.dtype == torch.bfloat16 and torch.equal(y16, y32.bfloat16())
    assert float(y16.float().abs().max()) > 0
    with pytest.raises(ValueError):
        ops.gemm_nt(A, W, out_bf16=True, stats=True)
    ops.set_mfma_operands("f32")
    with pytest.raises(RuntimeError):
        ops.gemm_nt(A, W, out_bf16=True)      # refused by the library (SPGAN_EINVAL): only the fp16-operand kernels store bfloat16
    ops.set_mfma_operands("f16")


@pytest.mark.parametrize("M,Na,Nb", [(65536, 128, 1280), (16384, 64, 640), (9000, 64, 640)])
def test_storage16_gemm_tn_half_operand(f16, M, Na, Nb):
    """conv_out's weight gradient with T stored as float16: the same bfloat16 operand values as from the float32-stored T, so the
    same bits; the fp32 by-product (column sums of A) unchanged."""
    ops = f16
    A, Bm = rnd("s16.tA%d.%d" % (M, Na), (M, Na)) * 1e-6, rnd("s16.tB%d.%d" % (M, Nb), (M, Nb))
    Bh = Bm.half()
    assert torch.equal(ops.gemm_tn(A, Bh), ops.gemm_tn(A, Bh.float()))
    (c1, s1), (c2, s2) = ops.gemm_tn(A, Bh, with_colsum=True), ops.gemm_tn(A, Bh.float(), with_colsum=True)
    assert torch.equal(c1, c2) and torch.equal(s1, s2)
    close(c1, km.gemm_tn(A, Bh), rtol=8e-3, atol=1e-12, what="vs model")
    with pytest.raises(ValueError):
        ops.gemm_tn(A, Bh, exact=True)
    with pytest.raises(ValueError):
        ops.gemm_tn(A[:4096], Bh[:4096])        # short reductions keep the fp32 kernel, which has no float16 operand


@pytest.mark.parametrize("B,N,H,F_", [(2, 200, 32, 64), (2, 130, 64, 128), (1, 77, 16, 36)])
def test_storage16_edge_attend(f16, B, N, H, F_):
    """T written as float16 == the float32 T rounded to half (up to the last bit of the float32 value in front of the rounding: the
    two instantiations may contract their multiply-adds differently); the backward reading a bfloat16 dT == the float32 kernel on the
    same values."""
    ops = f16
    k, M = 10, B * N
    x = rnd("s16.x%d" % N, (M, 3))
    idx = ops.knn(x, B, N, k, mode=1)
    PQR = rnd("s16.PQR%d" % N, (M, H + 2 * F_))
    bx = rnd("s16.bx%d" % F_, (F_,), 0.1)
    h2 = rnd("s16.h2%d" % N, (M * k, F_))
    sc2, sh2 = rnd("s16.sc2%d" % F_, (F_,)).abs() + 0.5, rnd("s16.sh2%d" % F_, (F_,), 0.3)
    scx, shx = rnd("s16.scx%d" % F_, (F_,)).abs() + 0.5, rnd("s16.shx%d" % F_, (F_,), 0.3)
    T32 = ops.edge_attend_fwd(h2, sc2, sh2, PQR, idx, bx, scx, shx, 0.01)
    T16 = ops.edge_attend_fwd(h2, sc2, sh2, PQR, idx, bx, scx, shx, 0.01, half=True)
    assert T16.dtype == torch.float16
    close(T16.float(), T32, rtol=6e-4, atol=6e-8, what="float16 T")                  # half a unit in the last place of a half (2^-11), subnormal spacing
    assert float((T16 == T32.half()).float().mean()) > 0.99
    dT = (rnd("s16.dT%d" % N, (M, k * F_)) * 1e-6).bfloat16()
    m2, i2 = rnd("s16.m2%d" % F_, (F_,), 0.2), rnd("s16.i2%d" % F_, (F_,)).abs() + 0.5
    mx, ix = rnd("s16.mx%d" % F_, (F_,), 0.2), rnd("s16.ix%d" % F_, (F_,)).abs() + 0.5
    got = ops.edge_attend_bwd(dT, h2, sc2, sh2, m2, i2, PQR, idx, bx, scx, shx, mx, ix, 0.01)
    want = ops.edge_attend_bwd(dT.float(), h2, sc2, sh2, m2, i2, PQR, idx, bx, scx, shx, mx, ix, 0.01)
    assert got[0].dtype == torch.bfloat16 and got[1].dtype == torch.bfloat16     # g2 (a GEMM operand only) and gy (edge_scatter only) travel on as bfloat16
    for g, w, what in zip(got, want, ("g2", "gy", "sums2", "sumsy")):
        close(g.float(), w, rtol=3e-3 if what in ("g2", "gy") else 2e-5, atol=1e-12, what="bfloat16 dT: " + what)      # the sums come from the unrounded values
    assert float((got[0] == want[0].bfloat16()).float().mean()) > 0.99 and float((got[1] == want[1].bfloat16()).float().mean()) > 0.99
    # h2pre stored as float16 (written by the edge GEMM with out_half=True): the same kernels reading halfs
    h2h = h2.half()
    close(ops.edge_attend_fwd(h2h, sc2, sh2, PQR, idx, bx, scx, shx, 0.01, half=True).float(),
          ops.edge_attend_fwd(h2h.float(), sc2, sh2, PQR, idx, bx, scx, shx, 0.01, half=True).float(), rtol=1e-6, atol=6e-8, what="float16 h2pre: T")
    gh = ops.edge_attend_bwd(dT, h2h, sc2, sh2, m2, i2, PQR, idx, bx, scx, shx, mx, ix, 0.01)
    wh = ops.edge_attend_bwd(dT, h2h.float(), sc2, sh2, m2, i2, PQR, idx, bx, scx, shx, mx, ix, 0.01)
    for g, w, what in zip(gh, wh, ("g2", "gy", "sums2", "sumsy")):
        close(g.float(), w.float(), rtol=2e-5, atol=1e-12, what="float16 h2pre: " + what)
    with pytest.raises(ValueError):
        ops.edge_attend_fwd(h2h, sc2, sh2, PQR, idx, bx, scx, shx, 0.01)
    # edge_scatter on the bfloat16 gy == the float kernel on the same values
    H_ = H
    rowptr, src = ops.csr_build(idx, B, N)
    g1 = rnd("s16.g1%d" % N, (M * k, H_)) * 1e-6
    b1 = rnd("s16.b1%d" % H_, (H_,), 0.1)
    gam1, gamx = rnd("s16.gam1%d" % H_, (H_,)).abs() + 0.5, rnd("s16.gamx%d" % F_, (F_,)).abs() + 0.5
    m1, i1 = rnd("s16.m1%d" % H_, (H_,), 0.2), rnd("s16.i1%d" % H_, (H_,)).abs() + 0.5
    s1, sx = rnd("s16.s1%d" % H_, (2 * H_,), 1e-5), rnd("s16.sx%d" % F_, (2 * F_,), 1e-5)
    d16 = ops.edge_scatter(g1, got[1], PQR, idx, rowptr, src, b1, m1, i1, gam1, s1, bx, mx, ix, gamx, sx)
    d32 = ops.edge_scatter(g1, got[1].float(), PQR, idx, rowptr, src, b1, m1, i1, gam1, s1, bx, mx, ix, gamx, sx)
    close(d16, d32, rtol=1e-6, atol=1e-12, what="edge_scatter on bfloat16 gy")


@pytest.mark.parametrize("B,N,H,F_", [(4, 2048, 64, 128), (3, 700, 48, 96)])
def test_storage16_edge_gemms(f16, B, N, H, F_):
    """The three edge GEMMs of the EdgeBlock in 16-bit storage mode: h2pre written as float16 by the edge-operand product (statistics
    from the fp32 accumulators: unchanged), the lazy BatchNorm-backward operand p*g2 + q*h2pre + r with g2 bfloat16 / h2pre float16 in
    the weight-gradient (gemm_tn) and input-gradient (gemm_nt_bnbwd, edge epilogue) products: each against the same kernel fed the same
    values from float32 storage."""
    ops = f16
    k, M = 10, B * N
    E = M * k
    x = rnd("s16g.x%d" % N, (M, 3))
    idx = ops.knn(x, B, N, k, mode=1)
    Pm = rnd("s16g.P%d.%d" % (N, H), (M, H))
    b1 = rnd("s16g.b1%d" % H, (H,), 0.1)
    sc1, sh1 = rnd("s16g.sc1%d" % H, (H,)).abs() + 0.5, rnd("s16g.sh1%d" % H, (H,), 0.3)
    W2, b2 = rnd("s16g.W2%d.%d" % (F_, H), (F_, H), 0.2), rnd("s16g.b2%d" % F_, (F_,), 0.1)
    mk = lambda: (rnd("s16g.ga%d" % F_, (F_,)).abs() + 0.5, rnd("s16g.be%d" % F_, (F_,), 0.2), torch.zeros(F_, device=x.device), torch.ones(F_, device=x.device))
    bn_a, bn_b = mk(), mk()
    y32, st32 = ops.gemm_nt(Pm, W2, b2, pro=(sc1, sh1, 0.2), edge=(idx, b1), bn=bn_a)
    y16, st16 = ops.gemm_nt(Pm, W2, b2, pro=(sc1, sh1, 0.2), edge=(idx, b1), bn=bn_b, out_half=True)
    assert y16.dtype == torch.float16 and torch.equal(y16, y32.half())
    for a_, b_ in zip(st16 + (bn_b[2], bn_b[3]), st32 + (bn_a[2], bn_a[3])):
        assert torch.equal(a_, b_)
    # the lazy operand
    g2 = (rnd("s16g.g2%d.%d" % (N, F_), (E, F_)) * 1e-2).bfloat16()        # (the fp16 MFMA operand of the input-gradient product would flush 1e-6)
    coef = torch.stack([rnd("s16g.p%d" % F_, (F_,)).abs() + 0.5, rnd("s16g.q%d" % F_, (F_,), 1e-3), rnd("s16g.r%d" % F_, (F_,), 1e-4)])
    lazy16, lazy32 = ops.Affine2(g2, y16, coef), ops.Affine2(g2.float(), y16.float(), coef)
    with pytest.raises(ValueError):
        ops.Affine2(g2, y16.float(), coef)
    wg16 = ops.gemm_tn(lazy16, Pm, pro=(sc1, sh1, 0.2), edge=(idx, b1))
    wg32 = ops.gemm_tn(lazy32, Pm, pro=(sc1, sh1, 0.2), edge=(idx, b1))
    close(wg16, wg32, rtol=1e-6, atol=1e-14, what="weight gradient from 16-bit storage")
    close(wg16, km.gemm_tn(km.Affine2(g2, y16, coef), Pm, pro=(sc1, sh1, 0.2), edge=(idx, b1)), rtol=8e-3, atol=1e-12, what="weight gradient vs model")
    m1, i1 = rnd("s16g.m1%d" % H, (H,), 0.2), rnd("s16g.i1%d" % H, (H,)).abs() + 0.5
    r16 = ops.gemm_nt_bnbwd(lazy16, W2.t().contiguous(), Pm, sc1, sh1, m1, i1, 0.2, edge=(idx, b1))
    r32 = ops.gemm_nt_bnbwd(lazy32, W2.t().contiguous(), Pm, sc1, sh1, m1, i1, 0.2, edge=(idx, b1))
    for a_, b_, what in zip(r16, r32, ("g1", "s0", "s1")):
        close(a_, b_, rtol=1e-6, atol=1e-14, what="input gradient from 16-bit storage: " + what)
    rm = km.gemm_nt_bnbwd(km.Affine2(g2, y16, coef), W2.t().contiguous(), Pm, sc1, sh1, m1, i1, 0.2, edge=(idx, b1))
    for a_, b_, what in zip(r16, rm, ("g1", "s0", "s1")):
        close(a_, b_, rtol=3e-3, atol=1e-9, what="input gradient vs model: " + what)


def test_storage16_edgeblock_close_to_float32_storage(f16, sp):
    """The generator's second EdgeBlock (64 -> 128 channels) at a bench-like edge count, "f16" operand mode, with and without 16-bit
    storage of the per-edge tensors (h2pre, T: float16; dT, g2, gy: bfloat16): within fp16 / bfloat16 rounding of those tensors."""
    ops = f16
    B, N, k = 4, 2048, 10
    assert ops.storage16(B * N * k, 128, k) and not ops.storage16(B * N * k, 128, 20) and not ops.storage16(8192, 128, k) and not ops.storage16(B * N * k, 64, k)
    torch.manual_seed(5)
    blk = sp.EdgeBlock(64, 128, k).cuda().train()
    x0 = rnd("s16.xb", (B, 64, N))
    dout = rnd("s16.dout", (B, 128, N)) * 1e-4
    res = {}
    for on in (True, False):
        ops.STORAGE16[0] = on
        try:
            x = x0.clone().requires_grad_(True)
            blk.zero_grad(set_to_none=True)
            out = blk(x)
            out.backward(dout)
            res[on] = (out.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in blk.named_parameters()})
        finally:
            ops.STORAGE16[0] = True
    close(res[True][0], res[False][0], rtol=1e-3, what="out")
    close(res[True][1], res[False][1], rtol=8e-3, what="dx")
    for n, t in res[False][2].items():         # BatchNorm weight / bias gradients are sums over all edges with heavy cancellation: looser
        close(res[True][2][n], t, rtol=2e-2 if ".1." in n or ".4." in n else 8e-3, atol=1e-9 + 1e-4 * float(t.abs().max()), what=n)



def test_networks_f16_close_to_f32(sp):
    """fp16 MFMA operands against fp32: the stage in front of EdgeConv2's graph and the discriminator logits stay within fp16-
    operand accuracy; most feature-space kNN rows coincide (a flipped near-tie row changes the downstream features discretely,
    so the generated cloud itself is compared statistically, not element-wise); a WGAN-GP train step runs and stays finite."""
    B, N = 4, 512
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    z = fr.latent(B, N, seed=82).cuda()
    real = fr.synthetic_real(B, N, seed=81).cuda()
    out = {}
    for kind in ("f32", "f16"):
        sp.ops.set_mfma_operands(kind)
        try:
            G = _load(sp.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=8)).train()
            D = _load(sp.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=8)).train()
            with torch.no_grad():
                fake = G(x, z)
                logit = D(real.transpose(2, 1).contiguous())
            out[kind] = (G.last_x1.clone(), G.EdgeConv2.last_idx.clone(), fake.clone(), logit.clone())
            if kind == "f16":
                tr = sp.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0)
                info = tr.step(x, real, z, fr.latent(B, N, seed=83).cuda(), alpha=fr.uniform("f16.alpha", (B, 1, 1), 0.0, 1.0).cuda())
                assert torch.isfinite(info["loss_d"]).item() and torch.isfinite(info["loss_g"]).item()
                assert all(torch.isfinite(p).all().item() for p in list(G.parameters()) + list(D.parameters()))
        finally:
            sp.ops.set_mfma_operands("f32")
    a, b = out["f32"], out["f16"]
    assert ((a[0] - b[0]).norm() / a[0].norm()).item() < 3e-3                       # x1: EdgeConv1 + AdaIN1
    assert (a[1] == b[1]).all(dim=1).float().mean().item() > 0.9                    # EdgeConv2 kNN rows
    assert ((a[3] - b[3]).abs().max() / a[3].abs().max()).item() < 2e-2             # D logits
    assert abs(a[2].std().item() - b[2].std().item()) / a[2].std().item() < 0.1     # generated cloud: same scale


# ------------------------------------------------------------------------------------------------------------------
# fp16-operand mode against the REFERENCE (golden vectors) and the oracle -- not against our own fp32 path.
# Stated tolerances: operands are rounded to fp16 (relative 2^-11 = 4.9e-4 each) once per contraction, products and sums are
# exact/fp32, everything between contractions is fp32.  A K-term dot product of rounded operands has a relative error of about
# 4.9e-4 * sqrt(2/K) * (|a||w| / |a.w|); through 3-6 chained layers with BatchNorm re-normalising in between that leaves
# ~1e-3 on activations and ~1e-2 on gradients.  Measured values are in profiles/r02_parity.json; bounds are <= 3x those.
from helpers import check, golden, rel_l2   # noqa: E402


@pytest.fixture()
def f16sp(sp):
    sp.ops.set_mfma_operands("f16")
    yield sp
    sp.ops.set_mfma_operands("f32")


def test_f16_generator_stage_vs_reference_golden(f16sp):
    """G4 (captured from the reference): the sphere graph stays bit-exact (fp64 kNN, not an MFMA product), the stage in front of
    EdgeConv2's graph is within fp16-operand accuracy of the REFERENCE's fp32 values."""
    sp = f16sp
    d = golden("g4_generator.npz")
    B, N = 4, 256
    G = _load(sp.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=4)).train()
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    z = fr.latent(B, N, seed=44).cuda()
    out = G(x, z)
    i1 = sp.ops.idx_to_local64(G.EdgeConv1.last_idx, B, N).view(B, N, 10).cpu().numpy()
    assert np.array_equal(i1, d["idx1"]), "sphere graph must be bit-exact in fp16-operand mode too"
    check(d, "stage|x1", sp.ops.pm_to_cm(G.last_x1, B, N), rtol=3e-3, what="f16")
    i2 = sp.ops.idx_to_local64(G.EdgeConv2.last_idx, B, N).view(B, N, 10).cpu().numpy()
    assert (i2 == d["idx2"]).all(axis=2).mean() >= 0.9
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("B,N,stored16", [(4, 256, False), (4, 2048, True)], ids=["f32-storage", "16-bit-storage"])
def test_f16_generator_vs_oracle_with_injected_graph(f16sp, B, N, stored16):
    """The fp32 CPU oracle evaluated on the kNN graphs the fp16-operand run chose (tie-aware protocol): output and parameter
    gradients within fp16-operand accuracy -- at a size where EdgeConv2's per-edge tensors stay fp32 and at one where ops.storage16
    keeps them in 16 bits (81920 edges), same tolerances."""
    sp = f16sp
    assert sp.ops.storage16(B * N * 10, 128, 10) == stored16
    p = fr.init_params(orc.generator_shapes(), salt=31)
    G = _load(sp.Generator(Opts), p).train()
    x = fr.synthetic_real(B, N, seed=32); z = fr.latent(B, N, seed=33)
    out = G(x.cuda(), z.cuda())
    idx1 = sp.ops.idx_to_local64(G.EdgeConv1.last_idx, B, N).cpu()
    idx2 = sp.ops.idx_to_local64(G.EdgeConv2.last_idx, B, N).cpu()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref = orc.generator_forward(po, x, z, training=True, buffers=orc.bn_buffers(orc.generator_shapes()), idx1=idx1, idx2=idx2)
    assert rel_l2(out.detach().cpu().numpy(), ref.detach().numpy(), "f16|G out vs oracle (injected graphs)") <= 1e-2
    dy = fr.normal("pg.dy", out.shape)
    (out * dy.cuda()).sum().backward()
    names = [n for n in po if not n.endswith(("conv_w.0.bias", "conv_w.3.bias", "conv_x.0.bias", "global_conv.0.bias", "global_conv.3.bias"))]
    grads = torch.autograd.grad((ref * dy).sum(), [po[n] for n in names])
    gsd = dict(G.named_parameters())
    fa = torch.cat([gsd[n].grad.cpu().reshape(-1) for n in names]); fb = torch.cat([g.reshape(-1) for g in grads])
    cos = (torch.dot(fa, fb) / (fa.norm() * fb.norm())).item()
    rel = rel_l2(fa.numpy(), fb.numpy(), "f16|G all parameter gradients vs oracle")
    # whole-network gradients are kink-limited already in fp32 (a handful of LeakyReLU / arg-max flips move them by
    # 1e-2); operands perturbed by 5e-4 flip more of them -- measured cosine 0.992 (fp32: 0.9999)
    assert cos >= 0.985, cos
    assert rel <= 0.2, rel


def test_f16_discriminator_vs_reference_golden(f16sp):
    """G5 (captured from the reference): logits, the input gradient (the WGAN-GP route) and the parameter gradients."""
    sp = f16sp
    d = golden("g5_discriminator.npz")
    B, N = 4, 256
    D = _load(sp.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=4)).train()
    real = fr.synthetic_real(B, N, seed=5).transpose(2, 1).contiguous().cuda().requires_grad_(True)
    logit = D(real)
    check(d, "logit", logit, rtol=5e-3, what="f16")
    ((logit - 1.0) ** 2).mean().backward()
    # The input gradient flows through the global max-pool: 1024 channels x B arg-max choices among 256 points.  An operand
    # perturbation of 5e-4 moves a few near-tied arg-max rows to another point, which moves the gradient of those points
    # discretely -- the fp16-operand input gradient is compared at that granularity (measured 8e-2; fp32: 1e-4).
    check(d, "dx", real.grad, rtol=0.25, what="f16")
    gref = torch.from_numpy(d["dx|full"]).reshape(-1) if "dx|full" in d else None
    for n, p in D.named_parameters():
        if not n.endswith(("mlps.0.bias", "mlps.3.bias", "mlps.6.bias", "fc2.0.bias")):
            check(d, "grad|" + n, p.grad, rtol=0.25, atol=1e-6, what="f16")
    # the gradient penalty against golden G7 (double backward through fp16-operand contractions)
    d7 = golden("g7_gradient_penalty.npz")
    D7 = _load(sp.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=7)).train()
    B7 = 3
    real7 = fr.synthetic_real(B7, N, seed=71).transpose(2, 1).contiguous().cuda()
    fake7 = (0.8 * fr.synthetic_real(B7, N, seed=72) + 0.05 * fr.normal("g7.n", (B7, N, 3))).transpose(2, 1).contiguous().cuda()
    gp = sp.GradientPenalty(10.0, gamma=1)(D7, real7, fake7, alpha=torch.from_numpy(d7["alpha"]).cuda())
    np.testing.assert_allclose(gp.item(), float(d7["gp"]), rtol=2e-2)


def test_f16_full_size_config_properties(f16sp):
    """The per-GPU shape of BASELINE configs[4] / [1] -- batch 32, N = 2048, WGAN-GP -- in fp16-operand mode: size-independent
    properties.  The sphere graph is bit-exact against the oracle's sort, two runs are bit-identical (no atomics, fixed-order
    reductions), everything stays finite, the generated cloud is inside tanh's range and one Adam step moves a parameter by
    at most lr."""
    sp = f16sp
    B, N = 32, 2048
    x = fr.sphere_template(N)[None].repeat(B, 1, 1).cuda()
    real = fr.synthetic_real(B, N, seed=1).cuda()
    zd, zg = fr.latent(B, N, seed=2)[:, :1].contiguous().cuda(), fr.latent(B, N, seed=3)[:, :1].contiguous().cuda()
    alpha = fr.uniform("f16.full.alpha", (B, 1, 1), 0.0, 1.0).cuda()

    class O2048(Opts):
        np = 2048
    runs = []
    for r in range(2):
        G = _load(sp.Generator(O2048), fr.init_params(orc.generator_shapes(), salt=8))
        D = _load(sp.Discriminator(O2048), fr.init_params(orc.discriminator_shapes(), salt=8))
        p0 = torch.cat([p.detach().reshape(-1).clone() for p in list(G.parameters()) + list(D.parameters())])
        tr = sp.TrainStep(G, D, gan="wgan", use_gp=True, lambda_gp=10.0, lr_g=1e-4, lr_d=1e-4)
        info = tr.step(x, real, zd, zg, alpha=alpha, keep_grads=True)
        torch.cuda.synchronize()
        p1 = torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())])
        runs.append((info["loss_d"].item(), info["loss_g"].item(), p1.clone(), info["fake_g"].clone()))
        assert torch.isfinite(p1).all() and np.isfinite(runs[-1][0]) and np.isfinite(runs[-1][1])
        assert (p1 - p0).abs().max().item() <= 1.01e-4             # |Adam step| <= lr, plus the rounding of p itself
        assert info["fake_g"].abs().max().item() <= 1.0
        if r == 0:
            own = orc.knn_sorted(fr.sphere_template(N)[None].transpose(2, 1).contiguous(), 10).reshape(1, -1)
            got = sp.ops.idx_to_local64(G.EdgeConv1.last_idx[:N].contiguous(), 1, N).cpu()
            assert torch.equal(own, got), "sphere graph differs from the oracle at N=2048"
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1]
    assert torch.equal(runs[0][2], runs[1][2]) and torch.equal(runs[0][3], runs[1][3]), "fp16-operand step is not deterministic"


# ------------------------------------------------------------------------------------------------------------------
# "bf16x3": fp32 operands split exactly into three bf16 terms, six cross products on the bf16 matrix pipe, fp32 accumulation
@pytest.fixture()
def b3(ops):
    ops.set_mfma_operands("bf16x3")
    yield ops
    ops.set_mfma_operands("f32")


@pytest.mark.parametrize("M,N,K", [(2048, 256, 128), (4096, 128, 1280), (1024, 1024, 256), (700, 64, 64), (8192, 256, 3 * 32 + 8)])
def test_gemm_nt_bf16x3_is_fp32_equivalent(b3, M, N, K):
    """Against the float64 product of the SAME fp32 operands: the split-bf16 result is as close as the exact-fp32-MFMA result
    (both are limited by fp32 accumulation), i.e. the dropped cross terms (<= 3*2^-24 per product) do not show."""
    ops = b3
    A, W, b = rnd("b3.A%d" % K, (M, K)), rnd("b3.W%d.%d" % (N, K), (N, K), 0.1), rnd("b3.b%d" % N, (N,))
    ref = (A.double() @ W.double().t() + b.double())
    got = ops.gemm_nt(A, W, b)
    ops.set_mfma_operands("f32")
    f32 = ops.gemm_nt(A, W, b)
    ops.set_mfma_operands("bf16x3")
    e3 = ((got.double() - ref).norm() / ref.norm()).item()
    e32 = ((f32.double() - ref).norm() / ref.norm()).item()
    m3 = (got.double() - ref).abs().max().item() / ref.abs().max().item()
    assert e3 <= max(3.0 * e32, 3e-7), (e3, e32)
    assert e3 <= 1e-6 and m3 <= 3e-6, (e3, m3)               # fp32 accumulation over K terms: ~sqrt(K) * 2^-24
    # the fused prologue / epilogues run unchanged in front of / behind the split
    sc, sh = rnd("b3.sc%d" % K, (K,)).abs() + 0.5, rnd("b3.sh%d" % K, (K,), 0.3)
    y, m, v = ops.gemm_nt(A, W, b, pro=(sc, sh, 0.01), stats=True)
    y2, m2, v2 = km.gemm_nt(A, W, b, pro=(sc, sh, 0.01), stats=True)
    close(y, y2, rtol=2e-6, atol=2e-6, what="affine"); close(m, m2, rtol=1e-5, atol=1e-6); close(v, v2, rtol=2e-5)
    refm = rnd("b3.ref%d.%d" % (M, N), (M, N))
    close(ops.gemm_nt_maskout(A, W, refm, 0.01), km.gemm_nt_maskout(A, W, refm, 0.01), rtol=2e-6, atol=2e-6, what="maskout")
    # extreme magnitudes: the split keeps fp32's range (bf16 has fp32's exponent)
    big = ops.gemm_nt(A * 1e18, W * 1e15)
    assert torch.isfinite(big).all()
    close(big, (A.double() * 1e18) @ (W.double() * 1e15).t(), rtol=1e-6, what="large magnitudes")
    tiny = ops.gemm_nt(A * 1e-18, W * 1e-15)
    close(tiny, (A.double() * 1e-18) @ (W.double() * 1e-15).t(), rtol=1e-6, atol=0.0, what="small magnitudes")


def test_networks_bf16x3_match_reference_goldens_at_fp32_tolerances(sp):
    """The split mode against the REFERENCE goldens at the SAME tolerances as the exact-fp32 path (test_parity_gpu.py): G4 stage,
    G5 discriminator forward/backward, G7 gradient penalty."""
    sp.ops.set_mfma_operands("bf16x3")
    try:
        d = golden("g4_generator.npz")
        B, N = 4, 256
        G = _load(sp.Generator(Opts), fr.init_params(orc.generator_shapes(), salt=4)).train()
        G(fr.sphere_template(N)[None].repeat(B, 1, 1).cuda(), fr.latent(B, N, seed=44).cuda())
        check(d, "stage|x1", sp.ops.pm_to_cm(G.last_x1, B, N), rtol=2e-5, what="bf16x3")
        d = golden("g5_discriminator.npz")
        D = _load(sp.Discriminator(Opts), fr.init_params(orc.discriminator_shapes(), salt=4)).train()
        real = fr.synthetic_real(B, N, seed=5).transpose(2, 1).contiguous().cuda().requires_grad_(True)
        logit = D(real)
        check(d, "logit", logit, rtol=1e-5, what="bf16x3")
        ((logit - 1.0) ** 2).mean().backward()
        check(d, "dx", real.grad, rtol=2e-4, what="bf16x3")
        for n, p in D.named_parameters():
            check(d, "grad|" + n, p.grad, rtol=3e-4, atol=2e-3 if n.endswith(("mlps.0.bias", "mlps.3.bias", "mlps.6.bias", "fc2.0.bias")) else 1e-7, what="bf16x3")
    finally:
        sp.ops.set_mfma_operands("f32")


# ------------------------------------------------------------------------------------------------------------------
# split-bf16 operands on 256-row tiles (csrc/gemm_wide3.hip): configurations 24 (256 x 256 tiles), 22 (256 x 128), 14 (128 x 256, with W's
# pre-split image) -- chosen by the library from the shape (spgan_nt_wide3_config); shapes below reach each of them
WIDE3_SHAPES = [   # M, N, K, what it exercises
    (16384, 1024, 256, "24: K > 128, tiles fill the chip"),
    (1024, 128, 1280, "22: N % 256 != 0, long K"),
    (2048, 256, 128, "22 / 14 with image: short K"),
    (512, 384, 64, "22: N = 3 x 128, two k-pairs"),
    (768, 512, 32, "the shortest k-loop (one pair of k-tiles)"),
]


def _images(ops):
    """A provider without a cache: every call splits W anew (a cache keyed by address would hand a recycled temporary's image to its successor --
    the staleness rules are the real provider's business, nets.w_image)."""
    return lambda Wt: ops.split_image(Wt)


@pytest.mark.parametrize("image", [False, True])
@pytest.mark.parametrize("M,N,K,what", WIDE3_SHAPES)
def test_gemm_nt_wide3(b3, M, N, K, what, image):
    """Every prologue / epilogue of the 256-row-tile split-bf16 kernel against the float64 product and the fp32 models; with and without the
    pre-split image of W (ops.split_image: the kernel copies W's planes instead of splitting its rows)."""
    ops = b3
    ops.w_image_provider = _images(ops) if image else None
    try:
        with ops.nt_tile_hint(2):
            A, W, b = rnd("w3.A%d.%d" % (M, K), (M, K)), rnd("w3.W%d.%d" % (N, K), (N, K), 0.1), rnd("w3.b%d" % N, (N,))
            ref = A.double() @ W.double().t() + b.double()
            got = ops.gemm_nt(A, W, b)
            e3 = ((got.double() - ref).norm() / ref.norm()).item()
            m3 = (got.double() - ref).abs().max().item() / ref.abs().max().item()
            assert e3 <= 1e-6 and m3 <= 3e-6, (e3, m3)
            with ops.nt_tile_hint(1):
                narrow = ops.gemm_nt(A, W, b)          # the 128-row split kernel: same products, another summation order
            close(got, narrow, rtol=2e-6, atol=2e-6, what="wide vs 128-row split kernel")
            # activations in the epilogue, a dense / a per-group row addend
            close(ops.gemm_nt(A, W, b, act=km.ACT_LRELU, slope=0.2), km.gemm_nt(A, W, b, act=km.ACT_LRELU, slope=0.2), rtol=2e-6, atol=2e-6, what="lrelu")
            close(ops.gemm_nt(A, W, b, act=km.ACT_TANH), km.gemm_nt(A, W, b, act=km.ACT_TANH), rtol=2e-6, atol=2e-6, what="tanh")
            rows = 256
            rbg = rnd("w3.rbg%d.%d" % (M, N), (M // rows, N))
            close(ops.gemm_nt(A, W, b, rowbias=rbg, rows_per_group=rows), km.gemm_nt(A, W, b, rowbias=rbg, rows_per_group=rows), rtol=2e-6, atol=2e-6, what="group bias")
            # BatchNorm + LeakyReLU prologue (one vector pair, and one pair per group of rows), column statistics
            sc, sh = rnd("w3.sc%d" % K, (K,)).abs() + 0.5, rnd("w3.sh%d" % K, (K,), 0.3)
            for slope in (0.01, 0.2, 1.0, 0.0):
                y, m, v = ops.gemm_nt(A, W, b, pro=(sc, sh, slope), stats=True)
                y2, m2, v2 = km.gemm_nt(A, W, b, pro=(sc, sh, slope), stats=True)
                close(y, y2, rtol=2e-6, atol=2e-6, what="affine %g" % slope); close(m, m2, rtol=1e-5, atol=1e-6); close(v, v2, rtol=2e-5)
            # a LeakyReLU slope outside [0, 1] (max(v, v*slope) is not the activation there) leaves this kernel for the 128-row one: same answer
            y = ops.gemm_nt(A, W, b, pro=(sc, sh, 1.5))
            close(y, km.gemm_nt(A, W, b, pro=(sc, sh, 1.5)), rtol=2e-6, atol=2e-6, what="slope 1.5 (fallback)")
            # epilogues of the backward passes
            refm = rnd("w3.ref%d.%d" % (M, N), (M, N))
            close(ops.gemm_nt_maskout(A, W, refm, 0.01), km.gemm_nt_maskout(A, W, refm, 0.01), rtol=2e-6, atol=2e-6, what="maskout")
            bsc, bsh, mu, inv = rnd("w3.bsc%d" % N, (N,)), rnd("w3.bsh%d" % N, (N,), 0.3), rnd("w3.mu%d" % N, (N,), 0.2), rnd("w3.inv%d" % N, (N,)).abs() + 0.5
            radd = rnd("w3.radd%d.%d" % (M, N), (M, N))
            for pro in (None, (sc, sh, 0.2)):
                for a_, b_ in zip(ops.gemm_nt_bnbwd(A, W, refm, bsc, bsh, mu, inv, 0.01, pro=pro, bias=b, rowadd=radd),
                                  km.gemm_nt_bnbwd(A, W, refm, bsc, bsh, mu, inv, 0.01, pro=pro, bias=b, rowadd=radd)):
                    close(a_, b_, rtol=5e-6, atol=2e-5, what="bnbwd")
            # BatchNorm + LeakyReLU + max-pool behind the product (the output is not stored)
            if M % 256 == 0 and M >= 512:
                gamma, beta = rnd("w3.g%d" % N, (N,)), rnd("w3.be%d" % N, (N,), 0.3)        # both signs: max and min records are read
                for keep in (False, True):
                    o = ops.gemm_bn_pool(A, W, b, (gamma, beta, None, None), 256, 0.2, pro=(sc, sh, 0.2), keep_y=keep)
                    o2 = km.gemm_bn_pool(A, W, b, (gamma, beta, None, None), 256, 0.2, pro=(sc, sh, 0.2), keep_y=keep)
                    close(o[2], o2[2], rtol=3e-5, atol=3e-5, what="pooled")
                    for x_, y_ in zip(o[1], o2[1]):
                        close(x_, y_, rtol=3e-5, atol=3e-6, what="bn vectors")
                    same = (o[3] == o2[3]).float().mean().item()
                    assert same >= 0.99, "arg-max rows differ on %.2f %% of the (shape, channel) pairs" % (100 * (1 - same))
            # extreme magnitudes: the split keeps fp32's range
            big = ops.gemm_nt(A * 1e18, W * 1e15)
            assert torch.isfinite(big).all()
            close(big, (A.double() * 1e18) @ (W.double() * 1e15).t(), rtol=1e-6, what="large magnitudes")
            close(ops.gemm_nt(A * 1e-18, W * 1e-15), (A.double() * 1e-18) @ (W.double() * 1e-15).t(), rtol=1e-6, atol=0.0, what="small magnitudes")
    finally:
        ops.w_image_provider = None


def test_gemm_nt_wide3_product_error_bound(b3):
    """The dropped cross terms: every single product a*b is reproduced within 4 * 2^-24 |a*b| (the review's bound; the round-to-nearest split
    keeps it at ~2^-26) -- on operands chosen to make the dropped terms as large as they can be (all three planes full: mantissas 0xffffff-like)
    and a reduction in which nothing cancels (one non-zero term per output), over 40 binades of magnitude."""
    ops = b3
    M, N, K = 512, 256, 64
    g = torch.Generator().manual_seed(7)
    mant = lambda shape: 1.0 + (torch.randint(0, 2 ** 23, shape, generator=g).double() / 2 ** 23)      # [1, 2): random 24-bit significands
    full = 2.0 - 2.0 ** -23                                                                             # 0x3fffffff: every plane saturated
    A = torch.zeros(M, K, dtype=torch.float64); W = torch.zeros(N, K, dtype=torch.float64)
    k_of_m = torch.arange(M) % K
    A[torch.arange(M), k_of_m] = mant((M,)) * (2.0 ** torch.randint(-20, 20, (M,), generator=g).double())
    A[::7, :] = 0; A[torch.arange(0, M, 7), k_of_m[::7]] = full
    W[:] = mant((N, K)) * (2.0 ** torch.randint(-20, 20, (N, K), generator=g).double())
    W[::5] = full * (2.0 ** torch.randint(-20, 20, (W[::5].shape[0], 1), generator=g).double())
    Af, Wf = A.float().cuda(), W.float().cuda()
    assert torch.equal(Af.double().cpu(), A) and torch.equal(Wf.double().cpu(), W)                      # exactly representable
    ref = (Af.double() @ Wf.double().t())                                                                # one term per output: the exact product
    for hint in (2, 1):
        with ops.nt_tile_hint(hint):
            got = ops.gemm_nt(Af, Wf)
        rel = ((got.double() - ref).abs() / ref.abs()).max().item()
        assert rel <= 4 * 2.0 ** -24, (hint, rel, rel / 2.0 ** -24)


def test_split_image_layout(b3):
    """ops.split_image: hi + mid + lo == +-W exactly, in the image layout [k/16][plane][n][16] with the 16-byte halves swapped where bit 3 of n is set and
    the rows of odd 32-row tiles negated."""
    ops = b3
    N, K = 256, 96
    W = rnd("w3.img", (N, K)) * torch.logspace(-12, 12, N, device="cuda")[:, None]
    img = ops.split_image(W).view(torch.bfloat16).view(K // 16, 3, N, 2, 8).float().double()
    n = torch.arange(N, device="cuda")
    swap = ((n >> 3) & 1).bool()
    img = torch.where(swap[None, None, :, None, None], img.flip(3), img)                 # undo the half swap
    img = torch.where(((n >> 5) & 1).bool()[None, None, :, None, None], -img, img)       # ... and the sign checkerboard (odd 32-row tiles are stored negated)
    planes = img.reshape(K // 16, 3, N, 16).permute(1, 2, 0, 3).reshape(3, N, K)
    assert torch.equal(planes.sum(0), W.double()), "hi + mid + lo != W"
    assert torch.equal(ops.split_image(W), km.split_image(W)), "image differs from its model (tests/kernel_model.py::split_image)"
    assert (planes[1].abs() <= planes[0].abs() * 2.0 ** -8 + 1e-300).all() and (planes[2].abs() <= planes[0].abs() * 2.0 ** -16 + 1e-300).all()


@pytest.mark.parametrize("M,Na,Nb", [(16384, 256, 256), (65536, 128, 1280), (65536, 256, 128), (32768, 128, 128), (20480, 128, 64), (9000, 64, 32), (8192, 320, 64), (12288, 132, 36)])
def test_gemm_tn_bf16x3_is_fp32_equivalent(b3, M, Na, Nb):
    """Weight gradients in the split-bf16 mode (spgan_gemm_tn_args.mfma_lp == 2; csrc/gemm_tn_wide3.hip for the shapes whose output tiles as
    256 x 256 / 128 x 256 / 256 x 128 -- the first three here --, the exact-fp32 kernel on the split plan otherwise): against the float64 product of
    the SAME fp32 operands the result is as close as the exact-fp32-MFMA one; per-point gradients of magnitude 1e-7 survive; every A-side /
    B-side operand mode and by-product; short reductions and exact=True keep fp32 operands."""
    ops = b3
    was = ops.TN_SPLIT_BF16[0]
    ops.TN_SPLIT_BF16[0] = True
    try:
        _gemm_tn_bf16x3(ops, M, Na, Nb)
    finally:
        ops.TN_SPLIT_BF16[0] = was


def _gemm_tn_bf16x3(ops, M, Na, Nb):
    A, Bm = rnd("t3.A%d.%d" % (M, Na), (M, Na)) * 1e-7, rnd("t3.B%d.%d" % (M, Nb), (M, Nb))
    sc, sh = rnd("t3.sc%d" % Nb, (Nb,)).abs() + 0.5, rnd("t3.sh%d" % Nb, (Nb,), 0.3)

    def err(x, ref):
        return ((x.double() - ref).norm() / ref.norm()).item()
    for pro in (None, (sc, sh, 0.01)):
        b = Bm if pro is None else torch.where(Bm * sc + sh > 0, Bm * sc + sh, (Bm * sc + sh) * 0.01)
        ref = A.double().t() @ b.double()
        got = ops.gemm_tn(A, Bm, pro=pro)
        e3, e32 = err(got, ref), err(ops.gemm_tn(A, Bm, pro=pro, exact=True), ref)
        assert e3 <= max(3.0 * e32, 3e-7) and e3 <= 2e-6, (e3, e32)
        close(got, km.gemm_tn(A, Bm, pro=pro), rtol=2e-5, atol=1e-12, what="vs fp32 model")
    out = rnd("t3.out%d.%d" % (Na, Nb), (Na, Nb)) * 1e-4
    close(ops.gemm_tn(A, Bm, out=out.clone(), beta=0.5), (0.5 * out.double() + A.double().t() @ Bm.double()).float(), rtol=2e-6, atol=1e-12, what="beta accumulate")
    o2, cs = ops.gemm_tn(A, Bm, with_colsum=True)
    close(o2, (A.double().t() @ Bm.double()).float(), rtol=2e-6, atol=1e-12, what="with_colsum: product")
    close(cs, A.double().sum(0).float(), rtol=3e-6, atol=1e-10, what="with_colsum: fp32 column sums (the sign of odd splits undone)")
    if Na % 4 == 0:
        y = rnd("t3.y%d.%d" % (M, Na), (M, Na)) * 1e-7
        coef = torch.stack([rnd("t3.p%d" % Na, (Na,)).abs() + 0.5, rnd("t3.q%d" % Na, (Na,), 0.3), rnd("t3.r%d" % Na, (Na,), 1e-8)])
        lazy = ops.Affine2(A, y, coef)
        close(ops.gemm_tn(lazy, Bm), (lazy.dense().double().t() @ Bm.double()).float(), rtol=3e-6, atol=1e-12, what="lazy A operand (sign folded into the coefficients)")
        asc, ash = rnd("t3.asc%d" % Na, (Na,)).abs() + 0.5, rnd("t3.ash%d" % Na, (Na,), 1e-8)
        a_act = torch.where(A * asc + ash > 0, A * asc + ash, (A * asc + ash) * 0.01)
        close(ops.gemm_tn(A, Bm, a_pro=(asc, ash, 0.01)), (a_act.double().t() @ Bm.double()).float(), rtol=3e-6, atol=1e-12, what="A-side LeakyReLU (sign by a multiply)")
    close(ops.gemm_tn(A[:4096], Bm[:4096]), km.gemm_tn(A[:4096], Bm[:4096]), rtol=2e-5, atol=1e-12, what="short reduction stays fp32")
    big = ops.gemm_tn(A * 1e25, Bm * 1e10)
    assert torch.isfinite(big).all()
    close(big, ((A.double() * 1e25).t() @ (Bm.double() * 1e10)).float(), rtol=2e-6, what="large magnitudes")
