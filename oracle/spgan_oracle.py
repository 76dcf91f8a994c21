"""CPU oracle for the SP-GAN G+D train-step hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional restatement (plain PyTorch fp32 on the
CPU) of the arithmetic that the reference performs on its train path.  It is the
*checker* for the HIP kernels; the product (`sp-gan_amd/spgan`) never imports it.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the real reference
from /root/reference (build container only), drives it with the deterministic
weights/inputs of `spgan.fixture_rng`, and stores the outputs under
`tests/golden/*.npz`.  `tests/test_oracle_golden.py` checks every function below
against those vectors.

Everything is written against a flat ``params`` dict that uses the reference's
own ``state_dict`` keys, so a reference checkpoint can be fed in unchanged.
Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

NEG = 0.01      # Generator.py:22 / Discriminator.py:19  (LeakyReLU slope inside the MLPs)
NEG_2 = 0.2     # Generator.py:23  (LeakyReLU before each AdaIN)
BN_EPS = 1e-5   # torch.nn.BatchNorm*/InstanceNorm1d default
BN_MOM = 0.1


# --------------------------------------------------------------------------- #
# graph ops                                                                   #
# --------------------------------------------------------------------------- #
def pairwise_sqdist(x: Tensor) -> Tensor:
    """dist[b,i,j] = -2 x_i.x_j + |x_i|^2 + |x_j|^2, x: [B,C,N].
    Generation/modules.py:695-699 (expanded form, evaluated in this order)."""
    xt = x.permute(0, 2, 1)
    inner = -2 * torch.bmm(xt, x)
    sq = torch.sum(xt ** 2, dim=2, keepdim=True)
    return inner + sq + sq.permute(0, 2, 1)


def knn_sorted(x: Tensor, k: int) -> Tensor:
    """Ranks 1..k of the full ascending sort of each distance row -> int64 [B,N,k].
    Rank 0 is dropped *positionally* (modules.py:702-703), not by "exclude self"."""
    order = torch.sort(pairwise_sqdist(x), dim=2)[1]
    return order[:, :, 1:k + 1].contiguous()


def knn_sorted_fp64_direct(x: Tensor, k: int) -> Tensor:
    """Same selection with distances evaluated as sum((x_i-x_j)^2) in fp64 from the
    fp32-rounded coordinates, ties broken by the lower index.  This is what the
    HIP kernel does for coordinate-space inputs (C<=4); on every sphere template
    it reproduces `knn_sorted` row for row (checked in tests)."""
    xd = x.double().permute(0, 2, 1)
    d = ((xd[:, :, None, :] - xd[:, None, :, :]) ** 2).sum(-1)
    order = torch.sort(d, dim=2, stable=True)[1]
    return order[:, :, 1:k + 1].contiguous()


def gather_neighbors(x: Tensor, idx: Tensor) -> Tensor:
    """x [B,C,N], idx [B,N,k] -> [B,C,N,k] (modules.py:708-714)."""
    B, C, N = x.shape
    k = idx.shape[2]
    flat = idx.reshape(B, 1, N * k).expand(B, C, N * k)
    return torch.gather(x, 2, flat).view(B, C, N, k)


def get_edge_features(x: Tensor, k: int, idx: Optional[Tensor] = None, return_idx: bool = False):
    """[B,C,N] -> ee [B,2C,N,k] = cat[central, neighbour-central] (modules.py:683-725).
    `idx` may be injected as int64 [B,N*k] or [B,N,k]."""
    B, C, N = x.shape
    if idx is None:
        idx = knn_sorted(x, k)
    idx3 = idx.view(B, N, k)
    nb = gather_neighbors(x, idx3)
    central = x.unsqueeze(3).expand(B, C, N, k)
    ee = torch.cat([central, nb - central], dim=1)
    if return_idx:
        return ee, idx3.reshape(B, N * k)
    return ee


# --------------------------------------------------------------------------- #
# norm helpers (train-mode statistics + running-stat side effects)            #
# --------------------------------------------------------------------------- #
def _bn(x: Tensor, p: Dict[str, Tensor], prefix: str, training: bool, buffers: Optional[Dict[str, Tensor]]):
    """BatchNorm over every dim but 1.  Train mode: biased batch var for the
    normalisation, unbiased var into running_var, momentum 0.1 (torch defaults;
    Generator.py:58,61,67,121,124; Discriminator.py:57,60,63,79)."""
    w, b = p[prefix + ".weight"], p[prefix + ".bias"]
    if buffers is None:
        rm = rv = None
    else:
        rm, rv = buffers[prefix + ".running_mean"], buffers[prefix + ".running_var"]
        if training and (prefix + ".num_batches_tracked") in buffers:
            buffers[prefix + ".num_batches_tracked"] += 1
    if not training and rm is None:
        raise ValueError("eval-mode batch norm needs running statistics")
    return F.batch_norm(x, rm, rv, w, b, training, BN_MOM, BN_EPS)


def _conv1x1(x: Tensor, p, prefix: str) -> Tensor:
    """1x1 Conv1d/Conv2d as a channel contraction."""
    w = p[prefix + ".weight"]
    b = p[prefix + ".bias"]
    w2 = w.reshape(w.shape[0], w.shape[1])
    y = torch.einsum("oc,bc...->bo...", w2, x)
    return y + b.view(1, -1, *([1] * (x.dim() - 2)))


# --------------------------------------------------------------------------- #
# Generator pieces                                                            #
# --------------------------------------------------------------------------- #
def edge_block(p, prefix: str, x: Tensor, k: int, idx: Optional[Tensor] = None,
               training: bool = True, buffers=None, return_idx: bool = False):
    """EdgeBlock.forward, Generation/Generator.py:75-88.  x [B,Fin,N] -> [B,Fout,N]."""
    B, C, N = x.shape
    ee, idx_used = get_edge_features(x, k, idx=idx, return_idx=True)
    # conv_w on the difference half (Generator.py:56-63,78)
    w = _conv1x1(ee[:, C:], p, prefix + ".conv_w.0")
    w = F.leaky_relu(_bn(w, p, prefix + ".conv_w.1", training, buffers), NEG)
    w = _conv1x1(w, p, prefix + ".conv_w.3")
    w = F.leaky_relu(_bn(w, p, prefix + ".conv_w.4", training, buffers), NEG)
    w = F.softmax(w, dim=-1)                                   # Generator.py:79
    # conv_x on the full edge feature (Generator.py:65-69,81)
    y = _conv1x1(ee, p, prefix + ".conv_x.0")
    y = F.leaky_relu(_bn(y, p, prefix + ".conv_x.1", training, buffers), NEG)
    y = y * w                                                  # Generator.py:82
    # conv_out: [1,k] kernel == contraction over (channel, neighbour rank) (Generator.py:71,84)
    wo = p[prefix + ".conv_out.weight"]                        # [F,F,1,k]
    out = torch.einsum("ocr,bcnr->bon", wo[:, :, 0, :], y) + p[prefix + ".conv_out.bias"].view(1, -1, 1)
    if return_idx:
        return out, idx_used
    return out


def adaptive_point_norm(p, prefix: str, x: Tensor, style: Tensor) -> Tensor:
    """AdaptivePointNorm.forward, Generator.py:38-45: per-point gamma/beta from a
    1x1 conv of the style, applied to the instance-normalised input."""
    s = _conv1x1(style, p, prefix + ".style")
    gamma, beta = s.chunk(2, 1)
    xhat = F.instance_norm(x, eps=BN_EPS)                       # affine=False, biased var over N
    return gamma * xhat + beta


def attention(p, prefix: str, x: Tensor) -> Tensor:
    """Attention.forward, Generation/modules.py:549-558 (--attn, Generator.py:116-117,191-192): x [B,C,N] ->
    gamma * o(g . softmax_j(theta^T phi)^T) + x; the four 1x1 convs carry no bias."""
    def proj(name):
        w = p[prefix + "." + name + ".weight"]
        return torch.einsum("oc,bcn->bon", w.reshape(w.shape[0], w.shape[1]), x)
    theta, phi, g = proj("theta"), proj("phi"), proj("g")
    beta = F.softmax(torch.bmm(theta.transpose(1, 2), phi), -1)                 # [B,N,N]
    o = torch.bmm(g, beta.transpose(1, 2))                                      # [B,C/2,N]
    wo = p[prefix + ".o.weight"]
    o = torch.einsum("oc,bcn->bon", wo.reshape(wo.shape[0], wo.shape[1]), o)
    return p[prefix + ".gamma"] * o + x


def eql_effective_params(p: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """--eql (Generator.py:103-104; EqualLR, Generation/modules.py:259-288): map the state_dict names of the equalised-LR
    layers (`head.0.conv.weight_orig`, `global_conv.0.linear.bias`, ...) to the plain names used by this file, with the
    run-time scaling weight = weight_orig * sqrt(2 / fan_in), fan_in = Cin * kernel size.  Differentiable."""
    out = {}
    for k, v in p.items():
        for inner in (".conv.", ".linear."):
            if inner in k:
                base, leaf = k.split(inner)
                if leaf == "weight_orig":
                    fan_in = v.shape[1] * (v[0][0].numel() if v.dim() > 2 else 1)
                    out[base + ".weight"] = v * math.sqrt(2.0 / fan_in)
                else:
                    out[base + "." + leaf] = v
                break
        else:
            out[k] = v
    return out


def generator_style(p, x: Tensor, z: Tensor, z_norm: bool = False) -> Tensor:
    """The style branch: head MLP on cat[x, z] (Generator.py:163-169 / 203-215).  -> [B,128,N]"""
    if z_norm:                                                  # Generator.py:163-164
        z = z / (z.norm(p=2, dim=-1, keepdim=True) + 1e-8)
    style = torch.cat([x, z], dim=-1).transpose(2, 1).contiguous()
    style = F.leaky_relu(_conv1x1(style, p, "head.0"), NEG)
    return F.leaky_relu(_conv1x1(style, p, "head.2"), NEG)


def generator_body(p, x: Tensor, style: Tensor, k: int = 10, training: bool = True, buffers=None,
                   idx1: Optional[Tensor] = None, idx2: Optional[Tensor] = None, off: bool = False,
                   stages: Optional[dict] = None) -> Tensor:
    """Everything behind the style branch (Generator.py:170-198 == 232-261): the two EdgeConv + AdaIN stages, the global
    feature, the tail."""
    B, N, _ = x.shape
    pc = x.transpose(2, 1).contiguous()
    if "pc_head.0.weight" in p:                                 # --use_head, Generator.py:138-143,171-172 (LeakyReLU default slope)
        pc = F.leaky_relu(_conv1x1(pc, p, "pc_head.0"), 0.01)
        pc = F.leaky_relu(_conv1x1(pc, p, "pc_head.2"), 0.01)
    x1, i1 = edge_block(p, "EdgeConv1", pc, k, idx=idx1, training=training, buffers=buffers, return_idx=True)
    x1 = adaptive_point_norm(p, "adain1", F.leaky_relu(x1, NEG_2), style)
    x2, i2 = edge_block(p, "EdgeConv2", x1, k, idx=idx2, training=training, buffers=buffers, return_idx=True)
    x2 = adaptive_point_norm(p, "adain2", F.leaky_relu(x2, NEG_2), style)

    g = torch.max(x2, 2)[0]                                     # [B,128]  Generator.py:183
    g = F.linear(g, p["global_conv.0.weight"], p["global_conv.0.bias"])
    g = F.leaky_relu(_bn(g, p, "global_conv.1", training, buffers), NEG)
    g = F.linear(g, p["global_conv.3.weight"], p["global_conv.3.bias"])
    g = F.leaky_relu(_bn(g, p, "global_conv.4", training, buffers), NEG)
    feat = torch.cat([g.view(B, -1, 1).expand(B, g.shape[1], N), x2], dim=1)    # [B,640,N]
    if "attn.gamma" in p:                                       # --attn, Generator.py:191-192
        feat = attention(p, "attn", feat)

    t = F.leaky_relu(_conv1x1(feat, p, "tail.0"), NEG)
    t = F.leaky_relu(_conv1x1(t, p, "tail.2"), NEG)
    out = torch.tanh(_conv1x1(t, p, "tail.4"))
    if stages is not None:
        stages.update(style=style, x1=x1, x2=x2, idx1=i1, idx2=i2, feat_global=g, out=out)
    return x.transpose(2, 1) + out if off else out              # Generator.py:196 (pc is the coordinates when use_head is off)


def generator_forward(p, x: Tensor, z: Tensor, k: int = 10, training: bool = True,
                      buffers=None, idx1: Optional[Tensor] = None, idx2: Optional[Tensor] = None,
                      off: bool = False, z_norm: bool = False, stages: Optional[dict] = None) -> Tensor:
    """Generator.forward with default flags (use_head=False, attn=False, eql=False),
    Generation/Generator.py:160-198.  x [B,N,3], z [B,N,nz] -> [B,3,N].
    `stages`, when given, receives the intermediate activations (used by the golden tests)."""
    return generator_body(p, x, generator_style(p, x, z, z_norm), k, training, buffers, idx1, idx2, off, stages)


def generator_interpolate(p, x: Tensor, z1: Tensor, z2: Tensor, selection: Tensor, alpha: float, use_latent: bool = False,
                          k: int = 10, training: bool = False, buffers=None, idx1: Optional[Tensor] = None,
                          idx2: Optional[Tensor] = None, off: bool = False, z_norm: bool = False,
                          stages: Optional[dict] = None) -> Tensor:
    """Generator.interpolate, Generation/Generator.py:200-261: on the points with selection == 1 blend the two latents
    (use_latent=False, :204-206) or the two styles (use_latent=True, :217-230) with weight alpha, then the common body.
    (The reference blends in place into z1 / style_1; inputs are not modified here.)"""
    sel = selection == 1
    if not use_latent:
        z = z1.clone()
        z[:, sel] = z1[:, sel] * (1 - alpha) + z2[:, sel] * alpha
        style = generator_style(p, x, z, z_norm)
    else:
        s1, s2 = generator_style(p, x, z1, z_norm), generator_style(p, x, z2, z_norm)
        style = s1.clone()
        style[:, :, sel] = s1[:, :, sel] * (1 - alpha) + s2[:, :, sel] * alpha
    return generator_body(p, x, style, k, training, buffers, idx1, idx2, off, stages)


def discriminator_forward(p, x: Tensor, training: bool = True, buffers=None,
                          stages: Optional[dict] = None) -> Tensor:
    """Discriminator.forward, Generation/Discriminator.py:97-115. x [B,3,N] -> [B,1]."""
    h = x
    for conv, bn in (("mlps.0", "mlps.1"), ("mlps.3", "mlps.4"), ("mlps.6", "mlps.7"), ("fc2.0", "fc2.1")):
        h = F.leaky_relu(_bn(_conv1x1(h, p, conv), p, bn, training, buffers), NEG)
    pooled = torch.max(h, 2)[0]                                 # adaptive_max_pool1d(.,1)
    m = pooled
    for i in (0, 2, 4):
        m = F.leaky_relu(F.linear(m, p["mlp.%d.weight" % i], p["mlp.%d.bias" % i]), NEG)
    out = F.linear(m, p["mlp.6.weight"], p["mlp.6.bias"])
    if stages is not None:
        stages.update(pooled=pooled, out=out)
    return out


# --------------------------------------------------------------------------- #
# losses (Common/loss_utils.py) and the WGAN-GP penalty                       #
# --------------------------------------------------------------------------- #
def dis_loss(d_real: Tensor, d_fake: Tensor, gan: str = "ls",
             real_label: Optional[Tensor] = None, fake_label: Optional[Tensor] = None) -> Tensor:
    """loss_utils.py:854-972.  For 'ls' the labels are shape [B] against logits
    [B,1]; F.mse_loss then broadcasts to [B,B] (reference quirk, kept)."""
    gan = gan.lower()
    if gan == "wgan":                                           # :859-863
        return d_fake.mean() - d_real.mean()
    if gan == "hinge":                                          # :870-871,877
        return F.relu(1.0 - d_real).mean() + F.relu(1.0 + d_fake).mean()
    if gan == "ls":                                             # :889-939
        B = d_fake.shape[0]
        rl = torch.ones(B) if real_label is None else real_label
        fl = torch.zeros(B) if fake_label is None else fake_label
        return (_mse_bcast(d_fake, fl) + _mse_bcast(d_real, rl)) / 2.0
    if gan == "gan":                                            # :947-953 (BCE-with-logits on both)
        ones = torch.ones_like(d_real)
        return (F.binary_cross_entropy_with_logits(d_fake, torch.zeros_like(d_fake))
                + F.binary_cross_entropy_with_logits(d_real, ones)) / 2.0
    raise NotImplementedError(gan)


def gen_loss(d_real: Tensor, d_fake: Tensor, gan: str = "ls", fake_label: Optional[Tensor] = None) -> Tensor:
    """loss_utils.py:727-802 (d_real is accepted and ignored for ls/wgan/hinge, as there)."""
    gan = gan.lower()
    if gan in ("wgan", "hinge"):                                # :728-729, :735-736
        return -d_fake.mean()
    if gan == "ls":                                             # :747-763
        fl = torch.ones(d_fake.shape[0]) if fake_label is None else fake_label
        return _mse_bcast(d_fake, fl)
    if gan == "gan":
        return F.binary_cross_entropy_with_logits(d_fake, torch.ones_like(d_fake))
    raise NotImplementedError(gan)


def _mse_bcast(logit: Tensor, label: Tensor) -> Tensor:
    """F.mse_loss(logit[B,1], label[B]) -> mean over the broadcast [B,B] square."""
    return ((logit.view(-1, 1) - label.view(1, -1)) ** 2).mean()


def gradient_penalty(d_fn, real: Tensor, fake: Tensor, alpha: Tensor,
                     lambda_gp: float = 10.0, gamma: float = 1.0) -> Tensor:
    """Common/gradient_penalty.py:19-37 with the per-sample mixing factor `alpha`
    [B,1,1] passed in (the reference draws it with torch.rand on the fly)."""
    B = real.shape[0]
    xhat = real + alpha * (fake[:B] - real)
    if not xhat.requires_grad:
        xhat.requires_grad_(True)
    out = d_fn(xhat)
    g = torch.autograd.grad(out, xhat, grad_outputs=torch.ones_like(out),
                            create_graph=True, retain_graph=True, only_inputs=True)[0]
    g = g.contiguous().view(B, -1)
    return (((g.norm(2, dim=1) - gamma) / gamma) ** 2).mean() * lambda_gp


def gradient_penalty_loss_utils(d_fn, real: Tensor, fake: Tensor, alpha: Tensor, lambda_gp: float = 10.0, gamma: float = 1.0,
                                mapping: bool = False) -> Tensor:
    """The second GradientPenalty, Common/loss_utils.py:1087-1131: x_hat = alpha*real + (1-alpha)*fake (:1108); with
    mapping=True (:1110-1118) every fake point is first paired with a real point by the auction EMD (`emd_auction(fake, real,
    0.005, 300)` above) and x_hat = alpha*fake + (1-alpha)*real[assignment].  real/fake [B,3,N]."""
    B = real.shape[0]
    fake = fake[:B]
    if mapping:
        f, r = fake.transpose(1, 2).contiguous(), real.transpose(1, 2).contiguous()
        _, ass = emd_auction(f.detach().numpy(), r.detach().numpy(), 0.005, 300)
        paired = torch.stack([r[i][torch.from_numpy(ass[i])] for i in range(B)])
        xhat = (alpha * f + (1.0 - alpha) * paired).transpose(1, 2).contiguous()
    else:
        xhat = alpha * real + (1 - alpha) * fake
    xhat = xhat.detach().requires_grad_(True)
    out = d_fn(xhat)
    g = torch.autograd.grad(out, xhat, grad_outputs=torch.ones_like(out), create_graph=True, retain_graph=True, only_inputs=True)[0]
    return (((g.contiguous().view(B, -1).norm(2, dim=1) - gamma) / gamma) ** 2).mean() * lambda_gp


# --------------------------------------------------------------------------- #
# optimiser + one train step (Generation/model.py:239-279)                    #
# --------------------------------------------------------------------------- #
def adam_update(param: Tensor, grad: Tensor, m: Tensor, v: Tensor, step: int,
                lr: float = 1e-4, beta1: float = 0.5, beta2: float = 0.99, eps: float = 1e-8) -> None:
    """torch.optim.Adam semantics (no weight decay, no amsgrad), in place.
    Hyper-parameters: Generation/model.py:94-97."""
    m.mul_(beta1).add_(grad, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-lr / bc1)


class AdamState:
    def __init__(self, params: Dict[str, Tensor]):
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.step = 0

    def apply(self, params, grads, lr=1e-4):
        self.step += 1
        for k in params:
            adam_update(params[k].data, grads[k], self.m[k], self.v[k], self.step, lr=lr)


def train_step(gp_, gbuf, dp_, dbuf, optG: AdamState, optD: AdamState, sphere: Tensor, real: Tensor,
               z_d: Tensor, z_g: Tensor, gan: str = "ls", use_gp: bool = False,
               alpha: Optional[Tensor] = None, lambda_gp: float = 10.0, k: int = 10, lr: float = 1e-4, graphs=None):
    """One iteration of the reference loop body (Generation/model.py:239-279):
    D-step (G frozen, no graph through G) then G-step (D frozen: input grads only).
    `real` is [B,N,3]; sphere [B,N,3]; z_* [B,N,nz].  With use_gp the D loss is
    dis_loss(gan) + GradientPenalty(lambda_gp, gamma=1).
    Params are leaf tensors with requires_grad=True; returns dict of scalars + grads.
    graphs = (idx2 of the D step's generator forward, idx2 of the G step's): EdgeConv2's neighbour graphs [B,N*k] handed in instead of
    built (tie-aware protocol of the multi-step golden G19)."""
    out = {}
    g_d, g_g = graphs if graphs is not None else (None, None)
    # ---- D step (model.py:240-260) ----
    with torch.no_grad():
        fake = generator_forward(gp_, sphere, z_d, k=k, training=True, buffers=gbuf, idx2=g_d)
    real_t = real.transpose(2, 1).contiguous()
    d_real = discriminator_forward(dp_, real_t, True, dbuf)
    d_fake = discriminator_forward(dp_, fake, True, dbuf)
    loss_d = dis_loss(d_real, d_fake, gan)
    if use_gp:
        loss_d = loss_d + gradient_penalty(lambda t: discriminator_forward(dp_, t, True, dbuf),
                                           real_t, fake, alpha, lambda_gp)
    dnames = list(dp_.keys())
    dgrads = dict(zip(dnames, torch.autograd.grad(loss_d, [dp_[n] for n in dnames])))
    optD.apply(dp_, dgrads, lr)
    out.update(loss_d=loss_d.detach(), d_grads=dgrads, fake_d=fake)
    # ---- G step (model.py:264-279) ----
    fake_g = generator_forward(gp_, sphere, z_g, k=k, training=True, buffers=gbuf, idx2=g_g)
    with torch.no_grad():
        d_real_g = discriminator_forward(dp_, real_t, True, dbuf)   # unused by the loss, but it
    #                                                                 advances D's BN running stats
    d_fake_g = discriminator_forward(dp_, fake_g, True, dbuf)
    loss_g = gen_loss(d_real_g, d_fake_g, gan)
    gnames = list(gp_.keys())
    ggrads = dict(zip(gnames, torch.autograd.grad(loss_g, [gp_[n] for n in gnames])))
    optG.apply(gp_, ggrads, lr)
    out.update(loss_g=loss_g.detach(), g_grads=ggrads, fake_g=fake_g.detach())
    return out


# --------------------------------------------------------------------------- #
# ball-query / grouping family (orphans named by the north star)              #
# --------------------------------------------------------------------------- #
def square_distance(src: Tensor, dst: Tensor) -> Tensor:
    """[B,N,C],[B,M,C] -> [B,N,M], same expanded form (Common/pointnet_util.py:19-40)."""
    d = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    d = d + torch.sum(src ** 2, -1).unsqueeze(-1)
    d = d + torch.sum(dst ** 2, -1).unsqueeze(1)
    return d


def index_points(points: Tensor, idx: Tensor) -> Tensor:
    """points [B,N,C], idx [B,S] or [B,S,K] -> [B,S,(K,)C] (pointnet_util.py:43-60)."""
    B = points.shape[0]
    bidx = torch.arange(B).view(B, *([1] * (idx.dim() - 1))).expand_as(idx)
    return points[bidx, idx]


def query_ball_point(radius: float, nsample: int, xyz: Tensor, new_xyz: Tensor) -> Tensor:
    """First `nsample` indices (ascending index order) with d^2 <= r^2, padded with the
    first hit (pointnet_util.py:87-107; the mask there is `> r^2`)."""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    d = square_distance(new_xyz, xyz)
    gi = torch.arange(N).view(1, 1, N).repeat(B, S, 1)
    gi[d > radius ** 2] = N
    gi = gi.sort(dim=-1)[0][:, :, :nsample]
    first = gi[:, :, :1].expand(B, S, gi.shape[2])
    return torch.where(gi == N, first, gi)


def knn_point(nsample: int, xyz: Tensor, new_xyz: Tensor) -> Tensor:
    """k nearest (self included), order unspecified in the reference (topk sorted=False,
    Common/pointconv_util.py:107-118); the oracle returns them sorted ascending."""
    d = square_distance(new_xyz, xyz)
    return torch.topk(d, nsample, dim=-1, largest=False, sorted=True)[1]


def group(nsample: int, xyz: Tensor, points: Optional[Tensor], idx: Optional[Tensor] = None):
    """kNN-group every point (pointconv_util.py:174-197): returns
    (new_points [B,N,K,C+D], grouped_xyz_norm [B,N,K,C])."""
    B, N, C = xyz.shape
    if idx is None:
        idx = knn_point(nsample, xyz, xyz)
    gx = index_points(xyz, idx) - xyz.view(B, N, 1, C)
    if points is None:
        return gx, gx
    return torch.cat([gx, index_points(points, idx)], dim=-1), gx


def farthest_point_sample(xyz: Tensor, npoint: int, start: Optional[Tensor] = None) -> Tensor:
    """Iterative FPS (pointnet_util.py:63-84 draws a random start; pointconv_util.py:60-83
    starts at 0).  `start` [B] int64; default zeros."""
    B, N, _ = xyz.shape
    cent = torch.zeros(B, npoint, dtype=torch.long)
    dist = torch.full((B, N), 1e10)
    far = torch.zeros(B, dtype=torch.long) if start is None else start.clone()
    ar = torch.arange(B)
    for i in range(npoint):
        cent[:, i] = far
        c = xyz[ar, far].view(B, 1, -1)
        d = torch.sum((xyz - c) ** 2, -1)
        dist = torch.minimum(dist, d)
        far = torch.max(dist, -1)[1]
    return cent


def sample_and_group(npoint, radius, nsample, xyz, points, start=None):
    """pointnet_util.py:110-143 -> (new_xyz [B,S,3], new_points [B,S,ns,3+D])."""
    B, N, C = xyz.shape
    fps = farthest_point_sample(xyz, npoint, start)
    new_xyz = index_points(xyz, fps)
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    gx = index_points(xyz, idx) - new_xyz.view(B, npoint, 1, C)
    if points is not None:
        return new_xyz, torch.cat([gx, index_points(points, idx)], dim=-1)
    return new_xyz, gx


# --------------------------------------------------------------------------- #
# parameter containers                                                        #
# --------------------------------------------------------------------------- #
def generator_shapes(nz: int = 128, k: int = 10, use_head: bool = False, attn: bool = False, eql: bool = False) -> Dict[str, Tuple[int, ...]]:
    """state_dict parameter shapes of Generator, Generator.py:107-153 (default flags; --use_head: 138-148; --attn: 116-117 with
    Generation/modules.py:541-546; --eql: the Conv/Linear of head, global_conv and pc_head become `<name>.conv.weight_orig` /
    `<name>.linear.weight_orig` + bias, Generator.py:103-104)."""
    d = 128
    s = {
        "head.0.weight": (d, 3 + nz, 1), "head.0.bias": (d,),
        "head.2.weight": (d, d, 1), "head.2.bias": (d,),
        "global_conv.0.weight": (d, d), "global_conv.0.bias": (d,),
        "global_conv.1.weight": (d,), "global_conv.1.bias": (d,),
        "global_conv.3.weight": (512, d), "global_conv.3.bias": (512,),
        "global_conv.4.weight": (512,), "global_conv.4.bias": (512,),
        "tail.0.weight": (256, 512 + d, 1), "tail.0.bias": (256,),
        "tail.2.weight": (64, 256, 1), "tail.2.bias": (64,),
        "tail.4.weight": (3, 64, 1), "tail.4.bias": (3,),
    }
    if use_head:
        s.update({"pc_head.0.weight": (d // 2, 3, 1), "pc_head.0.bias": (d // 2,), "pc_head.2.weight": (d, d // 2, 1), "pc_head.2.bias": (d,)})
    for name, fin, fout in ((("EdgeConv1", d, d), ("EdgeConv2", d, d)) if use_head else (("EdgeConv1", 3, 64), ("EdgeConv2", 64, d))):
        s.update({
            name + ".conv_w.0.weight": (fout // 2, fin, 1, 1), name + ".conv_w.0.bias": (fout // 2,),
            name + ".conv_w.1.weight": (fout // 2,), name + ".conv_w.1.bias": (fout // 2,),
            name + ".conv_w.3.weight": (fout, fout // 2, 1, 1), name + ".conv_w.3.bias": (fout,),
            name + ".conv_w.4.weight": (fout,), name + ".conv_w.4.bias": (fout,),
            name + ".conv_x.0.weight": (fout, 2 * fin, 1, 1), name + ".conv_x.0.bias": (fout,),
            name + ".conv_x.1.weight": (fout,), name + ".conv_x.1.bias": (fout,),
            name + ".conv_out.weight": (fout, fout, 1, k), name + ".conv_out.bias": (fout,),
        })
    s.update({"adain1.style.weight": (256 if use_head else 128, d, 1), "adain1.style.bias": (256 if use_head else 128,),
              "adain2.style.weight": (256, d, 1), "adain2.style.bias": (256,)})
    if attn:
        ch = d + 512
        s.update({"attn.theta.weight": (ch // 8, ch, 1), "attn.phi.weight": (ch // 8, ch, 1), "attn.g.weight": (ch // 2, ch, 1),
                  "attn.o.weight": (ch, ch // 2, 1), "attn.gamma": ()})
    if eql:
        ren = {}
        for key, shp in s.items():
            base, leaf = key.rsplit(".", 1)
            if base in ("head.0", "head.2", "pc_head.0", "pc_head.2"):
                ren[base + ".conv." + ("weight_orig" if leaf == "weight" else leaf)] = shp
            elif base in ("global_conv.0", "global_conv.3"):
                ren[base + ".linear." + ("weight_orig" if leaf == "weight" else leaf)] = shp
            else:
                ren[key] = shp
        s = ren
    return s


def discriminator_shapes(small_d: bool = False) -> Dict[str, Tuple[int, ...]]:
    """state_dict parameter shapes of Discriminator, Discriminator.py:55-95."""
    dim = 512 if small_d else 1024
    return {
        "mlps.0.weight": (64, 3, 1), "mlps.0.bias": (64,), "mlps.1.weight": (64,), "mlps.1.bias": (64,),
        "mlps.3.weight": (128, 64, 1), "mlps.3.bias": (128,), "mlps.4.weight": (128,), "mlps.4.bias": (128,),
        "mlps.6.weight": (256, 128, 1), "mlps.6.bias": (256,), "mlps.7.weight": (256,), "mlps.7.bias": (256,),
        "fc2.0.weight": (dim, 256, 1), "fc2.0.bias": (dim,), "fc2.1.weight": (dim,), "fc2.1.bias": (dim,),
        "mlp.0.weight": (512, dim), "mlp.0.bias": (512,), "mlp.2.weight": (256, 512), "mlp.2.bias": (256,),
        "mlp.4.weight": (64, 256), "mlp.4.bias": (64,), "mlp.6.weight": (1, 64), "mlp.6.bias": (1,),
    }


def bn_buffers(shapes: Dict[str, Tuple[int, ...]]) -> Dict[str, Tensor]:
    """Fresh running stats for every BN layer implied by `shapes` (1-D weight entries
    whose sibling conv/linear precedes them: *.1, *.4, *.7 in the Sequentials)."""
    out = {}
    for name, shp in shapes.items():
        if name.endswith(".weight") and len(shp) == 1:
            base = name[:-len(".weight")]
            out[base + ".running_mean"] = torch.zeros(shp)
            out[base + ".running_var"] = torch.ones(shp)
            out[base + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return out


# --------------------------------------------------------------------------- #
# evaluation metrics, Chamfer part (metrics/evaluation_metrics.py)            #
# --------------------------------------------------------------------------- #
def dist_chamfer(a: Tensor, b: Tensor) -> Tuple[Tensor, Tensor]:
    """distChamfer, evaluation_metrics.py:39-50 (expanded form |x|^2 + |y|^2 - 2xy; bs x N x 3 both, same N):
    returns (min over the points of a for every point of b, min over the points of b for every point of a)."""
    xx = torch.bmm(a, a.transpose(2, 1))
    yy = torch.bmm(b, b.transpose(2, 1))
    zz = torch.bmm(a, b.transpose(2, 1))
    n = a.shape[1]
    d = torch.arange(n)
    rx = xx[:, d, d].unsqueeze(1).expand_as(xx)
    ry = yy[:, d, d].unsqueeze(1).expand_as(yy)
    P = rx.transpose(2, 1) + ry - 2 * zz
    return P.min(1)[0], P.min(2)[0]


def nn_distance(a: Tensor, b: Tensor):
    """chamferFunction.forward (ChamferDistance.py:13-31 over chamfer.cu:12-113): direct differences; for every point of a the
    squared distance to and the index of its nearest point of b (first minimum), and vice versa."""
    d = ((a[:, :, None, :] - b[:, None, :, :]) ** 2).sum(-1)
    d1, i1 = d.min(2)
    d2, i2 = d.min(1)
    return d1, d2, i1.int(), i2.int()


def pairwise_cd(sample: Tensor, ref: Tensor) -> Tensor:
    """The CD half of _pairwise_EMD_CD_ (evaluation_metrics.py:89-126): [S, R] with dl.mean + dr.mean per pair."""
    rows = []
    for s in range(sample.shape[0]):
        dl, dr = dist_chamfer(sample[s:s + 1].expand(ref.shape[0], -1, -1).contiguous(), ref)
        rows.append((dl.mean(dim=1) + dr.mean(dim=1)).view(1, -1))
    return torch.cat(rows, dim=0)


def lgan_mmd_cov(all_dist: Tensor) -> dict:
    """evaluation_metrics.py:161-173: all_dist [N_sample, N_ref]."""
    n_ref = all_dist.shape[1]
    min_val_fromsmp, min_idx = torch.min(all_dist, dim=1)
    min_val, _ = torch.min(all_dist, dim=0)
    return {"lgan_mmd": min_val.mean(), "lgan_cov": torch.tensor(float(min_idx.unique().numel()) / float(n_ref)),
            "lgan_mmd_smp": min_val_fromsmp.mean()}


def one_nn_accuracy(Mxx: Tensor, Mxy: Tensor, Myy: Tensor, k: int = 1) -> dict:
    """knn, evaluation_metrics.py:129-158 (leave-one-out k-NN two-sample test on the joint distance matrix)."""
    n0, n1 = Mxx.shape[0], Myy.shape[0]
    label = torch.cat((torch.ones(n0), torch.zeros(n1)))
    M = torch.cat((torch.cat((Mxx, Mxy), 1), torch.cat((Mxy.t(), Myy), 1)), 0)
    M = M + torch.diag(float("inf") * torch.ones(n0 + n1))
    _, idx = M.topk(k, 0, False)
    count = torch.zeros(n0 + n1)
    for i in range(k):
        count = count + label.index_select(0, idx[i])
    pred = (count >= float(k) / 2).float()
    tp, fp = (pred * label).sum(), (pred * (1 - label)).sum()
    fn, tn = ((1 - pred) * label).sum(), ((1 - pred) * (1 - label)).sum()
    return {"acc_t": tp / (tp + fn + 1e-10), "acc_f": tn / (tn + fp + 1e-10), "acc": (label == pred).float().mean()}


# --------------------------------------------------------------------------- #
# JSD between occupancy grids (metrics/evaluation_metrics.py:208-322)         #
# --------------------------------------------------------------------------- #
def unit_cube_grid_point_cloud(resolution: int, clip_sphere: bool = False):
    """evaluation_metrics.py:210-229: float32 cell centres i*spacing - 0.5 (x slowest), optionally clipped to |g| <= 0.5."""
    import numpy as np
    spacing = 1.0 / float(resolution - 1)
    ax = (np.arange(resolution, dtype=np.float64) * spacing - 0.5).astype(np.float32)
    grid = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), axis=-1)
    if clip_sphere:
        grid = grid.reshape(-1, 3)
        grid = grid[np.linalg.norm(grid, axis=1) <= 0.5]
    return grid, spacing


def _np_entropy(p, base=None):
    import numpy as np
    p = np.asarray(p, dtype=np.float64)
    p = p / p.sum()
    h = -np.sum(np.where(p > 0, p * np.log(np.where(p > 0, p, 1.0)), 0.0))
    return h / np.log(base) if base else h


def entropy_of_occupancy_grid(pclouds, grid_resolution: int, in_sphere: bool = False):
    """evaluation_metrics.py:247-290 with the k-d tree replaced by an exhaustive float64 search (lowest index on ties).
    pclouds: numpy [S,N,3].  -> (mean Bernoulli entropy over the cells, grid_counters [G])."""
    import numpy as np
    grid = unit_cube_grid_point_cloud(grid_resolution, in_sphere)[0].reshape(-1, 3).astype(np.float64)
    counters = np.zeros(len(grid)); bern = np.zeros(len(grid))
    for pc in np.asarray(pclouds, dtype=np.float64):
        idx = np.empty(len(pc), dtype=np.int64)
        for lo in range(0, len(pc), 512):
            d = ((pc[lo:lo + 512, None, :] - grid[None, :, :]) ** 2).sum(-1)
            idx[lo:lo + 512] = d.argmin(1)
        np.add.at(counters, idx, 1)
        bern[np.unique(idx)] += 1
    n = float(len(pclouds))
    acc = sum(_np_entropy([g / n, 1.0 - g / n]) for g in bern if g > 0)
    return acc / len(counters), counters


def jensen_shannon_divergence(P, Q):
    """evaluation_metrics.py:293-312."""
    import numpy as np
    P = np.asarray(P, dtype=np.float64); Q = np.asarray(Q, dtype=np.float64)
    if np.any(P < 0) or np.any(Q < 0):
        raise ValueError("Negative values.")
    if len(P) != len(Q):
        raise ValueError("Non equal size.")
    P_, Q_ = P / P.sum(), Q / Q.sum()
    return _np_entropy((P_ + Q_) / 2.0, 2) - (_np_entropy(P_, 2) + _np_entropy(Q_, 2)) / 2.0


def jsd_between_point_cloud_sets(sample_pcs, ref_pcs, resolution: int = 28):
    """evaluation_metrics.py:232-244."""
    return jensen_shannon_divergence(entropy_of_occupancy_grid(sample_pcs, resolution, True)[1],
                                     entropy_of_occupancy_grid(ref_pcs, resolution, True)[1])


# --------------------------------------------------------------------------- #
# approximate EMD by auction (metrics/emd/emd_cuda.cu, emd_module.py)         #
# Parity status of THIS section: pinned up to the reference's documented     #
# race.  The reference implementation is CUDA-only (it cannot run in the     #
# build container) and its winner among near-tied bidders depends on thread  #
# timing (emd_cuda.cu:179-192).  tests/golden/make_emd_trace.py emulates the #
# CUDA kernels sequentially in the file's own arithmetic with a fixed thread #
# order (golden G21); tests/test_oracle_golden.py::test_emd_trace_* state    #
# round by round where this restatement equals it (rounds 1-3: every         #
# variant; <= 10: the lowest-bidder order) and where only the race-sized     #
# bound and the exact optimum (scipy linear_sum_assignment,                  #
# test_emd_auction_against_optimal_assignment) hold.                         #
# --------------------------------------------------------------------------- #
def emd_auction(xyz1, xyz2, eps: float = 0.005, iters: int = 50):
    """emd_cuda_forward (emd_cuda.cu:238-277) for numpy clouds [B,n,3] (float32): synchronous auction in float32 arithmetic.
    Per iteration every unassigned point bids for the object with the best value 3 - |x - y_k| - price_k with the increment
    best - second best + eps (:141-176); every object takes its highest bidder (lowest index on ties), evicts its previous owner
    and raises its price (:179-214); in the last iteration the still unassigned points take the objects they bid for (:200).
    -> (dist [B,n] squared distance to the assigned point, assignment [B,n])."""
    import numpy as np
    f = np.float32
    xyz1 = np.asarray(xyz1, dtype=f); xyz2 = np.asarray(xyz2, dtype=f)
    B, n, _ = xyz1.shape
    eps = f(eps)
    dist = np.zeros((B, n), dtype=f); assign = np.full((B, n), -1, dtype=np.int64)
    for b in range(B):
        a, inv, price = assign[b], np.full(n, -1, dtype=np.int64), np.zeros(n, dtype=f)
        for it in range(iters):
            un = np.nonzero(a < 0)[0]
            if len(un) == 0:
                break
            d = xyz2[b][None, :, :] - xyz1[b][un][:, None, :]                        # y - x like the kernel (squares are the same)
            d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
            val = (f(3.0) - np.sqrt(d2)) - price[None, :]
            best_i = val.argmax(1)                                                   # first (lowest) index among equal values
            best = val[np.arange(len(un)), best_i]
            val[np.arange(len(un)), best_i] = -np.inf
            better = np.maximum(val.max(1), f(-1e9)) if n > 1 else np.full(len(un), f(-1e9))
            inc = (best - better) + eps
            if it == iters - 1:
                a[un] = best_i
                break
            # highest increment wins an object, lowest bidder index on ties
            order = np.lexsort((un, -inc.astype(np.float64)))
            seen = set()
            for j in order:
                k = best_i[j]
                if k in seen:
                    continue
                seen.add(k)
                if inv[k] >= 0:
                    a[inv[k]] = -1
                inv[k] = un[j]; a[un[j]] = k
                price[k] = price[k] + inc[j]
        dd = xyz1[b] - xyz2[b][a]
        dist[b] = (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2]
    return dist, assign


def emd_approx(sample, ref, eps: float = 0.005, iters: int = 50):
    """evaluation_metrics.py:26-35 with the tree's auction module in place of the absent StructuralLosses.match_cost: [B]."""
    import numpy as np
    dist, _ = emd_auction(sample, ref, eps, iters)
    return np.sqrt(dist).mean(1)
