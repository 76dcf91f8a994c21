#!/usr/bin/env python3
"""Sequential CPU emulation of the reference's CUDA auction (metrics/emd/emd_cuda.cu:93-236, driver :238-277) and the trace fixture
tests/golden/g21_emd_trace.npz it produces.

The reference's EMD module is CUDA-only: it cannot be run in the build container, so the HIP auction (sp-gan_amd/csrc/emd.hip) had
no reference-produced vector behind it (round-4 review, "parity unpinned").  This file restates what the CUDA kernels compute,
kernel by kernel (Bid -> GetMax -> Assign, `iters` rounds, the `last` round, CalcDist), in the arithmetic the .cu file writes:

  * Bid (emd_cuda.cu:93-177): for every unassigned point the value of object k is `3.0 - sqrtf(x2*x2 + y2*y2 + z2*z2) - price[k]`
    -- `3.0` is a DOUBLE literal, so the two subtractions are done in double and rounded to float once (:145); the squared distance
    is float, contracted into fused multiply-adds the way nvcc does by default (-fmad=true; `contract="fma"`) or not (`"none"`).
    A point's threads split the objects into contiguous ranges, keep (best, second best, arg best) with strict `>` and are merged
    in ascending thread order with strict `>` (:166-176): best = the maximum, arg best = its LOWEST index, second best = the second
    largest value counted with multiplicity -- independent of the thread partition.  bid increment = best - better + eps (float).
    `max_increments[k]` = the largest increment bid for k (float atomicMax, :9-20, :176).
  * GetMax (:179-192): every unassigned bidder j whose increment lies within 1e-6 (a double tolerance) of max_increments[bid[j]]
    writes max_idx[bid[j]] = j -- a plain store: among bidders within the tolerance the LAST writer wins.  That is the file's one
    data race; the emulation resolves it with a fixed thread order: `order="ascending"` (threads retire in index order: the highest
    such j wins) or `"descending"` (the lowest wins).  Every outcome the hardware can produce picks one of the within-tolerance
    bidders per object; the two orders are the extremes.
  * Assign (:194-214): the winner evicts the previous owner, takes the object, raises its price by ITS OWN increment and resets
    max_increments[k] = -1e9; in the last round every unassigned point takes the object it bid for, without eviction (:200).
    (An evicted point that runs later in the same launch finds max_idx[its old object] = the new winner: no second effect.)
  * the .cu file has no eps schedule: eps is constant over the rounds (callers pass 0.005/50, 0.002/10000, 0.05/3000).

What differs in the HIP build, deliberately (csrc/emd.hip:2-10): the value is computed in float without contraction, and an object's
winner is the EXACT maximum increment, lowest bidder index on exact ties (a packed 64-bit atomicMax) -- no tolerance, no race.
tests/test_oracle_golden.py::test_emd_trace_* compares the two round by round on this fixture.

    python tests/golden/make_emd_trace.py        # rewrites tests/golden/g21_emd_trace.npz (numpy only; no reference import needed)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "sp-gan_amd"))

f32 = np.float32


def _fma(a, b, c):
    """fmaf(a, b, c) for float32 arrays: the product of two floats is exact in double; the sum is rounded to double, then to float
    (double rounding differs from a true fma in ~1e-9 of the cases; immaterial for a trace that already carries a data race)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def bid_values(x1, xyz2, price, contract="fma"):
    """[U, n] float: 3.0 - sqrtf(|y_k - x_i|^2) - price_k as emd_cuda.cu:141-145 evaluates it."""
    d = xyz2[None, :, :] - x1[:, None, :]                                  # x2 = xyz2_buf[k] - x1 (float)
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    if contract == "fma":
        d2 = _fma(dz, dz, _fma(dy, dy, dx * dx))                            # x2*x2 + y2*y2 + z2*z2 under nvcc's default contraction
    else:
        d2 = (dx * dx + dy * dy) + dz * dz
    return (3.0 - np.sqrt(d2).astype(np.float64) - price[None, :].astype(np.float64)).astype(f32)     # double subtractions, one rounding


def emulate_emd_cuda(xyz1, xyz2, eps, iters, order="ascending", contract="fma"):
    """emd_cuda_forward for ONE cloud pair [n,3] float32 -> (dist [n] float32, assignment [n] int32, rounds actually bid in)."""
    n = xyz1.shape[0]
    eps = f32(eps)
    assignment = np.full(n, -1, dtype=np.int32); assignment_inv = np.full(n, -1, dtype=np.int32)
    price = np.zeros(n, dtype=f32)
    bid = np.zeros(n, dtype=np.int32); bid_inc = np.zeros(n, dtype=f32)
    max_inc = np.zeros(n, dtype=f32); max_idx = np.zeros(n, dtype=np.int32)       # emd_module.py:46-51: zeros
    rounds = 0
    for it in range(iters):
        last = it == iters - 1
        un = np.nonzero(assignment == -1)[0]
        if len(un) == 0:
            continue                                                       # Bid: `if (_unass_cnt == 0) continue;` -- nothing else happens either
        rounds += 1
        # ---- Bid
        val = bid_values(xyz1[un], xyz2, price, contract)
        best_i = val.argmax(1)                                             # lowest index among equal values
        rows = np.arange(len(un))
        best = val[rows, best_i]
        masked = val.copy(); masked[rows, best_i] = -np.inf
        better = np.maximum(masked.max(1), f32(-1e9)) if n > 1 else np.full(len(un), f32(-1e9))
        inc = ((best - better) + eps).astype(f32)
        bid[un] = best_i; bid_inc[un] = inc
        np.maximum.at(max_inc, best_i, inc)                                # atomicMax
        # ---- GetMax: plain stores in thread order, the last writer within the 1e-6 tolerance stays
        seq = un if order == "ascending" else un[::-1]
        bi = bid_inc[seq].astype(np.float64); mi = max_inc[bid[seq]].astype(np.float64)
        ok = (bi - 1e-6 <= mi) & (mi <= bi + 1e-6)
        for j, k in zip(seq[ok], bid[seq][ok]):
            max_idx[k] = j
        # ---- Assign (sequential in the same thread order; the result does not depend on it, see the module docstring)
        for j in seq:
            if assignment[j] != -1:
                continue
            k = bid[j]
            if last or max_idx[k] == j:
                prev = assignment_inv[k]
                if not last and prev != -1:
                    assignment[prev] = -1
                assignment_inv[k] = j
                assignment[j] = k
                price[k] = f32(price[k] + bid_inc[j])
                max_inc[k] = f32(-1e9)
    dd = xyz1 - xyz2[assignment]                                           # CalcDist (:216-225); assignment is complete after a `last` round
    dist = (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1] + dd[:, 2] * dd[:, 2]).astype(f32)
    return dist, assignment, rounds


def trace_inputs(B=2, n=1024):
    """Two fixture cloud pairs in [0,1]^3 (the .cu file asks for normalised coordinates, :144); n a multiple of 1024 (:253)."""
    from spgan import fixture_rng as fr
    a = np.stack([(fr.synthetic_real(1, n, seed=2100 + i)[0].numpy() * 0.5 + 0.5) for i in range(B)]).astype(f32)
    b = np.stack([(fr.synthetic_real(1, n, seed=2150 + i)[0].numpy() * 0.45 + 0.5) for i in range(B)]).astype(f32)
    return a, b


ITERS = (1, 2, 3, 5, 10, 20, 50, 100, 300, 1000)
VARIANTS = (("ascending", "fma"), ("descending", "fma"), ("ascending", "none"), ("descending", "none"))
EPS = 0.005


def main():
    a, b = trace_inputs()
    B, n, _ = a.shape
    d = {"eps": np.float32(EPS), "iters": np.array(ITERS), "n": np.int64(n), "B": np.int64(B)}
    for order, contract in VARIANTS:
        for T in ITERS:
            asg = np.zeros((B, n), dtype=np.int16); dist = np.zeros((B, n), dtype=f32)
            for i in range(B):
                dist[i], asg_i, rounds = emulate_emd_cuda(a[i], b[i], EPS, T, order, contract)
                asg[i] = asg_i.astype(np.int16)
            d["assign|%s|%s|%d" % (order, contract, T)] = asg
            d["cost|%s|%s|%d" % (order, contract, T)] = np.sqrt(dist.astype(np.float64)).sum(1)
        print(order, contract, "done: cost after %d rounds %s" % (ITERS[-1], d["cost|%s|%s|%d" % (order, contract, ITERS[-1])]), flush=True)
    # how often do the emulation's own variants (thread order, contraction) disagree?  (rows of the assignment)
    base = ("ascending", "fma")
    for order, contract in VARIANTS[1:]:
        for T in ITERS:
            diff = (d["assign|%s|%s|%d" % (order, contract, T)] != d["assign|%s|%s|%d" % (base + (T,))]).mean()
            print("  %-10s %-4s T=%4d: %.4f of the rows differ from ascending/fma" % (order, contract, T, diff))
    path = os.path.join(HERE, "g21_emd_trace.npz")
    np.savez_compressed(path, **d)
    print("%s %.1f KB" % (path, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
