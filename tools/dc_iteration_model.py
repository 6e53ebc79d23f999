#!/usr/bin/env python3
"""tools/dc_iteration_model.py [trials] -- numpy model of how k_dc_chain_spec finds its start values (kernels.hip, dc_spec_blocks;
DESIGN.md section 9.1), used to choose the iteration before it was written as a kernel.

The integer form of the reference's DC recurrence (sdrj.cpp:277-283) around a rounding threshold T is  z' = z + e + [z < 0]
(z = m - T, e = q - r_lo - 1).  Lane l of a step owns 16 consecutive samples and evaluates them from a speculated start value;
the step stands when every lane started where its predecessor ended.  Modelled here: e ~ round(N(-1/2, sigma^2)) -- an estimate
hovering at T -- for sigma = 60 (the capture-like stream), 17 (a quiet front end, 2 LSB of noise), 8, 2.5 and 1.2 (an offset
many times the noise: pinned to T), 64 x NW lanes.  Printed: mean and maximum number of rounds until the step stands for
  plain    every start := the sum of the totals before it (round 5's first kernel: a Jacobi iteration)
  secant   starts moved by the solution of u' = (1 + s) u + gap, s = secant slope of the lane's total out of its last two evaluations
  final    the same with the first slopes out of the run's own range (-min(1, 16 / range) if it saw both sides of T) and the
           first guess "the middle of {no step below T, 0, every step below T}" -- what the kernel does
A slope is the change of a run's total per unit of its start value: between -1 and 0, because every step merges the states -1
and 0 (two trajectories only ever come closer)."""
import sys

import numpy as np


def truth(e, z0):
    z = z0
    for x in e:
        z = z + x + (1 if z < 0 else 0)
    return z


def evaluate(E, zs):
    z, zmin, zmax = zs.copy(), zs.copy(), zs.copy()
    for i in range(E.shape[1]):
        z = z + E[:, i] + (z < 0)
        zmin, zmax = np.minimum(zmin, z), np.maximum(zmax, z)
    return z, zmin, zmax


def affine_exclusive(A, B):  # u[l + 1] = A[l] u[l] + B[l], u[0] = 0
    u = np.zeros(len(A))
    for l in range(len(A) - 1):
        u[l + 1] = A[l] * u[l] + B[l]
    return u


def rounds(E, z0, scheme, limit=60):
    L = E.shape[0]
    prefix = np.concatenate([[0], np.cumsum(E.sum(1))[:-1]])
    a = z0 + prefix                      # no step before this lane saw the estimate below T
    b = a + E.shape[1] * np.arange(L)    # every step did
    zs = np.minimum(np.maximum(a, 0), b) if scheme == "final" else (b if z0 < 0 else a)
    slope, zp, tp = np.zeros(L), None, None
    for it in range(limit):
        end, zmin, zmax = evaluate(E, zs)
        tot = end - zs
        gap = np.zeros(L, np.int64)
        gap[:-1] = end[:-1] - zs[1:]
        if not gap.any():
            return it + 1
        if scheme != "plain":
            if zp is not None:
                moved = zs != zp
                slope = np.where(moved, np.clip((tot - tp) / np.where(moved, zs - zp, 1), -1, 0), slope)
            elif scheme == "final":
                slope = np.where((zmin < 0) & (zmax >= 0), -np.minimum(1.0, E.shape[1] / (zmax - zmin + 1)), 0.0)
            zp, tp = zs.copy(), tot.copy()
        u = affine_exclusive(1 + slope, gap.astype(float))
        zs = zs + np.rint(u).astype(np.int64)
    return limit


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(4)
    print(f"{'sigma':>6} {'waves':>5} | " + " | ".join(f"{s:>14}" for s in ("plain", "secant", "final")) + "   (mean rounds, max; limit 60)")
    for sigma in (60, 17, 8, 2.5, 1.2):
        for nw in (1, 4, 8):
            L = 64 * nw
            cases = []
            for _ in range(trials):
                z0 = int(truth(np.rint(rng.normal(-0.5, sigma, 4000)).astype(np.int64), 0))  # somewhere in its hover
                cases.append((np.rint(rng.normal(-0.5, sigma, L * 16)).astype(np.int64).reshape(L, 16), z0))
            cells = []
            for scheme in ("plain", "secant", "final"):
                r = [rounds(E, z0, scheme) for E, z0 in cases]
                cells.append(f"{np.mean(r):8.1f} {max(r):5d}")
            print(f"{sigma:6} {nw:5d} | " + " | ".join(cells))


if __name__ == "__main__":
    main()
