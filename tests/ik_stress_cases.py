"""The IK evaluation sweep (round 3's tools/ik_stress.py, now a test helper): every (frames, tasks, mode) shape the evaluation's
task split, task grouping and the solve's forms (dual, primal with 6 / 11 tiles, LDS fall-back) can take.  Each combination is
reproducible on its own from (seed, n, K, mode) — so an outlier is a NAMED case, and the build container can regenerate its inputs
to run the reference's own fp32 autograd Jacobian on it (oracle/gen_outliers.py -> tests/golden/ik_outliers.npz).

Not a test module itself (no test_ prefix): tests/test_ik_gpu.py::test_ik_eval_sweep_vs_oracle drives it.  As a script on the GPU box
it prints the spread and dumps every case beyond the curated bounds:

    python tests/ik_stress_cases.py gpurun_out/ik_outliers_dump.npz
"""
from __future__ import annotations

import itertools
import os
import sys

import numpy as np

NS = [1, 2, 3, 7, 8, 33, 64, 65, 127, 256, 257, 512]
KS = [1, 2, 5, 6, 7, 12, 13, 24, 41, 48]
MODES = ["plain", "normal", "offset", "phi", "beta", "mixed"]
SEED = 4
# curated bounds (relative to max(1, |J|_max) of the frame): what the cases of tests/test_ik_gpu.py hold on hand-picked faces.
# Rows whose derivative passes through a vertex normal — normal rows, and position rows of a task with a normal offset — carry a
# 1 / edge-length amplification of fp32 rounding; plain position rows do not.
BOUND_E = 2e-4
BOUND_J_POS = 1e-4
BOUND_J_NRM = 6e-4


# The cases of this sweep (SEED = 4) beyond the curated bounds when round 4 began, each explained by conditioning, not by an error
# (profiles/r04_ik_sweep.txt): a task face with a SLIVER among the faces around its vertices (the random synthetic skinning crumples
# the posed mesh: |cross| of the sliver in n3_K48_mixed is 5.6e-6 m^2 against ~5e-4 for its neighbours), whose unit normal turns by
# 1e-4 under a 1e-7 m move of one vertex.  The reference's own fp32 autograd residual / Jacobian on exactly these cases is kept in
# tests/golden/ik_outliers.npz (oracle/gen_outliers.py): the fp64 oracle is pinned against it there.
KNOWN_OUTLIERS = ["n1_K6_normal", "n3_K48_mixed", "n64_K12_phi", "n127_K13_normal", "n127_K41_normal"]
# The yardstick for a case beyond the curated bounds: how far the ORACLE's own answer moves when every template vertex of the
# model is displaced by up to 3e-7 m — the fused FK kernel's measured vertex error (fp16x2 operands), 30 x below
# the 1e-5 m bar.  An engine deviation inside that spread is the posed mesh's conditioning; one outside it is an error.
PERTURB_M = 3e-7


def perturbed_oracles(model, trials=4, seed=0):
    from oracle import cpu

    rng = np.random.default_rng(seed)
    out = []
    for _ in range(trials):
        m2 = dict(model)
        vt = model["vertices_template"]
        m2["vertices_template"] = (vt.astype(np.float64) + rng.uniform(-PERTURB_M, PERTURB_M, vt.shape)).astype(np.float32)
        out.append(cpu.OracleModel(m2))
    return out


def conditioning_spread(perturbed, c, r):
    """max over the perturbed models of (|de|, position-class dJ, normal-class dJ) against the unperturbed oracle result r."""
    sp = np.zeros(3)
    for o2 in perturbed:
        r2 = oracle_eval(o2, c)
        dp, dn = deviations(r["J"], r2["J"], c)
        sp = np.maximum(sp, [float(np.abs(r2["e"] - r["e"]).max()), dp, dn])
    return sp


def parse_key(k):
    import re

    m = re.fullmatch(r"n(\d+)_K(\d+)_(\w+)", str(k))
    return int(m.group(1)), int(m.group(2)), m.group(3)


def combinations():
    for n, K in itertools.product(NS, KS):
        if n * K > 512 * 13:
            continue
        for mode in MODES:
            yield n, K, mode


def make_case(n, K, mode, seed=SEED):
    """Inputs of one combination: a batch of n frames (random faces, targets, normals) and the frame the oracle checks."""
    from smplpp_amd import model_io

    rng = np.random.default_rng([seed, n, K, MODES.index(mode)])
    beta, theta = model_io.synthetic_inputs(n, seed=int(rng.integers(1 << 30)))
    theta = theta.copy()
    theta[:, 1:] *= 0.4
    faces = rng.integers(0, 13776, (n, K))
    tp = rng.normal(0, 0.4, (n, K, 3)).astype(np.float32)
    tn = rng.normal(0, 1, (n, K, 3)).astype(np.float32)
    tn /= np.linalg.norm(tn, axis=2, keepdims=True)
    nw, noff, pl, pw = np.zeros((n, K)), np.zeros((n, K)), np.zeros((n, K)), np.ones((n, K))
    ob = False
    if mode == "normal":
        nw[:] = 1.0
    if mode == "offset":
        noff[:] = 0.015
    if mode == "phi":
        pl[:] = 0.04
        nw[:] = 0.7
    if mode == "beta":
        ob = True
        nw[:] = 1.0
        pl[:] = 0.04
    if mode == "mixed":
        nw[:, ::3] = 1.3
        noff[:, 1::3] = 0.015
        pw[:, ::4] = 0.0
        pl[:, ::2] = 0.04
    f = int(rng.integers(n))
    return dict(n=n, K=K, mode=mode, beta=beta, theta=theta, faces=faces, tp=tp, tn=tn, nw=nw, noff=noff, pl=pl, pw=pw, ob=ob, f=f)


def key(n, K, mode):
    return "n%d_K%d_%s" % (n, K, mode)


def oracle_eval(o, c):
    """fp64 analytic e / J of the checked frame from the C oracle."""
    from oracle import cpu

    f = c["f"]
    ts = cpu.TaskSet(c["faces"][f], c["tp"][f], c["tn"][f], phi_limit=c["pl"][f], normal_offset=c["noff"][f])
    ts.normal_task_weight[:] = c["nw"][f]
    ts.pos_task_weight[:] = c["pw"][f]
    return o.ik_eval(c["beta"][f], c["theta"][f], ts, c["ob"])


def row_classes(c):
    """per row of the checked frame: True where the derivative passes through a vertex normal (see BOUND_J_NRM)."""
    f, K = c["f"], c["K"]
    through_normal = np.zeros((K, 4), bool)
    through_normal[:, 3] = True
    through_normal[c["noff"][f] != 0.0, :3] = True
    return through_normal.reshape(-1)


def deviations(J_ref, J, c):
    """(position-class, normal-class) max |J - J_ref| relative to max(1, |J_ref|_max)."""
    scale = max(1.0, float(np.abs(J_ref).max()))
    d = np.abs(np.asarray(J_ref) - np.asarray(J)).max(axis=1) / scale
    cls = row_classes(c)
    return (float(d[~cls].max()) if (~cls).any() else 0.0), (float(d[cls].max()) if cls.any() else 0.0)


def engine_eval(s, c):
    from smplpp_amd.ik import IkSolver

    sol = IkSolver(s, c["n"], c["K"])
    sol.setTasks(face_idx=c["faces"], target_pos=c["tp"], target_normal=c["tn"], phi_limit=c["pl"], normal_offset=c["noff"],
                 normal_task_weight=c["nw"], pos_task_weight=c["pw"])
    sol.setConfig(c["beta"], c["theta"])
    e, J = sol.eval(optimize_beta=c["ob"])
    e2 = sol.iterate(3, enable_qp=(c["mode"] in ("phi", "beta", "mixed")), optimize_beta_from=(1 if c["ob"] else -1))
    _, th = sol.getConfig()
    return e, J, e2, th


def main(dump_path):
    import time

    seed = int(os.environ.get("SWEEP_SEED", SEED))  # (other seeds: how the conditioning rule fares on cases nobody looked at)

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    from oracle import cpu
    from smplpp_amd import model_io
    from smplpp_amd.smpl import SMPL

    model = model_io.synthetic_model()
    s = SMPL()
    s.setDevice("cuda:0")
    s.init(model)
    o = cpu.OracleModel(model)
    t0 = time.time()
    pert = perturbed_oracles(model)
    dump, spread_p, spread_n = {}, [], []
    cnt = bad = 0
    for n, K, mode in combinations():
        c = make_case(n, K, mode, seed)
        e, J, e2, th = engine_eval(s, c)
        r = oracle_eval(o, c)
        f = c["f"]
        de = float(np.abs(r["e"] - e[f]).max())
        dp, dn = deviations(r["J"], J[f], c)
        spread_p.append(dp)
        spread_n.append(dn)
        cnt += 1
        finite = np.isfinite(e).all() and np.isfinite(J).all() and np.isfinite(e2).all() and np.isfinite(th).all()
        if not finite:
            bad += 1
            print("NON-FINITE", key(n, K, mode))
        if de > BOUND_E or dp > BOUND_J_POS or dn > BOUND_J_NRM:
            k = key(n, K, mode)
            sp = conditioning_spread(pert, c, r)
            if de > BOUND_E + 2.0 * sp[0] or dn > BOUND_J_NRM + 2.0 * sp[2] or dp > BOUND_J_POS:
                bad += 1
                print("NOT EXPLAINED BY CONDITIONING:", k)
            print("outlier %-22s frame %3d: de %.3g  position-class rows %.3g  normal-class rows %.3g | the oracle's own spread under a "
                  "%.0e m template perturbation: de %.3g, %.3g, %.3g%s" % (k, f, de, dp, dn, PERTURB_M, sp[0], sp[1], sp[2],
                                                                          "" if k in KNOWN_OUTLIERS else "  (not in KNOWN_OUTLIERS)"))
            dump[k + "/engine_e"], dump[k + "/engine_J"] = e[f], J[f]
            dump[k + "/oracle_e"], dump[k + "/oracle_J"] = r["e"], r["J"]
    sp, sn = np.array(spread_p), np.array(spread_n)
    print("seed %d: %d combinations, %d non-finite or unexplained, %d beyond the curated bounds, %.0f s" % (seed, cnt, bad, len(dump) // 4, time.time() - t0))
    for name, a in (("position-class rows", sp), ("normal-class rows", sn)):
        print("  %s: median %.2g  p90 %.2g  p99 %.2g  max %.2g (relative to max(1, |J|max))" % (name, np.median(a), np.quantile(a, 0.9), np.quantile(a, 0.99), a.max()))
    if dump_path:
        np.savez_compressed(dump_path, **dump)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else None))
