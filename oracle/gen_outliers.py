"""TEST INFRASTRUCTURE ONLY — the reference's OWN fp32 autograd residual / Jacobian (oracle/_ref: the reference's FK stage sources
compiled unmodified, differentiated row by row as node/node.cpp:823-869 does) on the named outlier cases of the IK evaluation
sweep (tests/ik_stress_cases.py).  Run in the build container (needs oracle/_ref, i.e. /root/reference at build time):

    python3 oracle/gen_outliers.py n1_K6_normal n64_K12_phi ...      -> tests/golden/ik_outliers.npz

Writes DATA only: per case the checked frame's index, the reference's e [4K] and J [4K, D], and the fp64 C oracle's deviation from
it — the yardstick the GPU test holds the engine to (the engine must be as close to the reference as the oracle is: what is left
is then the reference's own fp32 rounding, amplified by 1 / edge length through the vertex normals of ~2 cm triangles).
"""
from __future__ import annotations

import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(keys):
    import ik_stress_cases as S
    from oracle import cpu, ref
    from smplpp_amd import model_io

    if not ref.available():
        raise SystemExit("oracle/_ref/libsmplpp_ref.so is missing: make -C oracle ref (build container only)")
    model = model_io.synthetic_model()
    R = ref.RefModel(model)
    o = cpu.OracleModel(model)
    # the reference's unordered_map adjacency order decides the fp32 summation order of its vertex normals
    adj = np.load(os.path.join(ROOT, "tests", "golden", "ik_synth.npz"))["adjacency"]
    for v in range(o.V):
        row = adj[v]
        o.set_adjacency(v, row[row >= 0])
    out = {"keys": np.array(keys)}
    for k in keys:
        m = re.fullmatch(r"n(\d+)_K(\d+)_(\w+)", k)
        n, K, mode = int(m.group(1)), int(m.group(2)), m.group(3)
        c = S.make_case(n, K, mode)
        f = c["f"]
        r = R.ik_eval(c["beta"][f], c["theta"][f], c["faces"][f], c["tp"][f], c["tn"][f], c["pw"][f], c["nw"][f], c["pl"][f], c["noff"][f],
                      np.full((K, 3), 1 / 3, np.float32), optimize_beta=c["ob"])
        oc = S.oracle_eval(o, c)
        dp, dn = S.deviations(r["J"], oc["J"], c)
        de = float(np.abs(r["e"] - oc["e"]).max())
        out[k + "/frame"] = np.int64(f)
        out[k + "/ref_e"], out[k + "/ref_J"] = r["e"], r["J"].astype(np.float32)  # (fp32 autograd gradients, cast as node.cpp:830-831 does)
        out[k + "/oracle_dev"] = np.array([de, dp, dn])
        print("%-22s frame %3d: oracle vs reference autograd: de %.3g, position-class rows %.3g, normal-class rows %.3g" % (k, f, de, dp, dn), flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ik_outliers.npz"), **out)


if __name__ == "__main__":
    main(sys.argv[1:])
