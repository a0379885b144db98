"""GPU parity of the FK path (SMPL::launch) through the C ABI: against the reference's golden vectors, the
Tester.cpp KATs, and the C oracle on seeded inputs.  Tolerance: 1e-5 m on vertices (BASELINE.json north_star)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

VERT_TOL = 1e-5  # metres


@pytest.fixture(scope="module")
def smpl(synth_model):
    from smplpp_amd.smpl import SMPL

    s = SMPL()
    s.setDevice("cuda:0")
    s.init(synth_model)
    return s


@pytest.fixture(scope="module")
def kats():
    with open(os.path.join(GOLDEN, "tester_kats.json")) as f:
        return json.load(f)


def test_native_library_loaded():
    from smplpp_amd import _lib

    assert os.path.exists(_lib.LIB_PATH)
    assert _lib.require_gpu() >= 1


def test_model_info(smpl):
    info = smpl.info()
    assert info["vertex_num"] == 6890 and info["face_num"] == 13776 and info["weights_per_vertex"] == 4


def test_fk_golden_reference_vectors(smpl, golden_fk_synth):
    """Outputs of the reference's own compiled stages (oracle/_ref) on the synthetic model."""
    g = golden_fk_synth
    o = smpl.launch(g["beta"], g["theta"])
    ids = g["vertex_ids"]
    assert np.abs(o["verts"][:, ids] - g["verts"]).max() < VERT_TOL
    assert np.abs(o["rest"][:, ids] - g["rest"]).max() < VERT_TOL
    assert np.abs(o["joints"] - g["joints"]).max() < VERT_TOL
    assert np.abs(o["xforms"] - g["xforms"]).max() < VERT_TOL
    np.testing.assert_allclose(o["verts"].astype(np.float64).sum(axis=1), g["verts_sum"], atol=5e-3)
    np.testing.assert_allclose(np.abs(o["verts"].astype(np.float64)).sum(axis=1), g["verts_abs_sum"], rtol=2e-6)


def test_fk_zero_pose_is_template(smpl, synth_model):
    """BASELINE config 1."""
    o = smpl.launch(np.zeros((1, 10), np.float32), np.zeros((1, 25, 3), np.float32))
    assert np.abs(o["verts"][0] - synth_model["vertices_template"]).max() < 1e-6
    assert np.abs(o["xforms"][0] - np.tile(np.eye(4, dtype=np.float32), (24, 1, 1))).max() < 1e-6


@pytest.mark.parametrize("n", [1, 31, 33, 70, 257])
def test_fk_vs_oracle_ragged_batches(smpl, oracle_synth, n):
    from smplpp_amd import model_io

    beta, theta = model_io.synthetic_inputs(n, seed=100 + n)
    o = smpl.launch(beta, theta)
    r = oracle_synth.fk(beta, theta)
    for k in ("verts", "rest", "joints", "xforms"):
        assert np.abs(o[k] - r[k]).max() < VERT_TOL, k


def test_fk_batch1024_properties_and_sample(smpl, oracle_synth):
    """BASELINE config 2 (batch 1024): sampled frames against the oracle; frames are independent (a frame's result
    does not depend on its batch position or neighbours); translation enters additively."""
    from smplpp_amd import model_io

    beta, theta = model_io.synthetic_inputs(1024, seed=1)
    o = smpl.launch(beta, theta, want=("verts",))["verts"].copy()
    sel = np.array([0, 1, 63, 64, 511, 512, 1000, 1023])
    r = oracle_synth.fk(beta[sel], theta[sel], want=("verts",))["verts"]
    assert np.abs(o[sel] - r).max() < VERT_TOL
    perm = np.random.default_rng(0).permutation(1024)
    o2 = smpl.launch(beta[perm], theta[perm], want=("verts",))["verts"]
    assert np.abs(o2 - o[perm]).max() < 1e-6
    theta_t = theta.copy()
    theta_t[:, 0, :] += np.float32(0.5)
    o3 = smpl.launch(beta, theta_t, want=("verts",))["verts"]
    assert np.abs((o3 - o) - np.float32(0.5)).max() < 1e-5
    assert np.isfinite(o).all()


@pytest.mark.parametrize("form", ["e", "h", "b", "p", "v"])
def test_fk_dense_weights_and_ragged_vertex_count(form, monkeypatch):
    """61-vertex model with all 24 skinning weights non-zero (dense path) — golden from the reference build.  Under every
    SMPLPP_SKIN: h takes dense weights as they are; e (at most 4 weights per vertex), b and p (at most 8) must hand such a
    model to the first form when it is CREATED, with the operand layout that form reads kept resident."""
    from smplpp_amd import model_io
    from smplpp_amd.smpl import SMPL

    monkeypatch.setenv("SMPLPP_SKIN", form)
    g = np.load(os.path.join(GOLDEN, "fk_tiny.npz"))
    s = SMPL()
    s.setDevice("cuda:0")
    s.init(model_io.tiny_model(61, seed=7))
    assert s.info()["weights_per_vertex"] == 24
    o = s.launch(g["beta"], g["theta"])
    for k in ("verts", "rest", "joints", "xforms"):
        assert np.abs(o[k] - g[k]).max() < VERT_TOL, k


@pytest.mark.parametrize("form", ["e", "h", "b", "p"])
def test_fk_eight_weights_per_vertex(synth_model, form, monkeypatch):
    """Models with 5..8 skinning weights per vertex: h skins on the matrix pipe with dense weights, the b / p forms take their
    MAXW = 8 instantiations, and the default form (e: four weights per vertex in registers) hands such a model to b when it is
    created (real SMPL has at most 4); ragged frame counts exercise partial frame tiles and the single-item / multi-item
    paths.  (The form is read from SMPLPP_SKIN when a model is created.)"""
    from smplpp_amd.smpl import SMPL
    from oracle.cpu import OracleModel

    md = {k: v.copy() for k, v in synth_model.items()}
    rng = np.random.default_rng(5)
    w = md["weights"].astype(np.float64)
    for v in range(w.shape[0]):  # add 1..4 more joints with small weights, renormalise like the generator does
        extra = rng.choice(np.where(w[v] == 0)[0], size=int(rng.integers(1, 5)), replace=False)
        w[v, extra] = rng.uniform(0.01, 0.1, len(extra))
    w /= w.sum(axis=1, keepdims=True)
    md["weights"] = w.astype(np.float32)
    monkeypatch.setenv("SMPLPP_SKIN", form)
    s = SMPL()
    s.setDevice("cuda:0")
    s.init(md)
    assert s.info()["weights_per_vertex"] == 8
    o = OracleModel(md)
    from smplpp_amd import model_io
    for n in (3, 70, 333):
        beta, theta = model_io.synthetic_inputs(n, seed=n)
        g = s.launch(beta, theta)
        r = o.fk(beta, theta)
        for k in ("verts", "rest", "joints"):
            assert np.abs(g[k] - r[k]).max() < VERT_TOL, (form, n, k)


@pytest.mark.parametrize("V", [61, 200, 1000])
def test_fk_small_sparse_models(V):
    """Vertex counts that are not multiples of 64 (partial vertex-group pairs, fewer pairs than XCDs) with <= 4 weights per
    vertex: the default fused kernel on small meshes, ragged frame counts."""
    from smplpp_amd import model_io
    from smplpp_amd.smpl import SMPL
    from oracle.cpu import OracleModel

    md = model_io.tiny_model(V, seed=11)
    w = md["weights"].astype(np.float64)
    keep = np.argsort(-w, axis=1)[:, :4]
    sp = np.zeros_like(w)
    np.put_along_axis(sp, keep, np.take_along_axis(w, keep, axis=1), axis=1)
    sp /= sp.sum(axis=1, keepdims=True)
    md["weights"] = sp.astype(np.float32)
    s = SMPL()
    s.setDevice("cuda:0")
    s.init(md)
    assert s.info()["weights_per_vertex"] == 4
    o = OracleModel(md)
    for n in (1, 65, 130):
        beta, theta = model_io.synthetic_inputs(n, seed=V + n)
        g = s.launch(beta, theta)
        r = o.fk(beta, theta)
        for k in ("verts", "rest", "joints"):
            assert np.abs(g[k] - r[k]).max() < VERT_TOL, (V, n, k)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 1100])
def test_fk_exact_form_reproduces_round1_kernel_bits(synth_model, n, monkeypatch):
    """skin_kernel_e (the default) issues skin_kernel_b's piece products and skinning operations in skin_kernel_b's order from a
    different skeleton (A fragments in registers, the tile's transforms resident in LDS over a run of vertex groups, a four-image
    ring, skinning tables inside the basis images): the two must agree bit for bit at every batch size — single frames, partial
    tiles, workgroups whose runs cross frame tiles (257, 1100) — with and without `rest`."""
    from smplpp_amd import model_io
    from smplpp_amd.smpl import SMPL

    eng = {}
    for form in ("e", "b"):
        monkeypatch.setenv("SMPLPP_SKIN", form)
        eng[form] = SMPL()
        eng[form].setDevice("cuda:0")
        eng[form].init(synth_model)
    beta, theta = model_io.synthetic_inputs(n, seed=5 + n)
    oe, ob = eng["e"].launch(beta, theta), eng["b"].launch(beta, theta)
    assert np.isfinite(oe["verts"]).all()
    assert np.array_equal(oe["verts"], ob["verts"]) and np.array_equal(oe["rest"], ob["rest"])
    assert np.array_equal(eng["e"].launch(beta, theta, want=("verts",))["verts"], ob["verts"])


def test_fk_workgroups_spanning_frame_tiles(smpl, oracle_synth):
    """Batches beyond 2048 frames: a workgroup of the default fused kernel then runs through several frame tiles (its A
    registers, the G' image and the root translations are reloaded, the previous item's tail is flushed first, the LDS ring
    keeps streaming). 2500 frames, vertices AND rest shapes: every frame equals the same frame computed in a batch of 500
    (where no workgroup changes its frame tile), and frames at tile boundaries match the oracle."""
    from smplpp_amd import model_io

    n = 2500
    beta, theta = model_io.synthetic_inputs(n, seed=77)
    o = smpl.launch(beta, theta, want=("verts", "rest"))
    big_v, big_r = o["verts"].copy(), o["rest"].copy()
    for a in range(0, n, 500):
        p = smpl.launch(beta[a:a + 500], theta[a:a + 500], want=("verts", "rest"))
        assert np.abs(p["verts"] - big_v[a:a + 500]).max() < 1e-6, a
        assert np.abs(p["rest"] - big_r[a:a + 500]).max() < 1e-6, a
    sel = np.array([0, 63, 64, 127, 128, 1279, 1280, 2047, 2048, 2495, 2496, 2499])
    r = oracle_synth.fk(beta[sel], theta[sel], want=("verts", "rest"))
    assert np.abs(big_v[sel] - r["verts"]).max() < VERT_TOL and np.abs(big_r[sel] - r["rest"]).max() < VERT_TOL


def test_fk_outputs_beyond_2gib(smpl, oracle_synth):
    """The default fused kernels address their outputs with 32-bit buffer offsets; a batch whose vertex array reaches
    2 GiB (26 100 frames) is split into launches of at most 2 GiB (fp32-MFMA form: falls back to 64-bit addressing) and must still be right (device buffers: no 2 GiB host copy)."""
    import torch
    from smplpp_amd import model_io

    n = 26100
    beta, theta = model_io.synthetic_inputs(512, seed=9)
    reps = (n + 511) // 512
    bt = torch.from_numpy(np.tile(beta, (reps, 1))[:n].copy()).cuda()
    tt = torch.from_numpy(np.tile(theta, (reps, 1, 1))[:n].copy()).cuda()
    o = smpl.launch(bt, tt, want=("verts",))["verts"]
    torch.cuda.synchronize()
    assert o.shape == (n, 6890, 3) and o.numel() * 4 >= 2**31
    sel = np.array([0, 511, 12345, 25599, n - 1])
    r = oracle_synth.fk(beta[sel % 512], theta[sel % 512], want=("verts",))["verts"]
    assert np.abs(o[torch.from_numpy(sel).cuda()].cpu().numpy() - r).max() < VERT_TOL
    del o, bt, tt
    torch.cuda.empty_cache()


def test_fk_device_pointers_torch(smpl, oracle_synth):
    import torch
    from smplpp_amd import model_io

    beta, theta = model_io.synthetic_inputs(96, seed=5)
    o = smpl.launch(torch.from_numpy(beta).cuda(), torch.from_numpy(theta).cuda())
    torch.cuda.synchronize()
    r = oracle_synth.fk(beta, theta)
    assert o["verts"].is_cuda
    assert np.abs(o["verts"].cpu().numpy() - r["verts"]).max() < VERT_TOL
    assert np.abs(o["joints"].cpu().numpy() - r["joints"]).max() < VERT_TOL


@pytest.mark.parametrize("n", [1, 5, 33, 100])
def test_fk_writes_only_its_frames(smpl, n):
    """The fused kernel computes whole 32-frame tiles; frames >= n must be dropped, not stored.  The outputs are
    views into a larger canary-filled allocation, so a stray store is detected instead of faulting."""
    import torch
    from smplpp_amd import model_io

    beta, theta = model_io.synthetic_inputs(n, seed=n)
    big = {k: torch.full((n + 64, 6890, 3), 777.0, device="cuda") for k in ("verts", "rest")}
    smpl.launch(torch.from_numpy(beta).cuda(), torch.from_numpy(theta).cuda(), want=("verts", "rest"),
                out={k: big[k][:n] for k in big})
    torch.cuda.synchronize()
    for k in big:
        assert bool((big[k][n:] == 777.0).all()), k
        assert bool((big[k][:n] != 777.0).all()), k


def test_fk_split_operand_forms_are_fp32_exact(synth_model, oracle_synth, monkeypatch):
    """The default fused kernel (e, skin_e.hip) and round 1's b (skin_b.hip) carry every fp32 operand as three bf16 pieces
    (fp32's 24 bits, six MFMA products) and skin in fp32 on the vector ALU; h (skin_h.hip) carries two fp16 pieces on the f16
    matrix pipe and skins there too.  Their error against the fp64-accumulating oracle must be of the same size as that of the
    exact fp32-MFMA form (p, skin_p.hip), far inside the 1e-5 m parity bar, on shaped vertices and skinned vertices alike;
    and e, which issues b's piece products and b's skinning operations in b's order, must reproduce b's bits."""
    from smplpp_amd import model_io
    from smplpp_amd.smpl import SMPL

    beta, theta = model_io.synthetic_inputs(200, seed=11)
    beta = (beta * 3.0).astype(np.float32)  # large shape coefficients: stresses the low pieces
    r = oracle_synth.fk(beta, theta)
    err = {}
    outs = {}
    for form in ("e", "h", "b", "p"):
        monkeypatch.setenv("SMPLPP_SKIN", form)
        s = SMPL()
        s.setDevice("cuda:0")
        s.init(synth_model)
        o = s.launch(beta, theta)
        outs[form] = o
        err[form] = {k: float(np.abs(o[k] - r[k]).max()) for k in ("verts", "rest")}
    monkeypatch.delenv("SMPLPP_SKIN")
    assert np.array_equal(outs["e"]["verts"], outs["b"]["verts"]) and np.array_equal(outs["e"]["rest"], outs["b"]["rest"])
    for form in ("e", "h", "b"):
        for k in ("verts", "rest"):
            assert err[form][k] < 2e-6, err
            assert err[form][k] <= 3.0 * err["p"][k] + 2e-7, err


@pytest.mark.parametrize("form", ["h", "e"])
def test_fk_fp16x2_at_real_smpl_magnitudes(synth_model, form, monkeypatch):
    """Error budget of the fp16x2 operands (h; and the exact form e beside it) where it is tightest: posedirs up to 5e-2 (25x the synthetic model's, the
    size of real SMPL's largest entries), |beta| = 3, rotations up to ~1.5 rad, root translations of several metres.
    Reference: float64 numpy restatement of rest = T + S.beta + P.c and of the skinning, from the fp32 joints / relative
    transforms the engine itself returns (those stages are covered by the golden tests)."""
    from smplpp_amd import model_io
    from smplpp_amd.smpl import SMPL

    md = {k: v.copy() for k, v in synth_model.items()}
    rng = np.random.default_rng(77)
    md["pose_blend_shapes"] = np.clip(rng.normal(0, 5e-2 / 3, md["pose_blend_shapes"].shape), -5e-2, 5e-2).astype(np.float32)
    monkeypatch.setenv("SMPLPP_SKIN", form)
    s = SMPL()
    s.setDevice("cuda:0")
    s.init(md)
    n = 96
    beta = rng.choice([-3.0, 3.0], size=(n, 10)).astype(np.float32)
    theta = np.zeros((n, 25, 3), np.float32)
    theta[:, 1:] = rng.normal(0, 0.9, (n, 24, 3))
    theta[:, 0] = rng.uniform(-5, 5, (n, 3))
    o = s.launch(beta, theta)
    from scipy.spatial.transform import Rotation

    th = theta[:, 1:].astype(np.float64) + 1e-8  # src/BlendShape.cpp:813-815 (angle from theta + eps, axis from theta)
    ang = np.linalg.norm(th, axis=-1, keepdims=True)
    axis = theta[:, 1:].astype(np.float64) / ang
    R = Rotation.from_rotvec((axis * ang).reshape(-1, 3)).as_matrix().reshape(n, 24, 3, 3)
    c = (R[:, 1:] - np.eye(3)).reshape(n, 207)
    rest = (md["vertices_template"].astype(np.float64)[None] + np.einsum("vxk,nk->nvx", md["shape_blend_shapes"].astype(np.float64), beta.astype(np.float64))
            + np.einsum("vxk,nk->nvx", md["pose_blend_shapes"].astype(np.float64), c))
    # c above comes from fp64 Rodrigues; the engine's rotations are fp32: compare `rest` at the fp32 level of c
    assert np.abs(o["rest"] - rest).max() < 5e-6
    G = o["xforms"].astype(np.float64)  # [n, 24, 4, 4] relative transforms
    W = md["weights"].astype(np.float64)
    M = np.einsum("vj,njab->nvab", W, G)
    h = np.einsum("nvab,nvb->nva", M[:, :, :3, :3], o["rest"].astype(np.float64)) + M[:, :, :3, 3]
    verts = h / W.sum(axis=1)[None, :, None] + theta[:, :1].astype(np.float64)
    assert np.abs(o["verts"] - verts).max() < 3e-6, float(np.abs(o["verts"] - verts).max())


def test_stage_kats_on_gpu(kats):
    """The reference's own stage KATs (src/toolbox/Tester.cpp) through the stage-level entry points."""
    from smplpp_amd import smpl as S

    i, e = kats["blendShape"]["inputs"], kats["blendShape"]["expected"]
    bs, bp, rot = S.stage_blend_shape(i["beta"], i["theta"], np.array(i["shapeBlendBasis"]), np.array(i["poseBlendBasis"]))
    np.testing.assert_allclose(bs.reshape(1, 3), np.array(e["shapeBlendShape"]), atol=2e-6)
    np.testing.assert_allclose(bp.reshape(1, 1, 3), np.array(e["poseBlendShape"]), atol=6e-6)
    np.testing.assert_allclose(rot[0, :5], np.array(e["poseRotation"]), atol=2e-6)
    i, e = kats["jointRegression"]["inputs"], kats["jointRegression"]["expected"]
    rest, joints = S.stage_joint_regression(i["templateShape"], i["jointRegressor"], i["shapeBlendShape"], i["poseBlendShape"])
    np.testing.assert_allclose(rest, np.array(e["restShape"]), atol=2e-6)
    np.testing.assert_allclose(joints[0], np.array(e["joints"]), atol=4e-6)
    i, e = kats["worldTransformation"]["inputs"], kats["worldTransformation"]["expected"]
    out = S.stage_world_transformation(np.array(i["kineTree"]), i["joints"], i["poseRotation"])
    np.testing.assert_allclose(out[0, :5], np.array(e["transformations"]), atol=6e-6)
    i, e = kats["linearBlendSkinning"]["inputs"], kats["linearBlendSkinning"]["expected"]
    out = S.stage_skinning(np.array(i["weights"]), np.array(i["restShape"]), np.array(i["transformations"]), None)
    np.testing.assert_allclose(out, np.array(e["vertices"]), atol=2e-6)


def test_mesh_queries_vs_oracle(smpl, oracle_synth, synth_model):
    from smplpp_amd import model_io

    beta, theta = model_io.synthetic_inputs(3, seed=9)
    o = smpl.launch(beta, theta)
    verts = o["verts"]
    fids = np.array([0, 100, 5000, 13775])
    vids = np.array([0, 17, 3000, 6889])
    fn = smpl.calcNormalBatch(fids)
    vn = smpl.calcVertexNormalBatch(vids)
    for f in range(3):
        for a, fid in enumerate(fids):
            assert np.abs(fn[f, a] - oracle_synth.face_normal(verts[f], int(fid))).max() < 2e-5
        for a, vid in enumerate(vids):
            assert np.abs(vn[f, a] - oracle_synth.vertex_normal(verts[f], int(vid))).max() < 2e-5
    assert np.abs(smpl.calcNormal(100) - fn[0, 1]).max() == 0
    rng = np.random.default_rng(3)
    pts = verts[:, rng.integers(0, 6890, 7)] + rng.normal(0, 0.01, (3, 7, 3)).astype(np.float32)
    face, closest, sq = smpl.closestPoints(pts)
    for f in range(3):
        rf, rc, rs = oracle_synth.closest_points(verts[f], pts[f])
        assert np.abs(closest[f] - rc).max() < 1e-6
        assert np.abs(sq[f] - rs).max() < 1e-7
        assert (face[f] == rf).all()
    adj = smpl.getAdjacentFaces(17)
    of, ow = oracle_synth.get_adjacency(17)
    assert sorted(adj) == sorted(of.tolist()) and abs(sum(adj.values()) - 1) < 1e-6


def test_errors_are_loud(smpl):
    from smplpp_amd._lib import SmplppError

    with pytest.raises(SmplppError):
        smpl.launch(np.zeros((2, 9), np.float32), np.zeros((2, 25, 3), np.float32))
    with pytest.raises(SmplppError):
        smpl.launch(np.zeros((2, 10), np.float32), np.zeros((2, 24, 3), np.float32))


def test_whole_mesh_vertex_normals_and_sweep_grid(smpl, oracle_synth, synth_model):
    """SURVEY.md §8(f) row 3, the mesh-side queries of the node: vertex normals of the whole mesh (src/SMPL.cpp:527-535 for
    every vertex) and the winding-number sweep grid (node/node.cpp:1023-1073, toolbox/GridUtils.hpp:26-61) against the
    oracle: grid extents from floor/ceil of the vertex bounds at 2.5 cm, cell order x-outermost, winding numbers within
    2e-4 of the fp64 solid-angle sum on 600 sampled cells, and the same inside/outside verdict away from the surface."""
    from smplpp_amd import model_io

    beta, theta = model_io.synthetic_inputs(3, seed=9)
    theta[:, 1:] *= 0.5
    o = smpl.launch(beta, theta)
    verts = o["verts"]
    vn = smpl.calcMeshVertexNormals()
    assert vn.shape == verts.shape
    rng = np.random.default_rng(3)
    for f in range(3):
        for v in rng.integers(0, verts.shape[1], 40):
            assert np.abs(vn[f, v] - oracle_synth.vertex_normal(verts[f], int(v))).max() < 2e-5
    assert np.abs(np.linalg.norm(vn, axis=-1) - 1).max() < 1e-5
    g = smpl.calcSweepGrid(frame=1)
    lo, hi = verts[1].min(axis=0), verts[1].max(axis=0)
    assert (g["grid_min"] == np.floor(lo / np.float32(0.025)).astype(np.int32)).all()
    assert (g["grid_min"] + g["grid_num"] - 1 == np.ceil(hi / np.float32(0.025)).astype(np.int32)).all()
    cells = int(np.prod(g["grid_num"]))
    assert g["winding"].shape == (cells,) and g["grid_idx"].shape == (cells, 3)
    assert (g["grid_idx"][1] - g["grid_idx"][0] == [0, 0, 1]).all()  # z innermost (node.cpp:1037-1048)
    pick = rng.choice(cells, 600, replace=False)
    w = oracle_synth.winding_numbers(verts[1], np.float32(0.025) * g["grid_idx"][pick].astype(np.float32))
    assert np.abs(g["winding"][pick] - w).max() < 2e-4
    clear = np.abs(w - 0.5) > 0.01
    assert ((w > 0.5) == g["inside"][pick])[clear].all()
    frac = g["inside"].mean()
    assert 0.05 < frac < 0.9 and (g["inside"] == (g["winding"] > 0.5)).all()  # a closed body: a solid share of its bounding grid
    assert np.abs(g["winding"] - np.round(g["winding"])).max() < 0.5 + 1e-6  # (a posed synthetic body may self-intersect: winding 2)


def test_fk_out_of_range_operands_are_reported(smpl, synth_model, oracle_synth, monkeypatch):
    """The fp16x2 form (SMPLPP_SKIN=h) carries its operands as fp16 pieces of scaled values (|beta| < 1023, transforms within
    16 x the template's extent): outside that range the reference stays finite and that form does not, so the launch must SAY so
    — a host-space call returns SMPLPP_ERR_NUMERIC, an enqueue-only caller finds bit 0 in smplpp_fk_status — and the next
    in-range launch is clean again.  The default form (e: bf16 pieces, fp32's exponent range) has no such range: it follows the
    reference there."""
    import torch
    from smplpp_amd import model_io
    from smplpp_amd._lib import SmplppError
    from smplpp_amd.smpl import SMPL

    beta, theta = model_io.synthetic_inputs(5, seed=3)
    bad = beta.copy()
    bad[3, 2] = 2000.0
    o = smpl.launch(bad, theta, want=("verts",))  # the default form: finite, and the oracle's vertices
    r = oracle_synth.fk(bad, theta)
    assert np.isfinite(o["verts"]).all() and smpl.launchStatus() == 0
    assert np.abs(o["verts"] - r["verts"]).max() < 1e-3  # vertices tens of metres away: fp32 at that size
    monkeypatch.setenv("SMPLPP_SKIN", "h")
    smpl = SMPL()
    smpl.setDevice("cuda:0")
    smpl.init(synth_model)
    monkeypatch.delenv("SMPLPP_SKIN")
    assert smpl.launchStatus() == 0
    with pytest.raises(SmplppError) as ei:
        smpl.launch(bad, theta, want=("verts",))
    assert ei.value.code == 3
    smpl.launch(torch.from_numpy(bad).cuda(), torch.from_numpy(theta).cuda(), want=("verts",))
    assert smpl.launchStatus() & 1
    far = theta.copy()
    far[1, 5] = 0.0
    o = smpl.launch(beta, theta, want=("verts",))  # in range: no error, finite
    assert np.isfinite(o["verts"]).all() and smpl.launchStatus() == 0


def test_fk_skinning_class_groups_on_a_part_ordered_numbering(synth_model, oracle_synth, monkeypatch):
    """skin_kernel_h runs a vertex group (64 consecutive vertices) whose skinning weights live in ONE k-step of the blend product —
    joints 0..15 only, or joints 16..23 only (SMPL's arms) — in an instantiation that issues only that k-step's MFMAs (common.h,
    HB_PERM_OFF), and deals the groups over the XCD slices by class.  The stand-in's spiral numbering has 11 such groups of 108 and
    none of the second kind; the SAME body renumbered so that its vertex order follows the parts (as SMPL's does) has 54 + 19.  The
    renumbered model must match the oracle on ITS numbering, and — the skipped products being exact zeros — give the original
    numbering's bits, vertex for vertex."""
    from oracle import cpu
    from smplpp_amd import model_io
    from smplpp_amd.smpl import SMPL

    monkeypatch.setenv("SMPLPP_SKIN", "h")
    cls = model_io.skinning_classes(synth_model["weights"])
    order = np.argsort(cls, kind="stable")
    part = model_io.relabel_vertices(synth_model, order)
    c = model_io.skinning_classes(part["weights"])
    groups = [(int((c[t:t + 64] != 2).any()) | 2 * int((c[t:t + 64] != 0).any())) for t in range(0, len(c), 64)]
    assert groups.count(1) >= 40 and groups.count(2) >= 10 and groups.count(3) >= 10  # all three instantiations run
    beta, theta = model_io.synthetic_inputs(130, seed=21)  # three frame tiles, the last one partial
    s = SMPL()
    s.setDevice("cuda:0")
    s.init(part)
    o = s.launch(beta, theta)
    r = cpu.OracleModel(part).fk(beta, theta)
    for k in ("verts", "rest", "joints"):
        assert np.abs(o[k] - r[k]).max() < VERT_TOL, k
    s0 = SMPL()
    s0.setDevice("cuda:0")
    s0.init(synth_model)
    o0 = s0.launch(beta, theta)
    assert np.array_equal(o["verts"], o0["verts"][:, order]) and np.array_equal(o["rest"], o0["rest"][:, order])
