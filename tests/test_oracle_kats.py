"""The oracle (oracle/smpl_oracle.c) against the known-answer vectors the reference keeps in
src/toolbox/Tester.cpp (numpy seed 0 inputs, expected values printed to 6 decimals in its comments)."""
import json
import os

import numpy as np
import pytest

from oracle import cpu

from conftest import GOLDEN

TOL = 2e-6  # the KATs are printed with 6 decimals


@pytest.fixture(scope="module")
def kats():
    with open(os.path.join(GOLDEN, "tester_kats.json")) as f:
        return json.load(f)


def test_blend_shape_kat(kats):
    k = kats["blendShape"]
    i, e = k["inputs"], k["expected"]
    S = np.array(i["shapeBlendBasis"], np.float32)  # (1,3,10): one vertex
    P = np.array(i["poseBlendBasis"], np.float32)  # (1,3,207)
    bs, bp, rot = cpu.blend_shape(i["beta"], i["theta"], S, P)
    np.testing.assert_allclose(bs.reshape(1, 3), np.array(e["shapeBlendShape"]), atol=TOL)
    np.testing.assert_allclose(bp.reshape(1, 1, 3), np.array(e["poseBlendShape"]), atol=3 * TOL)
    np.testing.assert_allclose(rot[0, :5], np.array(e["poseRotation"]), atol=TOL)
    np.testing.assert_allclose(cpu.rodrigues(i["theta"])[0, :5], np.array(e["poseRotation"]), atol=TOL)


def test_joint_regression_kat(kats):
    k = kats["jointRegression"]
    i, e = k["inputs"], k["expected"]
    rest, joints = cpu.joint_regression(i["templateShape"], i["jointRegressor"], i["shapeBlendShape"], i["poseBlendShape"])
    np.testing.assert_allclose(rest, np.array(e["restShape"]), atol=TOL)
    np.testing.assert_allclose(joints[0], np.array(e["joints"]), atol=2 * TOL)


def test_world_transformation_kat(kats):
    k = kats["worldTransformation"]
    i, e = k["inputs"], k["expected"]
    kt = np.array(i["kineTree"], np.int64)
    from smplpp_amd import model_io

    assert (kt == model_io.KINEMATIC_TREE).all()  # "exactly the kinematic tree of SMPL" (Tester.cpp:710)
    out = cpu.world_transformation(kt, i["joints"], i["poseRotation"])
    np.testing.assert_allclose(out[0, :5], np.array(e["transformations"]), atol=3 * TOL)


def test_lbs_kat(kats):
    """1 vertex, 24 random 4x4 'transformations' with a non-trivial last row: only matches WITH the homogeneous
    divide (src/LinearBlendSkinning.cpp:545-550).  The KAT predates rootPos, so none is added."""
    k = kats["linearBlendSkinning"]
    i, e = k["inputs"], k["expected"]
    out = cpu.lbs(np.array(i["weights"]), np.array(i["restShape"]), np.array(i["transformations"]), None)
    np.testing.assert_allclose(out, np.array(e["vertices"]), atol=TOL)
