"""Mocap-side formats: C3D reader/writer round trip, Baseline-41 label matching, result writers."""
import os

import numpy as np
import pytest

from smplpp_amd import mocap

from conftest import GOLDEN


def test_c3d_roundtrip_with_missing_markers(tmp_path):
    rng = np.random.default_rng(0)
    names = sorted(mocap.BASELINE41)
    labels = ["Skeleton0:" + n for n in names] + ["Unlabeled_1", "Unlabeled_2"]
    pts = rng.normal(0, 1, (37, len(labels), 3)).astype(np.float32)
    valid = rng.random((37, len(labels))) > 0.1
    p = str(tmp_path / "t.c3d")
    mocap.write_c3d(p, labels, pts, valid, rate=120.0)
    c = mocap.read_c3d(p)
    assert c["labels"] == labels and c["rate"] == 120.0 and c["points"].shape == pts.shape
    assert np.array_equal(c["valid"], valid)
    assert np.array_equal(c["points"][valid], pts[valid])
    idx = mocap.match_markers(c["labels"], names)  # suffix match, node.cpp:587-593
    assert idx == list(range(len(names)))
    with pytest.raises(KeyError):
        mocap.match_markers(c["labels"], ["NoSuchMarker"])


def test_reference_sample_excerpt_matches_reader():
    """tests/golden/sample_walk_excerpt.npz was cut from the reference's data/sample_walk.c3d with this reader; when the
    reference tree is present the reader is re-run on the original file."""
    g = np.load(os.path.join(GOLDEN, "sample_walk_excerpt.npz"))
    assert g["points"].shape == (32, 41, 3) and int(g["n_frames"]) == 3163 and int(g["n_points"]) == 49
    assert int(g["missing_labelled"]) == 4420 and int(g["frames_any_missing"]) == 619
    assert (g["valid"][16:].sum(axis=1) < 20).any()  # the excerpt contains frames that must skip the solve (node.cpp:785)
    path = "/root/reference/data/sample_walk.c3d"
    if os.path.exists(path):
        c = mocap.read_c3d(path)
        idx = mocap.match_markers(c["labels"], list(g["task_names"]))
        assert np.array_equal(c["points"][g["frame_ids"]][:, idx], g["points"])
        assert np.array_equal(c["valid"][g["frame_ids"]][:, idx], g["valid"])
        assert c["rate"] == 120.0


def test_reference_sample_full_fixture_matches_reader():
    """tests/golden/sample_walk_full.npz (every frame of data/sample_walk.c3d, made by tools/make_sample_walk_fixture.py):
    the counts SURVEY.md §8d quotes for config 4, the excerpt is a subset, and — when the reference tree is present — the
    reader on the original file gives the same arrays."""
    g = np.load(os.path.join(GOLDEN, "sample_walk_full.npz"))
    assert g["points"].shape == (3163, 41, 3) and g["valid"].shape == (3163, 41)
    assert int((~g["valid"]).sum()) == 4420 and int((~g["valid"]).any(axis=1).sum()) == 619
    assert list(g["task_names"]) == sorted(mocap.BASELINE41)  # std::map order (node/node.cpp:47, 798)
    nv = g["valid"].sum(axis=1)
    assert int((nv < 20).sum()) == 60 and int((nv == 0).sum()) == 60  # the frames node.cpp:785 skips
    e = np.load(os.path.join(GOLDEN, "sample_walk_excerpt.npz"))
    assert np.array_equal(g["valid"][e["frame_ids"]], e["valid"])
    assert np.array_equal(g["points"][e["frame_ids"]][e["valid"]], e["points"][e["valid"]])
    path = "/root/reference/data/sample_walk.c3d"
    if os.path.exists(path):
        c = mocap.read_c3d(path)
        idx = mocap.match_markers(c["labels"], list(g["task_names"]))
        assert np.array_equal(c["valid"][:, idx], g["valid"])
        assert np.array_equal(c["points"][:, idx][g["valid"]], g["points"][g["valid"]])


def test_baseline41_table():
    assert len(mocap.BASELINE41) == 41 and mocap.BASELINE41["HeadTop"] == 7324 and mocap.BASELINE41["RHeel"] == 12705
    assert max(mocap.BASELINE41.values()) < 13776


def test_result_writers(tmp_path):
    th = np.arange(2 * 75, dtype=np.float32).reshape(2, 25, 3) / 7
    p = str(tmp_path / "m.txt")
    mocap.write_motion_text(p, th)
    rows = [list(map(float, ln.split())) for ln in open(p)]
    assert len(rows) == 2 and len(rows[0]) == 75 and np.allclose(rows, th.reshape(2, 75).astype(np.float64))
    y = str(tmp_path / "MocapBody.yaml")
    names = ["Chest", "HeadTop"]
    mocap.write_mocap_body_yaml(y, np.linspace(-1, 1, 10), names, [6842, 7324], [[0.2, 0.3, 0.5], [1 / 3, 1 / 3, 1 / 3]])
    beta, n2, faces, w = mocap.read_mocap_body_yaml(y)
    assert n2 == names and faces.tolist() == [6842, 7324] and np.allclose(beta, np.linspace(-1, 1, 10), atol=1e-6)
    assert np.allclose(w, [[0.2, 0.3, 0.5], [1 / 3, 1 / 3, 1 / 3]], atol=1e-6)


def test_cpp_side_of_the_formats_reads_what_python_reads(tmp_path):
    """include/smplpp/Mocap.h (the C++ side of the capture path's on-disk formats: what node.cpp takes from ezc3d at :580-594 /
    :667-690, /tmp/MocapBody.yaml of :1418-1441 / :509-534, the motion text of scripts/convertRosbagToText.py) against the Python
    side on the same files: a float C3D with missing markers and prefixed labels (and, when the reference tree is present, the
    reference's own data/sample_walk.c3d), frame by frame; the yaml the C++ writer makes is read by the Python reader and by the
    C++ reader; the motion text by numpy."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "mocap_formats")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "mocap_formats.cpp"), "-o", exe])
    rng = np.random.default_rng(3)
    names = sorted(mocap.BASELINE41)
    labels = ["Skeleton0:" + n for n in names] + ["Unlabeled_1", "Unlabeled_2"]
    pts = rng.normal(0, 1, (23, len(labels), 3)).astype(np.float32)
    valid = rng.random((23, len(labels))) > 0.15
    valid[5] = False  # a frame without any marker
    p = str(tmp_path / "t.c3d")
    mocap.write_c3d(p, labels, pts, valid, rate=120.0)
    files = [p] + (["/root/reference/data/sample_walk.c3d"] if os.path.exists("/root/reference/data/sample_walk.c3d") else [])
    for path in files:
        dump, yml, txt = str(tmp_path / "dump.txt"), str(tmp_path / "body.yaml"), str(tmp_path / "motion.txt")
        r = subprocess.run([exe, path, dump, yml, txt], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
        assert r.returncode == 0 and "OK" in r.stdout, r.stdout
        c = mocap.read_c3d(path)
        lines = open(dump).read().splitlines()
        head = lines[0].split()
        assert float(head[1]) == c["rate"] and int(head[3]) == c["points"].shape[0] and int(head[5]) == c["points"].shape[1]
        assert [ln[6:] for ln in lines if ln.startswith("label ")] == c["labels"]
        fr = [ln.split() for ln in lines if ln.startswith("frame ")]
        assert len(fr) == c["points"].shape[0]
        w = np.array([1.0, 2.0, 3.0])
        for t, (_, ti, nv, sm) in enumerate(fr):
            v = c["valid"][t]
            assert int(ti) == t and int(nv) == int(v.sum())
            assert abs(float(sm) - float((c["points"][t][v].astype(np.float64) @ w).sum())) < 1e-6 * max(1.0, abs(float(sm)))
        got = {ln.split()[1]: int(ln.split()[2]) for ln in lines if ln.startswith("match ")}
        assert [got[n] for n in names] == mocap.match_markers(c["labels"], names)
        assert "yaml_roundtrip 1" in lines
        beta, ynames, faces, weights = mocap.read_mocap_body_yaml(yml)  # the Python reader on the C++ writer's file
        assert ynames == names and [int(f) for f in faces] == [mocap.BASELINE41[n] for n in names]
        assert np.abs(beta - (0.1 * (np.arange(10) - 4) + 1e-7 * np.arange(10)).astype(np.float32)).max() < 1e-7
        assert np.abs(weights.sum(axis=1) - 1).max() < 1e-6
        th = np.loadtxt(txt)
        assert th.shape == (3, 75) and np.abs(th - (0.01 * np.arange(225, dtype=np.float32) + np.float32(1e-6)).reshape(3, 75)).max() < 1e-7
