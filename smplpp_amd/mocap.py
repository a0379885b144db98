"""Mocap-side data formats and the sequence driver of the reference's `solve_mocap_motion` mode (SURVEY.md §8(f) rows
"on-disk formats either side of the path").

* `read_c3d` — minimal C3D point reader replacing the reference's use of ezc3d (node/node.cpp:580-594, :667-690):
  Intel byte order, float or scaled-integer point data, POINT:LABELS/RATE; a negative residual marks a missing marker
  (ezc3d's `isEmpty()`).
* `BASELINE41` — the OptiTrack Baseline-41 marker -> SMPL face table of node/node.cpp:455-500.
* `match_markers` — suffix label match of node/node.cpp:583-594.
* `MocapMotionSolver` — node/node.cpp:1362-1412 for R independent restarts/sequences in lock step on one GPU:
  32 iterations on the first frame (the store block `ikIter > 30`, :1369, runs AFTER the solve of the same pass: passes
  0..31), then exactly ONE IK iteration per C3D frame, warm-started;
  missing markers get posTaskWeight_ = 0 (:674-683); frames with fewer than K/2 valid markers skip the solve (:785);
  QP on, phi limits 0, normal task off, normal offset 15 mm (:316-322, :553-567, :699).
* `MocapBodySolver` — solveMocapBody (node/node.cpp:652-656, 693-696, 1343-1352): the 51-iteration body stage that
  produces beta and the per-marker faces / weights (MocapBody.yaml).
* `write_motion_text` — scripts/convertRosbagToText.py:18-19 (one frame per line, theta 25x3 row-major).
* `write_mocap_body_yaml` / `read_mocap_body_yaml` — /tmp/MocapBody.yaml of node/node.cpp:1426-1441 and :509-534.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Sequence

import numpy as np

# node/node.cpp:455-500 (https://docs.optitrack.com/markersets/full-body/baseline-41)
BASELINE41: Dict[str, int] = {
    "HeadTop": 7324, "HeadFront": 7194, "HeadSide": 13450,
    "Chest": 6842, "WaistLFront": 2162, "WaistRFront": 13026, "WaistLBack": 5117, "WaistRBack": 12007,
    "BackTop": 8914, "BackRight": 11433, "BackLeft": 4309,
    "LShoulderTop": 2261, "LShoulderBack": 4599, "LUArmHigh": 4249, "LElbowOut": 4913, "LWristIn": 4091,
    "LWristOut": 2567, "LHandOut": 2636,
    "RShoulderTop": 13583, "RShoulderBack": 11491, "RUArmHigh": 11137, "RElbowOut": 11802, "RWristIn": 9712,
    "RWristOut": 9590, "RHandOut": 9733,
    "LThigh": 1122, "LKneeOut": 1165, "LShin": 1247, "LAnkleOut": 5742, "LToeIn": 5758, "LToeOut": 6000,
    "LToeTip": 5591, "LHeel": 5815,
    "RThigh": 8523, "RKneeOut": 8053, "RShin": 12108, "RAnkleOut": 12630, "RToeIn": 12896, "RToeOut": 12889,
    "RToeTip": 12478, "RHeel": 12705,
}  # fmt: skip


# ------------------------------------------------------------------------------------------------ C3D
def read_c3d(path: str) -> dict:
    """Points of a C3D file: dict(labels [P], rate, points [T,P,3] float32 (file units), valid [T,P] bool,
    residual [T,P], first_frame)."""
    d = open(path, "rb").read()
    if len(d) < 512 or d[1] != 0x50:
        raise ValueError("not a C3D file: %s" % path)
    param_block = d[0]
    npoints, nanalog, first, last, _gap = struct.unpack_from("<HHHHH", d, 2)
    scale, = struct.unpack_from("<f", d, 12)
    data_block, analog_per_frame = struct.unpack_from("<HH", d, 16)
    rate, = struct.unpack_from("<f", d, 20)
    p0 = (param_block - 1) * 512
    if d[p0 + 3] != 84:
        raise ValueError("only Intel (little-endian IEEE) C3D files are supported (processor type %d)" % d[p0 + 3])
    groups: Dict[int, str] = {}
    params: Dict[str, object] = {}
    pos = p0 + 4
    while True:
        name_len = struct.unpack_from("<b", d, pos)[0]
        gid = struct.unpack_from("<b", d, pos + 1)[0]
        n = abs(name_len)
        if n == 0:
            break
        name = d[pos + 2:pos + 2 + n].decode("ascii", "replace").upper()
        nxt_at = pos + 2 + n
        nxt, = struct.unpack_from("<h", d, nxt_at)
        if gid < 0:
            groups[-gid] = name
        else:
            q = nxt_at + 2
            typ = struct.unpack_from("<b", d, q)[0]
            ndim = d[q + 1]
            dims = list(d[q + 2:q + 2 + ndim])
            q += 2 + ndim
            count = int(np.prod(dims)) if dims else 1
            if typ == -1:
                raw = d[q:q + count]
                if len(dims) == 2:  # [chars per string, number of strings]
                    val = [raw[i * dims[0]:(i + 1) * dims[0]].decode("ascii", "replace").strip() for i in range(dims[1])]
                else:
                    val = raw.decode("ascii", "replace").strip()
            else:
                fmt = {1: "b", 2: "h", 4: "f"}[typ]
                val = np.array(struct.unpack_from("<%d%s" % (count, fmt), d, q))
            params["%d:%s" % (gid, name)] = val
        if nxt == 0:
            break
        pos = nxt_at + nxt
    inv = {v: k for k, v in groups.items()}

    def get(group, name, default=None):
        return params.get("%d:%s" % (inv.get(group, -1), name), default)

    labels: List[str] = list(get("POINT", "LABELS", []) or [])
    k = 2
    while get("POINT", "LABELS%d" % k) is not None:  # files with more than 255 points continue in LABELS2, ...
        labels += list(get("POINT", "LABELS%d" % k))
        k += 1
    labels = (labels + ["*%d" % i for i in range(len(labels), npoints)])[:npoints]
    r = get("POINT", "RATE")
    if r is not None:
        rate = float(np.atleast_1d(r)[0])
    ds = get("POINT", "DATA_START")
    if ds is not None:
        data_block = int(np.atleast_1d(ds)[0]) & 0xFFFF
    fr = get("POINT", "FRAMES")
    frames = last - first + 1
    if fr is not None and frames <= 0:
        frames = int(np.atleast_1d(fr)[0]) & 0xFFFF
    off = (data_block - 1) * 512
    words = npoints * 4 + nanalog
    if scale < 0:  # float data
        raw = np.frombuffer(d, dtype="<f4", count=frames * words, offset=off).reshape(frames, words)
        pts = raw[:, :npoints * 4].reshape(frames, npoints, 4)
        xyz = pts[:, :, :3].astype(np.float32)
        resid = pts[:, :, 3].astype(np.float32)
    else:  # scaled 16-bit integers; residual byte in the 4th word
        raw = np.frombuffer(d, dtype="<i2", count=frames * words, offset=off).reshape(frames, words)
        pts = raw[:, :npoints * 4].reshape(frames, npoints, 4)
        xyz = (pts[:, :, :3].astype(np.float32) * np.float32(scale))
        resid = pts[:, :, 3].astype(np.float32)
    valid = resid >= 0
    return dict(labels=labels, rate=rate, points=np.ascontiguousarray(xyz), valid=valid, residual=resid, first_frame=first,
                units=get("POINT", "UNITS", ""))


def write_c3d(path: str, labels: Sequence[str], points: np.ndarray, valid: Optional[np.ndarray] = None, rate: float = 120.0):
    """Minimal float-format C3D writer (tests and synthetic sequences): POINT:LABELS/RATE/USED/SCALE/DATA_START/FRAMES."""
    points = np.asarray(points, np.float32)
    T, P, _ = points.shape
    valid = np.ones((T, P), bool) if valid is None else np.asarray(valid, bool)
    width = max(max(len(s) for s in labels), 1)

    def group(gid, name):
        b = name.encode()
        return struct.pack("<bb", len(b), -gid) + b + struct.pack("<h", 3) + b"\x00"

    def param(gid, name, typ, dims, payload):
        b = name.encode()
        body = struct.pack("<bB", typ, len(dims)) + bytes(dims) + payload + b"\x00"
        return struct.pack("<bb", len(b), gid) + b + struct.pack("<h", 2 + len(body)) + body

    lab = b"".join(s.encode().ljust(width) for s in labels)
    recs = [group(1, "POINT"),
            param(1, "USED", 2, [], struct.pack("<h", P)),
            param(1, "FRAMES", 2, [], struct.pack("<h", T if T < 32768 else -1)),
            param(1, "SCALE", 4, [], struct.pack("<f", -1.0)),
            param(1, "RATE", 4, [], struct.pack("<f", rate)),
            param(1, "UNITS", -1, [1], b"m"),
            param(1, "LABELS", -1, [width, P], lab)]
    blob = b"".join(recs)
    nblocks = (4 + len(blob) + 2 + 511) // 512
    data_block = 2 + nblocks
    recs.insert(2, param(1, "DATA_START", 2, [], struct.pack("<H", data_block)))
    blob = b"".join(recs)
    nblocks2 = (4 + len(blob) + 2 + 511) // 512
    assert nblocks2 == nblocks
    # terminate the list: last record's "next" must be 0
    blob = blob + struct.pack("<bb", 0, 0)
    pblock = (struct.pack("<BBBB", 1, 0x50, nblocks, 84) + blob).ljust(nblocks * 512, b"\x00")
    hdr = struct.pack("<BBHHHHHfHHf", 2, 0x50, P, 0, 1, T if T < 65536 else 65535, 0, -1.0, data_block, 0, rate).ljust(512, b"\x00")
    data = np.zeros((T, P, 4), "<f4")
    data[:, :, :3] = points
    data[:, :, 3] = np.where(valid, 0.0, -1.0)
    with open(path, "wb") as f:
        f.write(hdr + pblock + data.tobytes())


def match_markers(point_labels: Sequence[str], task_names: Sequence[str]) -> List[int]:
    """node/node.cpp:583-594: first label that ENDS with the task name (labels carry a 'Skeleton:' style prefix)."""
    out = []
    for name in task_names:
        idx = next((i for i, s in enumerate(point_labels) if len(s) >= len(name) and s.endswith(name)), len(point_labels))
        if idx == len(point_labels):
            raise KeyError("mocap marker %s not found" % name)
        out.append(idx)
    return out


# ------------------------------------------------------------------------------------------------ sequence driver
class MocapMotionSolver:
    """R independent chains (restarts or sequences) over T frames, in lock step on one GPU."""

    WARMUP_ITERS = 32  # frame 0 is solved in passes ikIter = 0..31: the store block (ikIter > 30, node/node.cpp:1369) follows the solve

    def __init__(self, smpl, face_idx, vertex_weights, restarts: int, vposer=None, marker_thickness=0.015, chain_base: int = 0):
        """`chain_base`: global index of this solver's chain 0 when it holds one GPU's share of the chains (dist.shard_range):
        a chain's trajectory has the same bits whichever shard it runs in."""
        from .ik import IkSolver

        self.K = len(face_idx)
        self.R = restarts
        self.solver = IkSolver(smpl, restarts, self.K, vposer=vposer, frame_base=chain_base)
        self.vposer = vposer
        K = self.K
        self.solver.setTasks(face_idx=np.asarray(face_idx, np.int64), vertex_weights=np.asarray(vertex_weights, np.float32),
                             normal_task_weight=np.zeros(K), normal_offset=np.full(K, marker_thickness),  # :553-562
                             phi_limit=np.zeros(K))  # :567, :699

    def solve(self, markers: np.ndarray, valid: np.ndarray, beta: np.ndarray, theta0: np.ndarray, frame_interval: int = 1,
              max_frames: Optional[int] = None, host_loop: bool = False):
        """markers [R,T,K,3] (or [T,K,3] shared), valid [R,T,K]; returns theta per solved frame [R,Ts,theta_dim]
        and the list of solved frame indices. The frame loop runs on the device (smplpp_ik_solve_sequence);
        host_loop=True drives it frame by frame from here (same kernels, same results; kept as the cross-check)."""
        R, K = self.R, self.K
        markers = np.asarray(markers, np.float32)
        valid = np.asarray(valid, bool)
        shared = markers.ndim == 3  # one capture for every restart: its targets are repeated on the device, not here
        if shared and not host_loop:
            T = markers.shape[0]
            frames = list(range(0, T, frame_interval))
            if max_frames is not None:
                frames = frames[:max_frames]
            self.solver.setConfig(np.broadcast_to(np.asarray(beta, np.float32), (R, 10)).copy(), theta0)
            v = np.ascontiguousarray(valid[frames])  # [Ts,K]
            tp = np.where(v[..., None], markers[frames], 0.0).astype(np.float32)  # node.cpp:681-690
            th = self.solver.solveSequence(tp, v, warmup_iters=self.WARMUP_ITERS, iters_per_frame=1, enable_qp=True, min_valid=K // 2)
            return th.transpose(1, 0, 2), frames  # (a view: [R,Ts,theta_dim] without a second pass over the result)
        if shared:
            markers = np.broadcast_to(markers, (R,) + markers.shape)
            valid = np.broadcast_to(valid, (R,) + valid.shape)
        T = markers.shape[1]
        frames = list(range(0, T, frame_interval))
        if max_frames is not None:
            frames = frames[:max_frames]
        self.solver.setConfig(np.broadcast_to(np.asarray(beta, np.float32), (R, 10)).copy(), theta0)
        out = np.empty((R, len(frames), self.solver.theta_dim), np.float32)
        min_valid = K // 2  # integer division, node.cpp:785
        if not host_loop:
            v = np.ascontiguousarray(valid[:, frames].transpose(1, 0, 2))  # [Ts,R,K]
            tp = np.where(v[..., None], markers[:, frames].transpose(1, 0, 2, 3), 0.0).astype(np.float32)  # :681-690
            th = self.solver.solveSequence(tp, v, warmup_iters=self.WARMUP_ITERS, iters_per_frame=1, enable_qp=True,
                                           min_valid=min_valid)
            return np.ascontiguousarray(th.transpose(1, 0, 2)), frames
        for i, t in enumerate(frames):
            v = valid[:, t]
            tp = np.where(v[..., None], markers[:, t], 0.0).astype(np.float32)  # :681-690
            self.solver.setTasks(target_pos=np.ascontiguousarray(tp), pos_task_weight=v.astype(np.float64))
            self.solver.iterate(self.WARMUP_ITERS if i == 0 else 1, enable_qp=True, min_valid=min_valid)
            _, th = self.solver.getConfig()
            out[:, i] = th.reshape(R, -1)
        return out, frames

    def decode_theta(self, g_theta: np.ndarray) -> np.ndarray:
        """theta [.., 25, 3] from the stored configuration; with a VPoser the 44-d layout is spliced like
        node/node.cpp:1376-1391."""
        g = np.asarray(g_theta, np.float32)
        if self.vposer is None:
            return g.reshape(g.shape[:-1] + (25, 3))
        flat = g.reshape(-1, 44)
        body = self.vposer.forward(flat[:, 6:38]).reshape(-1, 63)
        th = np.concatenate([flat[:, :6], body, flat[:, 38:44]], axis=1)
        return th.reshape(g.shape[:-1] + (25, 3))


class MocapBodySolver:
    """solveMocapBody (node/node.cpp:652-656, 674-679, 693-696, 1343-1352, 1418-1431) for R independent restarts in lock
    step on one GPU: 51 iterations on ONE capture frame whose markers are all present; iterations 0-24 move theta only
    (phiLimit_ = 0), from iteration 25 on theta + phi (|phi| <= 0.04 m: the markers slide on the surface) + beta
    (|dbeta| <= 0.5 per iteration) by box QP; marker-thickness normal offset 15 mm, normal task off (:553-562). The result
    is what the motion stage starts from: beta and, per marker, the face and barycentric weights it ended on
    (/tmp/MocapBody.yaml, :1418-1431)."""

    ITERS = 51       # ikIter 0..50: the loop breaks at the END of pass 50 (node.cpp:1349)
    BETA_FROM = 25   # optimizeBeta = ikIter >= 25 (:655); phiLimit_ = ikIter < 25 ? 0 : 0.04 (:695)
    PHI_LIMIT = 0.04

    def __init__(self, smpl, names: Sequence[str], restarts: int = 1, vposer=None, marker_thickness=0.015, chain_base: int = 0):
        from .ik import IkSolver

        self.names = sorted(names)  # std::map<std::string, IkTask> order (node.cpp:47, 798)
        self.faces = np.array([BASELINE41[n] for n in self.names], np.int64)
        self.K = len(self.names)
        self.R = restarts
        self.solver = IkSolver(smpl, restarts, self.K, vposer=vposer, frame_base=chain_base)
        K = self.K
        self.solver.setTasks(face_idx=self.faces, vertex_weights=np.full((K, 3), 1 / 3, np.float32),
                             normal_task_weight=np.zeros(K), normal_offset=np.full(K, marker_thickness),
                             phi_limit=np.full(K, self.PHI_LIMIT))

    def solve(self, markers: np.ndarray, theta0: np.ndarray, beta0: Optional[np.ndarray] = None, iters: Optional[int] = None):
        """markers [K,3] (or [R,K,3]) in the task order `self.names`, every one present (a missing marker is an error in
        this stage, node.cpp:674-679); theta0 [R,25,3] (or [R,44] with a VPoser). Returns dict(beta [R,10], theta,
        face_idx [R,K], vertex_weights [R,K,3], e_sqnorm [R])."""
        R, K = self.R, self.K
        m = np.asarray(markers, np.float32)
        if m.ndim == 2:
            m = np.broadcast_to(m, (R,) + m.shape)
        if m.shape != (R, K, 3) or not np.isfinite(m).all():
            raise ValueError("All mocap markers must be found to solve mocap body")  # node.cpp:677
        # every call starts from the marker table's faces at their centroids, like a fresh run of the node
        self.solver.setTasks(face_idx=self.faces, vertex_weights=np.full((K, 3), 1 / 3, np.float32), target_pos=np.ascontiguousarray(m),
                             pos_task_weight=np.ones((R, K)))
        b0 = np.zeros((R, 10), np.float32) if beta0 is None else np.broadcast_to(np.asarray(beta0, np.float32), (R, 10)).copy()
        self.solver.setConfig(b0, theta0)
        e2 = self.solver.iterate(self.ITERS if iters is None else iters, enable_qp=True, optimize_beta_from=self.BETA_FROM)
        beta, theta = self.solver.getConfig()
        t = self.solver.getTasks()
        return dict(beta=beta, theta=theta, face_idx=t["face_idx"], vertex_weights=t["vertex_weights"], e_sqnorm=e2)

    def write_yaml(self, path: str, result: dict, restart: Optional[int] = None) -> int:
        """MocapBody.yaml (node.cpp:1418-1431) of one restart (default: the one with the smallest final residual)."""
        r = int(np.argmin(result["e_sqnorm"])) if restart is None else int(restart)
        write_mocap_body_yaml(path, result["beta"][r], self.names, result["face_idx"][r], result["vertex_weights"][r])
        return r


# ------------------------------------------------------------------------------------------------ result files
def write_motion_text(path: str, theta: np.ndarray):
    """scripts/convertRosbagToText.py:18-19: one line per frame, the 75 numbers of theta (25x3 row-major) as float64."""
    th = np.asarray(theta, np.float64).reshape(len(theta), -1)
    with open(path, "w") as f:
        for row in th:
            f.write(" ".join(repr(float(x)) for x in row) + "\n")


def write_mocap_body_yaml(path: str, beta, names: Sequence[str], face_idx, vertex_weights):
    """node/node.cpp:1426-1441 (/tmp/MocapBody.yaml), Eigen FullPrecision formatting."""
    def vec(v):
        return "[" + ", ".join(repr(float(np.float32(x))) if False else ("%.9g" % float(x)) for x in v) + "]"

    with open(path, "w") as f:
        f.write("beta: %s\n" % vec(beta))
        f.write("ikTaskList:\n")
        for n, fi, w in zip(names, face_idx, vertex_weights):
            f.write("  - name: %s\n    faceIdx: %d\n    vertexWeights: %s\n" % (n, int(fi), vec(w)))


def read_mocap_body_yaml(path: str):
    """node/node.cpp:509-534: beta + per-task {name, faceIdx, vertexWeights}."""
    import yaml

    y = yaml.safe_load(open(path))
    beta = np.asarray(y["beta"], np.float32)
    if beta.shape != (10,):
        raise ValueError("Size of beta must be 10 but %d" % beta.size)  # node.cpp:513-517
    names = [t["name"] for t in y["ikTaskList"]]
    faces = np.array([t["faceIdx"] for t in y["ikTaskList"]], np.int64)
    weights = np.array([t["vertexWeights"] for t in y["ikTaskList"]], np.float32)
    return beta, names, faces, weights
