"""Generate the face topology of the synthetic SMPL-shaped model (run once; output is committed).

The real SMPL topology is license-gated and absent.  Any closed genus-0 triangulation of V = 6890 vertices has
E = 3V - 6 = 20664 edges and F = 2V - 4 = 13776 faces — exactly SMPL's counts (def.h: FACE_INDEX_NUM = 13776) — so
the convex hull of 6890 points in general position on a sphere gives a manifold of the right size.  Faces are
oriented outward and stored 0-based as uint16 [13776, 3]; smplpp_amd.model_io adds 1 (the reference model files
are 1-based, scripts/preprocess.py:91).
"""
import numpy as np
from scipy.spatial import ConvexHull

V = 6890


def fibonacci_sphere(n):
    i = np.arange(n, dtype=np.float64) + 0.5
    phi = np.arccos(1.0 - 2.0 * i / n)
    theta = np.pi * (1.0 + 5.0 ** 0.5) * i
    return np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)


def main():
    pts = fibonacci_sphere(V)
    hull = ConvexHull(pts)
    f = hull.simplices.astype(np.int64)
    assert f.shape == (2 * V - 4, 3), f.shape
    a, b, c = pts[f[:, 0]], pts[f[:, 1]], pts[f[:, 2]]
    flip = np.einsum("ij,ij->i", np.cross(b - a, c - a), a + b + c) < 0
    f[flip] = f[flip][:, [0, 2, 1]]
    # canonical order: rotate each face so its smallest index is first, then sort rows
    k = np.argmin(f, axis=1)
    f = np.stack([f[np.arange(len(f)), (k + j) % 3] for j in range(3)], axis=1)
    f = f[np.lexsort((f[:, 2], f[:, 1], f[:, 0]))]
    assert len(np.unique(f)) == V
    edges = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    ue, cnt = np.unique(edges, axis=0, return_counts=True)
    assert len(ue) == 3 * V - 6 and (cnt == 2).all()
    np.save("smplpp_amd/data/synthetic_faces.npy", f.astype(np.uint16))
    print("faces", f.shape, "max valence", np.bincount(f.ravel()).max())


if __name__ == "__main__":
    main()
