"""One IK step of the reference's loop body in the 44-d VPoser layout, on the CPU — test infrastructure.

node/node.cpp:761-772 splices theta25 = [pos | root | vposer(z) (21 joints) | aa22 | aa23] from g_theta44 = [pos3 | root3 | z32 |
aa22 | aa23]; autograd then differentiates the residual w.r.t. g_theta44 (:823-869), i.e. the direct Jacobian pulled back through
d vposer / d z; :895-904 adds the prior (w_i to A_ii, w_i * theta_i to b_i); :906-943 solves (box QP on phi and d beta, or LLT);
:945-1001 updates and re-projects.  Built from the pinned pieces: oracle.ik_eval (C restatement, pinned by the reference-compiled
goldens), oracle/vposer_torch.py (op-for-op restatement of src/VPoser.cpp, value and Jacobian by torch autograd),
cpu.normal_equations / llt_solve / box_qp.  The box QP is the oracle's own active-set solver (QLD is absent from /root/reference:
parity unpinned for its iterates; the optimum of the strictly convex QP is unique)."""
import numpy as np

from oracle import cpu

LATENT = slice(6, 38)                              # the 32 latent coordinates of the 44-d layout
PASS = np.r_[0:6, 38:44]                           # the 12 entries that pass through to theta25: metres and radians


def splice(g44, vout):
    """theta25 [25,3] of node.cpp:763-771 from the 44-vector and the decoder's 21 x 3 axis-angles."""
    th25 = np.zeros((25, 3), np.float32)
    th25[0], th25[1] = g44[:3], g44[3:6]
    th25[2:23] = np.asarray(vout, np.float32).reshape(21, 3)
    th25[23], th25[24] = g44[38:41], g44[41:44]
    return th25


def latent_step(oracle, ref_decoder, beta, g44, tasks, enable_qp=True, optimize_beta=False, phi_live=None, project=True):
    """One pass of node.cpp:750-1001 in the latent layout from (beta [10], g44 [44], tasks: faces + weights as the engine holds
    them).  `phi_live` [K]: this pass's phiLimit_ (0 pins the surface coordinates; default: tasks.phi_limit).  Returns a dict:
    g44 / beta after the update, x (the step), e_sqnorm, and — with `project` — the re-projected faces, weights and closest points."""
    K = tasks.K
    ts = tasks.copy()
    if phi_live is not None:
        ts.phi_limit[:] = np.asarray(phi_live, np.float64)
    g44 = np.asarray(g44, np.float32).reshape(44)
    beta = np.asarray(beta, np.float32).reshape(10)
    vout, vjac = ref_decoder.forward_with_jacobian(g44[None, LATENT])
    th25 = splice(g44, vout[0])
    r = oracle.ik_eval(beta, th25, ts, optimize_beta=optimize_beta, want_verts=project)
    J75 = r["J"]
    Jl = np.concatenate([J75[:, :6], J75[:, 6:69] @ vjac[0].reshape(63, 32).astype(np.float64), J75[:, 69:75], J75[:, 75:]], axis=1)
    bd = 10 if optimize_beta else 0
    A, b = cpu.normal_equations(r["e"], Jl, 44, 2 * K, bd, vposer_theta=g44)
    D = 44 + 2 * K + bd
    if enable_qp:  # node.cpp:911-929: theta free, |phi| <= phiLimit_, |d beta| <= 0.5
        lo, hi = np.full(D, -np.inf), np.full(D, np.inf)
        for k in range(K):
            lo[44 + 2 * k: 46 + 2 * k] = -ts.phi_limit[k]
            hi[44 + 2 * k: 46 + 2 * k] = ts.phi_limit[k]
        lo[44 + 2 * K:] = -0.5
        hi[44 + 2 * K:] = 0.5
        x = cpu.box_qp(A, b, lo, hi)
    else:
        x = cpu.llt_solve(A, b)
    out = dict(x=x, e_sqnorm=float(r["e"] @ r["e"]), theta25_before=th25)
    out["g44"] = (g44 + x[:44].astype(np.float32)).astype(np.float32)  # :947 (fp32 update)
    out["beta"] = (beta + x[44 + 2 * K:].astype(np.float32)).astype(np.float32) if optimize_beta else beta.copy()
    if project:  # :949-1001: p_k = actualPos_k + tangents_k . x_phi_k on the PRE-update mesh, closest face, area-ratio weights
        tang = ts.tangents.reshape(K, 3, 2)
        xphi = x[44: 44 + 2 * K].reshape(K, 2).astype(np.float32)
        pts = r["actual_pos"] + np.einsum("kxc,kc->kx", tang, xphi)
        face, closest, _ = oracle.closest_points(r["verts"], pts.astype(np.float32))
        out.update(face_idx=face, closest=closest, verts=r["verts"], query=pts)
    return out


def decoded_angles(ref_decoder, g44):
    """The 63 body angles the decoder emits for the latent part of a 44-vector (radians)."""
    import torch

    with torch.no_grad():
        z = torch.from_numpy(np.ascontiguousarray(np.asarray(g44, np.float32).reshape(1, 44)[:, LATENT]))
        return ref_decoder.forward(z).numpy().reshape(63)


def compare_states(ref_decoder, g_engine, g_oracle):
    """Distances between two 44-d configurations in the units the north star speaks of: (metres / radians on the 12 pass-through
    entries, radians on the 63 decoded body angles, latent units on the 32 latent coordinates)."""
    g_engine, g_oracle = np.asarray(g_engine, np.float32).reshape(44), np.asarray(g_oracle, np.float32).reshape(44)
    d_pass = float(np.abs(g_engine[PASS] - g_oracle[PASS]).max())
    d_ang = float(np.abs(decoded_angles(ref_decoder, g_engine) - decoded_angles(ref_decoder, g_oracle)).max())
    d_lat = float(np.abs(g_engine[LATENT] - g_oracle[LATENT]).max())
    return d_pass, d_ang, d_lat
