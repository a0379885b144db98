#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: SMPL FK evals/s (+ IK iterations/s), batch 1024 frames.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (SMPL::launch: pose/chain kernel + fused blend-shape GEMM + skinning kernel)
over one batch of 1024 synthetic frames per GPU, inputs and outputs resident in HBM (BASELINE.json configs[1]).
Frames are independent, so N GPUs run N independent shards (weak scaling) with no data-path collective; timing is
barrier + synchronize on both sides, max over ranks.  Rank 0 prints ONE JSON line.

Also reported in the same line:
  roofline      dominant kernel (skin_kernel): algorithmic FLOPs / bytes per launch divided by
                the kernel's mean duration measured with HIP events on its launch stream over the timed region;
  ik            IK iterations/s on BASELINE.json configs[2] (6 targets, 50 iterations, 256 frames per GPU);
  cpu_baseline  the reference's own compiled FK stages (oracle/_ref, libtorch-CPU) — or the C port when that
                library is absent — timed on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

V = 6890
# model constants once per batch + per-frame beta/theta in and vertices out
ALG_BYTES_CONST = 19_347_120
ALG_BYTES_PER_FRAME = 83_020
# fp32 FLOPs of the dense contraction the MFMA pipe executes per frame: 2 * 20670 * (207 posedirs + 10 shapedirs)
ALG_MFMA_FLOPS_PER_FRAME = 2 * 20670 * 217
ALG_FLOPS_PER_FRAME = 15.5e6  # whole FK
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_32x32x2_f32)
PEAK_MFMA_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (no sparsity)
# The default fused kernel (skin_b.hip) evaluates every fp32 product as 6 bf16 piece products, K padded 220 -> 224
BF16X3_ISSUE_FACTOR = 6.0 * 224.0 / 217.0


def usable_cpus():
    """Host cores this process may actually use: min(affinity mask, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(model, frames, budget_s=12.0):
    """Reference libtorch-CPU FK (oracle/_ref) on this box's cores; falls back to the C port.  Checker code is only
    ever used here as the thing timed for the reported baseline — never on the GPU product path."""
    from smplpp_amd import model_io

    beta, theta = model_io.synthetic_inputs(frames, seed=1)
    kind, cores, run = None, 1, None
    try:
        from oracle import ref

        if ref.available():
            R = ref.RefModel(model)
            ref.lib().ref_set_num_threads(usable_cpus())  # libtorch defaults to every core of the HOST, not our share
            cores = ref.lib().ref_get_num_threads()
            run = lambda: R.fk_launch_only(beta, theta)
            kind = "reference"
    except Exception as e:  # libtorch unusable here: fall back to the port
        sys.stderr.write("cpu_baseline: reference build unavailable (%s); using the C port\n" % e)
    if run is None:
        from oracle import cpu

        O = cpu.OracleModel(model)
        cores = min(usable_cpus(), cpu.lib().oracle_max_threads())
        run = lambda: O.fk(beta, theta, want=("verts",), threads=cores)
        kind = "port"
    run()  # warm
    t0 = time.perf_counter()
    batches = 0
    while True:
        run()
        batches += 1
        el = time.perf_counter() - t0
        if el >= budget_s or batches >= 40:
            break
    return {
        "value": frames * batches / el, "unit": "FK evals/s", "cores": int(cores), "kind": kind,
        "sample": "%d batches of %d frames (%.1f s) of the same synthetic workload, SMPL::launch only" % (batches, frames, el),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=1024, help="frames per GPU per step (BASELINE: 1024)")
    ap.add_argument("--ik-frames", type=int, default=256)
    ap.add_argument("--ik-iters", type=int, default=50)
    ap.add_argument("--no-ik", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[3] (mocap excerpt) and configs[4] (VPoser-latent IK) legs")
    ap.add_argument("--mocap-restarts", type=int, default=8, help="restarts per GPU of the capture excerpt (BASELINE: 64 over 8 GPUs)")
    ap.add_argument("--vposer-frames", type=int, default=128, help="frames per GPU of the VPoser-latent IK leg (BASELINE: 512 over 4 GPUs)")
    ap.add_argument("--gather", action="store_true", help="also time one final RCCL gather of all vertices to rank 0")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--all-ranks-on-device0", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses GPU 0 (use with --backend gloo)")
    args = ap.parse_args()

    import torch

    from smplpp_amd import dist as D
    from smplpp_amd import model_io
    from smplpp_amd.smpl import SMPL

    rank, world, local = D.env_rank_world()
    if world != args.gpus and world != 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    if args.all_ranks_on_device0:
        local = 0
    torch.cuda.set_device(local)
    D.init_process_group((args.backend or "nccl") if world > 1 else None)

    model = model_io.synthetic_model()
    smpl = SMPL()
    smpl.setDevice("cuda:%d" % local)
    smpl.init(model)

    n = args.frames
    beta_h, theta_h = model_io.synthetic_inputs(n, seed=1 + rank)
    beta = torch.from_numpy(beta_h).cuda()
    theta = torch.from_numpy(theta_h).cuda()
    out = {"verts": torch.empty((n, V, 3), dtype=torch.float32, device="cuda")}

    def step():
        smpl.launch(beta, theta, want=("verts",), out=out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    D.barrier()
    smpl.profileEnable(True)
    smpl.profileRead()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    D.barrier()
    elapsed = time.perf_counter() - t0
    launches, skin_ms = smpl.profileRead()
    smpl.profileEnable(False)
    elapsed = D.max_over_ranks(elapsed)
    skin_ms = D.max_over_ranks(skin_ms)

    # ---- IK leg (BASELINE.json configs[2]): 6 targets, 50 iterations, 256 frames per GPU
    ik = None
    if not args.no_ik:
        from smplpp_amd.ik import IkSolver, reference_task_faces

        K = 6
        _, faces = reference_task_faces(K)
        rng = np.random.default_rng(100 + rank)
        hid = np.zeros((args.ik_frames, 25, 3), np.float32)
        hid[:, 1:] = rng.normal(0, 0.2, (args.ik_frames, 24, 3))
        hv = smpl.launch(np.zeros((args.ik_frames, 10), np.float32), hid, want=("verts",))["verts"]
        f0 = model["face_indices"][faces] - 1
        tp = hv[:, f0].mean(axis=2)  # reachable targets: task points of a hidden pose ...
        tn = smpl.calcVertexNormalBatch(f0.reshape(-1)).reshape(args.ik_frames, K, 3, 3).mean(axis=2)  # ... and its normals there
        tn = -(tn / np.linalg.norm(tn, axis=-1, keepdims=True)).astype(np.float32)  # e_n = w (n.n_target + 1): zero when opposed (node.cpp:813)
        theta0 = np.zeros((args.ik_frames, 25, 3), np.float32)
        theta0[:, 1:] = rng.normal(0, 0.05, (args.ik_frames, 24, 3))
        solver = IkSolver(smpl, args.ik_frames, K)
        solver.setTasks(face_idx=faces, target_pos=tp, target_normal=tn, phi_limit=np.zeros(K), normal_task_weight=np.ones(K))
        reps = 3
        ik_t = 0.0
        for rep in range(reps + 1):
            solver.setTasks(face_idx=faces, vertex_weights=np.full((K, 3), 1 / 3, np.float32))
            solver.setConfig(np.zeros((args.ik_frames, 10), np.float32), theta0)
            torch.cuda.synchronize()
            D.barrier()
            t1 = time.perf_counter()
            e2 = solver.iterate(args.ik_iters)
            torch.cuda.synchronize()
            D.barrier()
            if rep > 0:  # rep 0 is warm-up
                ik_t += time.perf_counter() - t1
        ik_t = D.max_over_ranks(ik_t / reps)
        ik = {
            "value": world * args.ik_frames * args.ik_iters / ik_t, "unit": "IK iterations/s", "frames_per_gpu": args.ik_frames,
            "iters": args.ik_iters, "tasks": K, "ms_per_iter_batch": ik_t / args.ik_iters * 1e3,
            "final_max_e_sqnorm": float(np.max(e2)), "final_median_e_sqnorm": float(np.median(e2)),
            "frames_below_1e-3": int((e2 < 1e-3).sum()),  # the normal terms make the problem non-convex: a start can end in a local minimum
            "workload": "configs[2]: 6-target IK (position + normal term per target), 50 iterations, direct theta (D = 87)",
        }

    # ---- configs[3]: the capture excerpt (tests/golden/sample_walk_excerpt.npz: 32 frames x 41 Baseline markers of
    # data/sample_walk.c3d), R warm-started chains per GPU, marker-thickness normal offsets, QP on, 31 warm-up iterations
    # on frame 0 then one iteration per frame (node.cpp:1369-1407); configs[4]: VPoser-latent IK (44-d layout, synthetic
    # decoder: the real weights cannot travel), 6 targets, 50 iterations
    mocap_leg = vposer_leg = None
    if not args.no_ik and not args.no_extra:
        from smplpp_amd import mocap
        from smplpp_amd.ik import VPoserDecoder

        g = np.load(os.path.join(ROOT, "tests", "golden", "sample_walk_excerpt.npz"))
        names = list(g["task_names"])
        mfaces = np.array([mocap.BASELINE41[nm] for nm in names], np.int64)
        Km = len(names)
        pts = g["points"] - g["points"][0][g["valid"][0]].mean(axis=0) + np.array([0, -0.3, 0], np.float32)
        R = args.mocap_restarts
        rng = np.random.default_rng(200 + rank)
        th0 = np.zeros((R, 25, 3), np.float32)
        th0[:, 1:] = rng.normal(0, 0.03, (R, 24, 3))  # the restarts differ in their initial pose
        ms = mocap.MocapMotionSolver(smpl, mfaces, np.full((Km, 3), 1 / 3, np.float32), restarts=R)
        ms.solve(pts, g["valid"], np.zeros(10, np.float32), th0, max_frames=2)  # warm-up of the code path
        torch.cuda.synchronize()
        D.barrier()
        t1 = time.perf_counter()
        thm, fr = ms.solve(pts, g["valid"], np.zeros(10, np.float32), th0)
        torch.cuda.synchronize()
        D.barrier()
        mt = D.max_over_ranks(time.perf_counter() - t1)
        iters = mocap.MocapMotionSolver.WARMUP_ITERS + len(fr) - 1
        mocap_leg = {
            "value": world * R * len(fr) / mt, "unit": "solved capture frames/s", "ik_iterations_per_s": world * R * iters / mt,
            "restarts_per_gpu": R, "frames": len(fr), "markers": Km, "finite": bool(np.isfinite(thm).all()),
            "workload": "configs[3]: sample_walk.c3d excerpt (32 frames x 41 markers), warm-started chains, box QP, direct theta (D = 157); "
                        "frame loop on the device (smplpp_ik_solve_sequence: targets uploaded once, no host round trip per frame)",
        }

        Kv = 6
        nv = args.vposer_frames
        vp = VPoserDecoder(VPoserDecoder.synthetic_params(seed=3), device=local)
        _, vfaces = reference_task_faces(Kv)
        hidv = np.zeros((nv, 25, 3), np.float32)
        hidv[:, 1:22] = rng.normal(0, 0.15, (nv, 21, 3))
        hvv = smpl.launch(np.zeros((nv, 10), np.float32), hidv, want=("verts",))["verts"]
        tpv = hvv[:, model["face_indices"][vfaces] - 1].mean(axis=2)
        # the reference's own capture setting (node.cpp:316-322 forces VPoser + QP on): 44-d layout, D = 44 + 2 * 41
        msv = mocap.MocapMotionSolver(smpl, mfaces, np.full((Km, 3), 1 / 3, np.float32), restarts=R, vposer=vp)
        gv0 = np.zeros((R, 44), np.float32)
        gv0[:, 6:38] = rng.normal(0, 0.05, (R, 32))
        msv.solve(pts, g["valid"], np.zeros(10, np.float32), gv0, max_frames=2)
        torch.cuda.synchronize()
        D.barrier()
        t1 = time.perf_counter()
        thv, frv = msv.solve(pts, g["valid"], np.zeros(10, np.float32), gv0)
        torch.cuda.synchronize()
        D.barrier()
        mtv = D.max_over_ranks(time.perf_counter() - t1)
        mocap_leg["vposer_latent"] = {
            "value": world * R * len(frv) / mtv, "unit": "solved capture frames/s", "finite": bool(np.isfinite(thv).all()),
            "workload": "same excerpt with the 44-d VPoser layout the reference forces on capture solves (D = 126), synthetic decoder weights",
        }
        vs = IkSolver(smpl, nv, Kv, vposer=vp)
        vs.setTasks(face_idx=vfaces, target_pos=tpv, phi_limit=np.zeros(Kv), normal_task_weight=np.zeros(Kv))
        g0 = np.zeros((nv, vs.theta_dim), np.float32)
        vt = 0.0
        for rep in range(3):
            vs.setTasks(face_idx=vfaces, vertex_weights=np.full((Kv, 3), 1 / 3, np.float32))
            vs.setConfig(np.zeros((nv, 10), np.float32), g0)
            torch.cuda.synchronize()
            D.barrier()
            t1 = time.perf_counter()
            ev = vs.iterate(args.ik_iters)
            torch.cuda.synchronize()
            D.barrier()
            if rep > 0:
                vt += time.perf_counter() - t1
        vt = D.max_over_ranks(vt / 2)
        vposer_leg = {
            "value": world * nv * args.ik_iters / vt, "unit": "IK iterations/s", "frames_per_gpu": nv, "iters": args.ik_iters, "tasks": Kv,
            "ms_per_iter_batch": vt / args.ik_iters * 1e3, "final_max_e_sqnorm": float(np.max(ev)),
            "workload": "configs[4]: VPoser-latent IK (32-d latent + decoder in the loop, 44-d layout, D = 56), synthetic decoder weights",
        }

    gather_ms = None
    if args.gather and world > 1:
        torch.cuda.synchronize()
        D.barrier()
        t1 = time.perf_counter()
        full = D.gather_rows(out["verts"], n * world)
        torch.cuda.synchronize()
        gather_ms = D.max_over_ranks((time.perf_counter() - t1) * 1e3)
        del full

    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    value = world * n * args.steps / elapsed
    mfma_tflops = ALG_MFMA_FLOPS_PER_FRAME * n / (skin_ms * 1e-3) / 1e12 if skin_ms > 0 else 0.0
    hbm_gbs = (ALG_BYTES_CONST + ALG_BYTES_PER_FRAME * n) / (skin_ms * 1e-3) / 1e9 if skin_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):  # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/README.md)
        try:
            traffic = json.load(open(tpath)).get("skin_kernel_hbm_bytes_per_launch_n%d" % n)
        except Exception:
            traffic = None
    form = (os.environ.get("SMPLPP_SKIN") or "b")[0]
    if form == "b":
        issued = mfma_tflops * BF16X3_ISSUE_FACTOR
        roofline = {
            "kernel": "skin_kernel_b<4,false> (fused blend-shape GEMM + linear blend skinning; fp32 operands as exact bf16x3 "
                      "pieces on the bf16 matrix pipe, fp32 accumulate)",
            "bound": "mfma", "achieved": mfma_tflops, "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
            "frac": mfma_tflops / PEAK_MFMA_F32_TFLOPS, "traffic": traffic,
            "kernel_ms": skin_ms, "launches_timed": launches,
            "issued": {"dtype": "bf16", "achieved": issued, "peak": PEAK_MFMA_BF16_TFLOPS, "unit": "TFLOP/s",
                       "frac": issued / PEAK_MFMA_BF16_TFLOPS,
                       "note": "6 bf16 MFMA products per fp32 product (a1b1 a1b2 a2b1 a1b3 a2b2 a3b1), K 220 padded to 224"},
            "hbm": {"achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_gbs / PEAK_HBM_GBS,
                    "algorithmic_bytes_per_launch": ALG_BYTES_CONST + ALG_BYTES_PER_FRAME * n},
            "note": "achieved = ALGORITHMIC fp32 FLOPs of the blend-shape contraction (2*20670*217 per frame) / kernel time, "
                    "priced against the fp32 matrix peak (the arithmetic type of the path: results carry fp32 accuracy — max error vs "
                    "the fp64 oracle 7e-7 m, same as the fp32-MFMA form); 'issued' prices the bf16 instructions actually "
                    "executed against the dense bf16 peak. Dense bf16 MFMA holds ~1.6 GHz on this chip (tools/micro/mfma_lds.hip), "
                    "so the issued-rate ceiling is ~2/3 of the datasheet figure. Batch 1024 has 152 FLOP/B: the matrix pipe, "
                    "LDS and issue slots bind long before HBM",
        }
    else:
        roofline = {
            "kernel": "skin_kernel_p<4,false> (fused blend-shape GEMM + linear blend skinning, fp32 MFMA, persistent)"
                      if form == "p" else "fp32-MFMA fused kernel, form %s" % form,
            "bound": "mfma", "achieved": mfma_tflops, "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
            "frac": mfma_tflops / PEAK_MFMA_F32_TFLOPS, "traffic": traffic,
            "kernel_ms": skin_ms, "launches_timed": launches,
            "hbm": {"achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_gbs / PEAK_HBM_GBS,
                    "algorithmic_bytes_per_launch": ALG_BYTES_CONST + ALG_BYTES_PER_FRAME * n},
            "note": "batch 1024 in exact fp32 has 152 FLOP/B: the fp32 MFMA pipe binds before HBM (ridge ~20 FLOP/B); "
                    "peak is the 157.3 TF datasheet figure at 2.4 GHz — under sustained fp32 MFMA the chip holds ~1.7 GHz "
                    "(profiles/README.md), i.e. a practical ceiling near 110 TF",
        }
    line = {
        "metric": "SMPL FK evals/s + IK iters/s, batch 1024 frames, 1/2/4/8 MI355X",
        "value": value,
        "unit": "FK evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if form != "b" else "f32 (exact bf16x3 operand pieces, fp32 accumulate)",
        "data": "synthetic",
        "config": {
            "workload": "configs[1]: batch-%d random beta/theta FK+LBS per GPU, synthetic SMPL-shaped model "
                        "(6890 verts, 24 joints, 207 pose / 10 shape PCs), HBM-resident in/out" % n,
            "frames_per_gpu": n, "parallelism": "frames sharded x%d, no data-path collective" % world,
        },
        "roofline": roofline,
    }
    if ik is not None:
        line["ik"] = ik
    if mocap_leg is not None:
        line["mocap"] = mocap_leg
    if vposer_leg is not None:
        line["vposer_ik"] = vposer_leg
    if gather_ms is not None:
        line["final_gather_ms"] = gather_ms
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(model, n)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
