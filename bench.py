#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: SMPL FK evals/s (+ IK iterations/s), batch 1024 frames.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (SMPL::launch: pose/chain kernel + fused blend-shape GEMM + skinning kernel)
over one batch of 1024 synthetic frames per GPU, inputs and outputs resident in HBM (BASELINE.json configs[1]).
The contract's region (W warm-up steps, then exactly K timed ones between barrier + synchronize) is timed behind the IK legs,
on a chip that has been busy; `cold_start` is the same region timed first, right behind model creation (clocks still ramping).
Frames are independent, so N GPUs run N independent shards (weak scaling) with no data-path collective; timing is
barrier + synchronize on both sides, max over ranks.  Rank 0 prints ONE JSON line.

Also reported in the same line:
  roofline      dominant kernel (skin_kernel): algorithmic FLOPs / bytes per launch divided by
                the kernel's mean duration measured with HIP events on its launch stream over the timed region;
  ik            IK iterations/s on BASELINE.json configs[2] (6 targets, 50 iterations, 256 frames per GPU), with its own
                cpu_baseline (the reference's per-row autograd Jacobian + fp64 solve on the host cores);
  mocap         configs[3]: sample_walk.c3d, every frame, 64 restarts in all sharded over the GPUs, both layouts;
  vposer_ik     configs[4]: VPoser-latent IK, 512 frames in all sharded over the GPUs;
  cpu_baseline  the reference's own compiled FK stages (oracle/_ref, libtorch-CPU) — or the C port when that
                library is absent — timed on this box's host cores on a bounded sample (rank 0, N = 1 only);
  final_gather_ms  (N > 1) the gather of the results to rank 0 (grouped point-to-point sends over RCCL), the path's only exchange;
  sustained     the same FK step over a 2000-launch run (steady clocks);
  within_tolerance_form  the same step on the fp16x2 form (SMPLPP_SKIN=h: 22-bit operands, 3e-7 m — narrower than the reference's
                fp32, so never the headline; the form the IK loops' internal forward passes run);
  pipelined     the sustained run with its steps alternating between two model handles on two streams (a double-buffered caller).

The headline (`value`, `dtype`, `roofline`) is the library's default FK form: skin_kernel_e, every fp32 operand carried exactly
(three bf16 pieces, six MFMA products per fp32 product, fp32 accumulate, fp32 VALU skinning) — the reference's arithmetic
(/root/reference/src/BlendShape.cpp:762-765, src/LinearBlendSkinning.cpp:463-467).  `roofline.frac` is SURVEY 8(d)'s HBM fraction:
algorithmic bytes per launch / the fused kernel's measured duration / 8 TB/s; the issued-MFMA occupancy is a sub-field.
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
PEAK_MFMA_16BIT_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / f16 MFMA (no sparsity)
MFMA_32X32X16_FLOPS = 2 * 32 * 32 * 16


def issued_mfma_flops(form, n):
    """FLOPs of the matrix instructions the fused kernel actually ISSUES per launch.
    h (skin_h.hip): per 64 x 64 item and wavefront 126 GEMM MFMAs (14 k-steps x 3 coordinates x 3 piece products) + 60
    skinning MFMAs (12 entries x 5: K = 24 joints is 1.5 k-steps, the half k-step carries two of its products in one MFMA); b (skin_b.hip): 252 (14 x 3 x 6)."""
    items = ((n + 63) // 64) * ((V + 63) // 64)
    per_wave = {"h": 126 + 60, "b": 252, "e": 252}[form]
    return items * 4 * per_wave * MFMA_32X32X16_FLOPS


KERNEL_NAMES = {
    "e": "skin_kernel_e<4,false> (fused blend-shape GEMM + linear blend skinning; fp32 operands carried exactly as bf16x3 pieces, 6 MFMA "
         "products per fp32 product, fp32 accumulate; fp32 VALU skinning in the MFMA shadows; A fragments in registers, the frame "
         "tile's transforms resident in LDS, basis through a 4-image LDS-DMA ring)",
    "h": "skin_kernel_h<false> (fused blend-shape GEMM + linear blend skinning; fp32 operands as fp16x2 pieces, 3 MFMA products "
         "per fp32 product; skinning as MFMA products too; fp32 accumulate)",
    "b": "skin_kernel_b<4,false> (bf16x3 operand pieces, 6 MFMA products per fp32 product; VALU skinning in MFMA shadows)",
    "p": "skin_kernel_p<4,false> (exact fp32 MFMA, persistent)", "v": "skin_kernel<2,4> (exact fp32 MFMA, first form)",
}
DTYPES = {"e": "f32 (exact operands: every fp32 value as three bf16 pieces = fp32's 24 bits on the bf16 matrix pipe, fp32 accumulate; fp32 VALU skinning)",
          "h": "f32 (fp16x2 operand pieces on the f16 matrix pipe, fp32 accumulate: 22-bit operands, error 3e-7 m)",
          "b": "f32 (bf16x3, exact operands: three bf16 pieces = fp32's 24 bits, fp32 accumulate)"}


def roofline_object(form, n, skin_ms, launches, traffic):
    """The roofline entry of the fused kernel of `form` at batch n from its measured mean duration (ms)."""
    t_k = skin_ms * 1e-3
    alg_bytes = ALG_BYTES_CONST + ALG_BYTES_PER_FRAME * n
    f32_equiv_tflops = ALG_MFMA_FLOPS_PER_FRAME * n / t_k / 1e12 if t_k > 0 else 0.0
    hbm_gbs = alg_bytes / t_k / 1e9 if t_k > 0 else 0.0
    hbm = {"achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_gbs / PEAK_HBM_GBS, "algorithmic_bytes_per_launch": alg_bytes}
    if form in ("e", "h", "b"):
        issued = issued_mfma_flops(form, n)
        issued_tflops = issued / t_k / 1e12 if t_k > 0 else 0.0
        t_mfma, t_hbm = issued / (PEAK_MFMA_16BIT_TFLOPS * 1e12), alg_bytes / (PEAK_HBM_GBS * 1e9)
        # achieved = algorithmic bytes per launch / the kernel's measured duration, against the 8 TB/s HBM peak.  (The
        # matrix instructions the kernel issues, priced at the dense 16-bit MFMA peak, would take longer than the bytes at the HBM
        # peak: `mfma_issue` says how busy that pipe is; it is occupancy, not useful work.)
        roofline = {"kernel": KERNEL_NAMES[form], "bound": "hbm", "achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": hbm_gbs / PEAK_HBM_GBS}
        roofline.update({
            "traffic": traffic, "kernel_ms": skin_ms, "launches_timed": launches, "algorithmic_bytes_per_launch": alg_bytes,
            "roof_us": {"mfma": t_mfma * 1e6, "hbm": t_hbm * 1e6},
            "mfma_issue": {"achieved": issued_tflops, "peak": PEAK_MFMA_16BIT_TFLOPS, "unit": "TFLOP/s", "frac": issued_tflops / PEAK_MFMA_16BIT_TFLOPS,
                           "issued_mfma_flops_per_launch": issued,
                           "note": "matrix FLOPs the kernel ISSUES (v_mfma_f32_32x32x16 count x 32768: the operand split, the K padding and, "
                                   "for h, the dense K = 24 skinning are all in it) / kernel time against the dense 16-bit MFMA peak (2.5 PF, "
                                   "no sparsity): pipe occupancy, not useful work"},
            "useful": {"achieved": ALG_FLOPS_PER_FRAME * n / t_k / 1e12 if t_k > 0 else 0.0, "unit": "TFLOP/s",
                       "note": "algorithmic fp32 FLOPs of the whole FK (15.5 MFLOP per frame, SURVEY.md 8d) / kernel time"},
            "note": "frac = SURVEY 8(d): algorithmic bytes (19,347,120 + 83,020 N) / kernel time / 8 TB/s.  The kernel is bound by instruction "
                    "issue and the matrix pipe, not by bytes: `traffic` (PMC) is within 1.37x of the algorithmic bytes, and the chip holds "
                    "~1.7 GHz of its 2.4 GHz under it (power-limited; in-kernel stamps, DESIGN.md 3.2)",
        })
    else:
        roofline = {
            "kernel": KERNEL_NAMES[form], "bound": "mfma", "achieved": f32_equiv_tflops, "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
            "frac": f32_equiv_tflops / PEAK_MFMA_F32_TFLOPS, "traffic": traffic, "kernel_ms": skin_ms, "launches_timed": launches, "hbm": hbm,
            "note": "exact fp32 on v_mfma_f32_32x32x2_f32 (1/16 of the 16-bit MFMA rate): the fp32 matrix pipe binds",
        }
    return roofline


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


def ik_cpu_baseline(model, faces, tp, tn, theta0, budget_s=8.0):
    """configs[2] on the host: one IK iteration per frame as the reference computes it — residual and Jacobian by libtorch
    autograd, one backward() per Jacobian row, through the reference's own compiled FK stages (oracle/_ref,
    node/node.cpp:823-869), then the fp64 normal equations and LLT (node/node.cpp:883-943, restated in oracle/cpu.py).
    Bounded sample: whole frame-iterations until the budget is spent (at least one)."""
    from oracle import cpu, ref

    if not ref.available():
        return None
    R = ref.RefModel(model)
    ref.lib().ref_set_num_threads(usable_cpus())
    cores = ref.lib().ref_get_num_threads()
    K = len(faces)
    done = 0
    t0 = time.perf_counter()
    while True:
        f = done % len(theta0)
        r = R.ik_eval(np.zeros(10, np.float32), theta0[f], faces, tp[f], tn[f], np.ones(K), np.ones(K), np.zeros(K), np.zeros(K),
                      np.full((K, 3), 1 / 3, np.float32))
        A, b = cpu.normal_equations(r["e"], r["J"], 75, 2 * K, 0)
        cpu.llt_solve(A, b)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 400:
            break
    return {"value": done / el, "unit": "IK iterations/s", "cores": int(cores), "kind": "reference",
            "sample": "%d frame-iterations (%.1f s) of configs[2]: 24 autograd backward() calls per frame through the reference's "
                      "compiled FK stages + fp64 normal equations and LLT" % (done, el)}


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
    ap.add_argument("--mocap-restarts", type=int, default=64, help="restarts of the capture fit IN ALL (BASELINE configs[3]: 64, sharded over the GPUs)")
    ap.add_argument("--mocap-frames", type=int, default=0, help="frames of the capture sequence to fit (0 = all 3163)")
    ap.add_argument("--vposer-frames", type=int, default=512, help="frames of the VPoser-latent IK leg IN ALL (BASELINE configs[4]: 512, sharded over the GPUs)")
    ap.add_argument("--preroll-steps", type=int, default=0,
                    help="extra untimed launches in front of the warm-up (profiling runs only: after idle the chip needs ~400 launches "
                         "= 25 ms of load to reach its steady clocks; the contract's region is never pre-rolled by default)")
    ap.add_argument("--sustained-steps", type=int, default=2000, help="launches of the long run reported as `sustained` (0 = skip)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the two-handle / two-stream leg reported as `pipelined`")
    ap.add_argument("--no-exact-form", "--no-side-form", dest="no_side_form", action="store_true",
                    help="skip the fp16x2 (SMPLPP_SKIN=h) leg reported as `within_tolerance_form`")
    ap.add_argument("--profile-steps", type=int, default=40, help="launches of the separate loop that times the fused kernel with HIP events")
    ap.add_argument("--model", default=os.environ.get("SMPLPP_MODEL"),
                    help="a real model in the reference's schema (smpl_male.npz / .json from scripts/preprocess.py:98-117; default: "
                         "$SMPLPP_MODEL): replaces the synthetic stand-in everywhere in this run, `data` becomes \"real\" (or "
                         "\"synthetic (model file)\" when the file holds the synthetic stand-in)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--all-ranks-on-device0", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses GPU 0 (use with --backend gloo)")
    args = ap.parse_args()

    from smplpp_amd import dist as D

    # `python bench.py --gpus N` with no launcher in front starts its own N ranks: fresh child processes, before anything in THIS
    # process touches the GPU (no torch import yet), rank 0's line forwarded, non-zero exit if any rank fails.  Under
    # torch.distributed.run (RANK / WORLD_SIZE set) this process is one of the ranks.  A WORLD_SIZE that contradicts --gpus is refused.
    if D.launch_plan(args.gpus) == "spawn":
        raise SystemExit(D.launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    import torch

    from smplpp_amd import model_io
    from smplpp_amd.smpl import SMPL

    rank, world, local = D.env_rank_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    if args.all_ranks_on_device0:
        local = 0
    torch.cuda.set_device(local)
    D.init_process_group((args.backend or "nccl") if world > 1 else None)

    # a real smpl_male.{npz,json} overrides the synthetic model (the files are license-gated: none travels here)
    model = model_io.load_model(args.model) if args.model else model_io.synthetic_model()
    global V
    V = int(model["vertices_template"].shape[0])
    smpl = SMPL()
    smpl.setDevice("cuda:%d" % local)
    smpl.init(model)

    n = args.frames
    beta_h, theta_h = model_io.synthetic_inputs(n, seed=1 + rank)
    beta = torch.from_numpy(beta_h).cuda()
    theta = torch.from_numpy(theta_h).cuda()
    out = {"verts": torch.empty((n, V, 3), dtype=torch.float32, device="cuda")}

    def region(work):
        """The contract's bracket (dist.timed_region): synchronize + barrier, clock, work, synchronize, THIS rank's clock read, closing
        barrier, then max / min over ranks — no collective inside the clock.  Returns {"max" (the job's seconds), "min", "per_rank", ...}."""
        return D.timed_region(work, torch.cuda.synchronize)

    def timed(engine, steps):
        """`steps` launches of the FK step in the contract's bracket; spread over ranks (seconds)."""
        def work():
            for _ in range(steps):
                engine.launch(beta, theta, want=("verts",), out=out)
        return region(work)

    def rank_ms(spread, per=1.0):
        """per-rank milliseconds of a region (min / max / which rank was slowest) for the N > 1 lines"""
        return {"min": spread["min"] / per * 1e3, "max": spread["max"] / per * 1e3, "slowest_rank": spread["rank_of_max"]}

    def kernel_ms(engine, steps):
        """the fused kernel's own duration: a separate short loop with HIP events on the launch stream (never inside a timed region)"""
        engine.profileEnable(True)
        engine.profileRead()
        for _ in range(max(1, steps)):
            engine.launch(beta, theta, want=("verts",), out=out)
        torch.cuda.synchronize()
        launches, ms = engine.profileRead()
        engine.profileEnable(False)
        return launches, D.max_over_ranks(ms)

    # The contract's region (W warm-up launches, then exactly K timed ones) is taken TWICE.  Here, right behind model creation, the
    # chip is still ramping its clocks up (after idle it needs ~400 launches = 25 ms of load: 57-60 us per step falling to 49-52,
    # tools/fk_ramp.py, profiles/r03_a_fk_ramp.txt) — reported as `cold_start`.  The headline `value` is the same region timed
    # behind the IK legs below, i.e. on a chip that has been busy, which is the state a production pipeline keeps it in; with
    # `sustained` (2000 launches) beside it the line no longer depends on how short --steps is.
    for _ in range(args.preroll_steps + args.warmup):
        smpl.launch(beta, theta, want=("verts",), out=out)
    cold_elapsed = timed(smpl, args.steps)["max"]

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
        ik_t, ik_min = 0.0, 0.0
        res = {}
        for rep in range(reps + 1):
            solver.setTasks(face_idx=faces, vertex_weights=np.full((K, 3), 1 / 3, np.float32))
            solver.setConfig(np.zeros((args.ik_frames, 10), np.float32), theta0)
            sp = region(lambda: res.__setitem__("e2", solver.iterate(args.ik_iters)))
            if rep > 0:  # rep 0 is warm-up
                ik_t += sp["max"] / reps
                ik_min += sp["min"] / reps
        e2 = res["e2"]
        ik = {
            "value": world * args.ik_frames * args.ik_iters / ik_t, "unit": "IK iterations/s", "frames_per_gpu": args.ik_frames,
            "iters": args.ik_iters, "tasks": K, "ms_per_iter_batch": ik_t / args.ik_iters * 1e3,
            "ms_per_iter_batch_fastest_rank": ik_min / args.ik_iters * 1e3,
            "final_max_e_sqnorm": float(np.max(e2)), "final_median_e_sqnorm": float(np.median(e2)),
            "frames_below_1e-3": int((e2 < 1e-3).sum()),  # the normal terms make the problem non-convex: a start can end in a local minimum
            "workload": "configs[2]: 6-target IK (position + normal term per target), 50 iterations, direct theta (D = 87)",
        }

    # ---- configs[3]: sample_walk.c3d, every frame (tests/golden/sample_walk_full.npz: 3163 frames x 41 Baseline markers),
    # 64 restarts IN ALL sharded over the GPUs (dist.shard_sizes), each a serial warm-started chain: 32 iterations on frame 0
    # then one per frame (node.cpp:1369-1407), marker-thickness normal offsets, QP on; in both layouts (direct theta, and
    # the 44-d VPoser layout the reference forces on capture solves).  configs[4]: VPoser-latent IK, 512 frames IN ALL
    # sharded over the GPUs, 6 targets, 50 iterations (synthetic decoder: the real weights cannot travel).
    mocap_leg = vposer_leg = None
    if not args.no_ik and not args.no_extra:
        from smplpp_amd import mocap
        from smplpp_amd.ik import VPoserDecoder

        g = np.load(os.path.join(ROOT, "tests", "golden", "sample_walk_full.npz"))
        names = list(g["task_names"])
        mfaces = np.array([mocap.BASELINE41[nm] for nm in names], np.int64)
        Km = len(names)
        T = int(args.mocap_frames) if args.mocap_frames > 0 else g["points"].shape[0]
        pts = (g["points"][:T] - g["points"][0][g["valid"][0]].mean(axis=0) + np.array([0, -0.3, 0], np.float32)).astype(np.float32)
        mvalid = g["valid"][:T]
        # the job is defined by its GLOBAL inputs (seeded alike on every rank); a rank takes its contiguous slice, so restart r /
        # latent frame f is the same problem — and, chain_base / frame_base given, the same bits — on 1, 2, 4 or 8 GPUs
        rlo, rhi = D.shard_range(args.mocap_restarts, rank, world)
        R = rhi - rlo
        rng = np.random.default_rng(200)
        th0_all = np.zeros((args.mocap_restarts, 25, 3), np.float32)
        th0_all[:, 1:] = rng.normal(0, 0.03, (args.mocap_restarts, 24, 3))  # the restarts differ in their initial pose
        gv0_all = np.zeros((args.mocap_restarts, 44), np.float32)
        gv0_all[:, 6:38] = rng.normal(0, 0.05, (args.mocap_restarts, 32))
        # configs[4]'s hidden poses are drawn in LATENT space and decoded (below), so the targets lie inside the decoder's range and the
        # 50 timed iterations run on converging solves (round 4 drew joint angles directly: with random decoder weights such targets
        # are unreachable and every solve settled at a residual of 3e-3..6e-2)
        hidz_all = rng.normal(0, 0.5, (args.vposer_frames, 32)).astype(np.float32)
        if R > 0:
            th0 = np.ascontiguousarray(th0_all[rlo:rhi])
            ms = mocap.MocapMotionSolver(smpl, mfaces, np.full((Km, 3), 1 / 3, np.float32), restarts=R, chain_base=rlo)
            ms.solve(pts, mvalid, np.zeros(10, np.float32), th0, max_frames=2)  # warm-up of the code path
        res = {}
        msp = region(lambda: res.__setitem__("m", ms.solve(pts, mvalid, np.zeros(10, np.float32), th0)) if R > 0 else None)
        mt = msp["max"]
        if R > 0:
            thm, fr = res["m"]
        nfr = T
        iters = mocap.MocapMotionSolver.WARMUP_ITERS + nfr - 1
        skipped = int((mvalid.sum(axis=1) < Km // 2).sum())
        mocap_leg = {
            "value": args.mocap_restarts * nfr / mt, "unit": "solved capture frames/s", "ik_iterations_per_s": args.mocap_restarts * iters / mt,
            "restarts_total": args.mocap_restarts, "restarts_this_rank": R, "frames": nfr, "markers": Km, "seconds": mt,
            "frames_with_missing_markers": int((~mvalid).any(axis=1).sum()), "frames_skipped_below_20_valid": skipped,
            "finite": bool(np.isfinite(thm).all()) if R > 0 else True,
            "workload": "configs[3]: sample_walk.c3d, all %d frames x 41 markers, %d restarts sharded x%d, warm-started chains, box QP, "
                        "direct theta (D = 157); frame loop on the device (smplpp_ik_solve_sequence)" % (nfr, args.mocap_restarts, world),
        }

        # one IK iteration of a rank's chains in lock step: the job's (slowest rank's) period, and every rank's own — at N = 8 each rank
        # holds 8 chains and this IS the 8-chain figure the 1-GPU line reports as per_frame_us_at_8_chains
        mocap_leg["per_frame_us"] = mt / max(1, iters) * 1e6
        mocap_leg["per_frame_us_per_rank"] = [t / max(1, iters) * 1e6 for t in msp["per_rank"]]
        mocap_leg["chains_per_rank"] = D.shard_sizes(args.mocap_restarts, world)
        mocap_leg["scaling_note"] = ("a chain is serial along the sequence (node.cpp:1369-1407: one warm-started iteration per frame), so N GPUs "
                                     "shorten this leg only by the shorter per-frame period of 64 / N chains in lock step — its floor is %d "
                                     "iterations x per_frame_us_at_8_chains (on this line: 64-chain period / 8-chain period, about 1.25x for 8x "
                                     "the hardware); the leg scales in RESTARTS "
                                     "(more chains per GPU at the same period), not in time" % iters)
        # one GPU's share of the 8-GPU split (64 restarts -> 8 chains per GPU): the serial per-frame period the 8-GPU number is made of
        # (chains of a frame sequence cannot be parallelised along the sequence: node.cpp:1369-1407), measured here on ONE GPU
        if world == 1 and args.mocap_restarts >= 8:
            T8 = min(nfr, 600)
            ms8 = mocap.MocapMotionSolver(smpl, mfaces, np.full((Km, 3), 1 / 3, np.float32), restarts=8)
            ms8.solve(pts[:T8], mvalid[:T8], np.zeros(10, np.float32), np.ascontiguousarray(th0_all[:8]), max_frames=2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ms8.solve(pts[:T8], mvalid[:T8], np.zeros(10, np.float32), np.ascontiguousarray(th0_all[:8]))
            torch.cuda.synchronize()
            t8 = time.perf_counter() - t1
            mocap_leg["per_frame_us_at_8_chains"] = t8 / (mocap.MocapMotionSolver.WARMUP_ITERS + T8 - 1) * 1e6
            mocap_leg["frames_timed_at_8_chains"] = T8
            del ms8

        vp = VPoserDecoder(VPoserDecoder.synthetic_params(seed=3), device=local)
        # the reference's own capture setting (node.cpp:316-322 forces VPoser + QP on): 44-d layout, D = 44 + 2 * 41
        if R > 0:
            msv = mocap.MocapMotionSolver(smpl, mfaces, np.full((Km, 3), 1 / 3, np.float32), restarts=R, vposer=vp, chain_base=rlo)
            gv0 = np.ascontiguousarray(gv0_all[rlo:rhi])
            msv.solve(pts, mvalid, np.zeros(10, np.float32), gv0, max_frames=2)
        msvp = region(lambda: res.__setitem__("v", msv.solve(pts, mvalid, np.zeros(10, np.float32), gv0)) if R > 0 else None)
        mtv = msvp["max"]
        if R > 0:
            thv, frv = res["v"]
        mocap_leg["vposer_latent"] = {
            "value": args.mocap_restarts * nfr / mtv, "unit": "solved capture frames/s", "seconds": mtv,
            "per_frame_us": mtv / max(1, iters) * 1e6, "per_frame_us_per_rank": [t / max(1, iters) * 1e6 for t in msvp["per_rank"]],
            "finite": bool(np.isfinite(thv).all()) if R > 0 else True,
            "workload": "same sequence with the 44-d VPoser layout the reference forces on capture solves (D = 126), synthetic decoder weights",
        }
        if world == 1 and args.mocap_restarts >= 8:  # the latent layout's share of the 8-GPU split, like per_frame_us_at_8_chains above
            T8 = min(nfr, 600)
            msv8 = mocap.MocapMotionSolver(smpl, mfaces, np.full((Km, 3), 1 / 3, np.float32), restarts=8, vposer=vp)
            gv8 = np.ascontiguousarray(gv0_all[:8])
            msv8.solve(pts[:T8], mvalid[:T8], np.zeros(10, np.float32), gv8, max_frames=2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            msv8.solve(pts[:T8], mvalid[:T8], np.zeros(10, np.float32), gv8)
            torch.cuda.synchronize()
            t8 = time.perf_counter() - t1
            mocap_leg["vposer_latent"]["per_frame_us_at_8_chains"] = t8 / (mocap.MocapMotionSolver.WARMUP_ITERS + T8 - 1) * 1e6
            del msv8

        Kv = 6
        vlo, vhi = D.shard_range(args.vposer_frames, rank, world)
        nv = vhi - vlo
        _, vfaces = reference_task_faces(Kv)
        vt, ev = 0.0, np.zeros(1)
        if nv > 0:
            hidv = np.zeros((nv, 25, 3), np.float32)
            hidv[:, 2:23] = vp.forward(np.ascontiguousarray(hidz_all[vlo:vhi]), frame_base=vlo)  # theta25 = [pos | root | vposer(z) | aa22 | aa23]
            hvv = smpl.launch(np.zeros((nv, 10), np.float32), hidv, want=("verts",))["verts"]
            tpv = hvv[:, model["face_indices"][vfaces] - 1].mean(axis=2)
            vs = IkSolver(smpl, nv, Kv, vposer=vp, frame_base=vlo)
            vs.setTasks(face_idx=vfaces, target_pos=tpv, phi_limit=np.zeros(Kv), normal_task_weight=np.zeros(Kv))
            g0 = np.zeros((nv, vs.theta_dim), np.float32)
        for rep in range(3):
            if nv > 0:
                vs.setTasks(face_idx=vfaces, vertex_weights=np.full((Kv, 3), 1 / 3, np.float32))
                vs.setConfig(np.zeros((nv, 10), np.float32), g0)
            sp = region(lambda: res.__setitem__("ev", vs.iterate(args.ik_iters)) if nv > 0 else None)
            if rep > 0:
                vt += sp["max"] / 2
        if nv > 0:
            ev = res["ev"]
        # convergence of the timed solves, over ALL ranks' frames (like the `ik` leg: the line shows whether the 50 iterations ran on
        # converging solves, not only the worst frame)
        ev_all = D.gather_values(np.asarray(ev, np.float64) if nv > 0 else np.zeros(0))
        vposer_leg = {
            "value": args.vposer_frames * args.ik_iters / vt, "unit": "IK iterations/s", "frames_total": args.vposer_frames,
            "frames_this_rank": nv, "iters": args.ik_iters, "tasks": Kv,
            "ms_per_iter_batch": vt / args.ik_iters * 1e3, "final_max_e_sqnorm": float(np.max(ev_all)),
            "final_median_e_sqnorm": float(np.median(ev_all)), "frames_below_1e-3": int((ev_all < 1e-3).sum()),
            "workload": "configs[4]: VPoser-latent IK (32-d latent + decoder in the loop, 44-d layout, D = 56), %d frames sharded x%d, "
                        "synthetic decoder weights, targets = task points of hidden latents (reachable)" % (args.vposer_frames, world),
        }

    # ---- the headline FK region (see `cold_start` above): W warm-up launches, then exactly K timed ones
    for _ in range(args.warmup):
        smpl.launch(beta, theta, want=("verts",), out=out)
    elapsed_sp = timed(smpl, args.steps)
    elapsed = elapsed_sp["max"]
    launches, skin_ms = kernel_ms(smpl, args.profile_steps)
    # the steady clock: after idle the chip ramps its clocks UP over the first ~400 launches (57 -> 49 us per step over 25 ms,
    # tools/fk_ramp.py), so a short --steps region right behind model creation is timed on a chip that has not settled; the
    # same step is timed once more over a long run, after the contract's region
    sustained = None
    if args.sustained_steps > 0:
        sus_t = timed(smpl, args.sustained_steps)["max"]
        sustained = {"launches": args.sustained_steps, "ms_per_step": sus_t / args.sustained_steps * 1e3,
                     "value": world * n * args.sustained_steps / sus_t, "unit": "FK evals/s",
                     "note": "the same step over a long run behind the contract's region: the steady-clock figure (after idle the chip "
                             "needs ~25 ms of load to ramp up; a 20-step region right after start-up is timed during that ramp)"}
    # the same steps as a caller streaming batches would issue them: two model handles (each its own workspace and output buffer) on two
    # streams, steps alternating between them, so that the pose step, launch, prologue and tail of one step run in the shadow of the
    # other's fused kernel.  Same kernels, same outputs (tools/fk_two_streams.py compares them); reported BESIDE `value`, never as it.
    pipelined = None
    if args.sustained_steps > 0 and not args.no_pipelined:
        smpl2 = SMPL()
        smpl2.setDevice("cuda:%d" % local)
        smpl2.init(model)
        out2 = {"verts": torch.empty((n, V, 3), dtype=torch.float32, device="cuda")}
        lanes = [(smpl, out, torch.cuda.Stream()), (smpl2, out2, torch.cuda.Stream())]

        def piped(steps):
            def work():
                for i in range(steps):
                    eng, o, st = lanes[i % 2]
                    with torch.cuda.stream(st):
                        eng.launch(beta, theta, want=("verts",), out=o)
            return region(work)
        piped(args.warmup + 100)
        pt = piped(args.sustained_steps)["max"]
        pipelined = {"launches": args.sustained_steps, "ms_per_step": pt / args.sustained_steps * 1e3,
                     "value": world * n * args.sustained_steps / pt, "unit": "FK evals/s", "handles": 2, "streams": 2,
                     "note": "the `sustained` run with its steps alternating between two model handles on two streams (a double-buffered "
                             "caller): throughput of the same kernels when one step's pose / launch / tail hides behind the other's "
                             "fused kernel; a step's latency is unchanged. Not the headline."}
        del smpl2, out2, lanes
    # the same step on the fp16x2 form (SMPLPP_SKIN=h at model creation): two fp16 pieces per operand (22 bits), three MFMA products
    # per fp32 product, skinning on the matrix pipe too — inside the 1e-5 m bar (3e-7 m) but narrower than the reference's fp32, so it
    # is reported BESIDE the headline, never as it
    side = None
    form_env = (os.environ.get("SMPLPP_SKIN") or "e")[0]
    if not args.no_side_form and form_env == "e":
        orig_env = os.environ.get("SMPLPP_SKIN")
        os.environ["SMPLPP_SKIN"] = "h"  # read once, at model creation
        try:
            smpl_h = SMPL()
            smpl_h.setDevice("cuda:%d" % local)
            smpl_h.init(model)
        finally:
            if orig_env is None:
                del os.environ["SMPLPP_SKIN"]
            else:
                os.environ["SMPLPP_SKIN"] = orig_env
        for _ in range(args.warmup):
            smpl_h.launch(beta, theta, want=("verts",), out=out)
        sd_steps = max(args.steps, 200)
        sd_t = timed(smpl_h, sd_steps)["max"]
        sd_launches, sd_kms = kernel_ms(smpl_h, args.profile_steps)
        side = {"steps": sd_steps, "t": sd_t, "kernel_ms": sd_kms, "launches": sd_launches}
        del smpl_h

    # the only exchange of the path: the final gather of the results to rank 0 (every peer sends its block once, into its slot
    # of rank 0's array: dist.gather_rows), always timed when N > 1
    gather_ms = None
    counted = None
    if world > 1:
        counted = D.count_ranks()  # an all-reduce(SUM) of a one per rank through the collective library itself (on the device under RCCL)
        D.gather_rows(out["verts"][:1], world)  # one row per rank first: the point-to-point channels are set up outside the timed gather
        res = {}
        gather_ms = region(lambda: res.__setitem__("full", D.gather_rows(out["verts"], n * world)))["max"] * 1e3
        res.clear()

    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    value = world * n * args.steps / elapsed
    # "synthetic" unless a model FILE was given; a file is "real" only when it is not the synthetic stand-in written to disk
    # (tests do that to walk the on-disk schema): the stand-in's template is recognised by content
    data_label = "synthetic"
    if args.model:
        syn = model_io.synthetic_model()
        same = syn["vertices_template"].shape == model["vertices_template"].shape and np.array_equal(
            np.asarray(syn["vertices_template"], np.float32), np.asarray(model["vertices_template"], np.float32))
        data_label = "synthetic (model file)" if same else "real"
    form = form_env if form_env in ("e", "h", "b", "p", "v") else "e"
    traffic_all = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):  # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/pmc_fk.sh, profiles/README.md)
        try:
            traffic_all = json.load(open(tpath))
        except Exception:
            traffic_all = {}
    roofline = roofline_object(form, n, skin_ms, launches, traffic_all.get("skin_kernel_%s_hbm_bytes_per_launch_n%d" % (form, n)))
    # `traffic` is a constant of the committed profile collection (separate rocprofv3 --pmc passes cannot run inside this program): which one
    # the same bytes against the whole STEP (pose kernel + fused kernel + launch gaps): what `value` is made of
    alg_bytes = ALG_BYTES_CONST + ALG_BYTES_PER_FRAME * n
    roofline["step_hbm_frac"] = alg_bytes / (ms_per_step * 1e-3) / (PEAK_HBM_GBS * 1e9)
    roofline["traffic_source"] = {"file": "profiles/traffic.json", "tag": traffic_all.get("tag"),
                                  "kernel_us_in_that_run": traffic_all.get("skin_kernel_%s_rocprofv3_avg_us_same_run" % form)}
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
        "dtype": DTYPES.get(form, "f32"),
        "data": data_label,
        "config": {
            "workload": "configs[1]: batch-%d random beta/theta FK+LBS per GPU, %s "
                        "(%d verts, 24 joints, 207 pose / 10 shape PCs), HBM-resident in/out"
                        % (n, ("model file " + os.path.basename(args.model)) if args.model else "synthetic SMPL-shaped model", V),
            "frames_per_gpu": n, "parallelism": "frames sharded x%d, no data-path collective" % world,
        },
        "roofline": roofline,
        "model_source": ("file " + os.path.basename(args.model) + (" (from $SMPLPP_MODEL)" if os.environ.get("SMPLPP_MODEL") == args.model and "--model" not in sys.argv else "")) if args.model else "built in (model_io.synthetic_model)",
        #  where the contract's W + K region sits in this run, so rounds compare like for like: r01-r02 lines timed it
        # right behind model creation (= today's `cold_start`), r03 on behind the IK legs
        "value_region": "W warm-up + K timed launches BEHIND the IK legs (a chip that has been busy); the same region right behind "
                        "model creation is `cold_start`; a 2000-launch run is `sustained`",
    }
    line["cold_start"] = {"ms_per_step": cold_elapsed / args.steps * 1e3, "value": world * n * args.steps / cold_elapsed, "unit": "FK evals/s",
                          "note": "the same W + K launches timed right behind model creation, while the chip is still ramping its clocks "
                                  "up; `value` is timed behind the IK legs (a chip that has been busy), `sustained` over a long run"}
    if sustained is not None:
        line["sustained"] = sustained
    if pipelined is not None:
        line["pipelined"] = pipelined
    if side is not None:
        sd_ms = side["t"] / side["steps"] * 1e3
        line["within_tolerance_form"] = {
            "kernel": KERNEL_NAMES["h"], "dtype": DTYPES["h"], "value": world * n * side["steps"] / side["t"], "unit": "FK evals/s",
            "steps": side["steps"], "ms_per_step": sd_ms, "kernel_ms": side["kernel_ms"],
            "roofline": roofline_object("h", n, side["kernel_ms"], side["launches"], traffic_all.get("skin_kernel_h_hbm_bytes_per_launch_n%d" % n)),
            "note": "the same 1024-frame step with SMPLPP_SKIN=h at model creation: operands as two fp16 pieces (22 bits), within the 1e-5 m "
                    "bar with 15-30x margin but narrower than the reference's fp32 — a side figure; the headline `value` is the exact form",
        }
    if ik is not None:
        line["ik"] = ik
    if mocap_leg is not None:
        line["mocap"] = mocap_leg
    if vposer_leg is not None:
        line["vposer_ik"] = vposer_leg
    if gather_ms is not None:
        line["final_gather_ms"] = gather_ms
        line.update(counted)  # "collective_backend" ("nccl" = RCCL, or "gloo" in rehearsals), "ranks_counted_by_allreduce"
        line["ms_per_step_ranks"] = rank_ms(elapsed_sp, args.steps)
        line["timing"] = "per rank: synchronize + barrier, clock, K launches, synchronize, clock; max over ranks (no collective inside the clock)"
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(model, n)
        if ik is not None:
            try:
                ikb = ik_cpu_baseline(model, faces, tp, tn, theta0)
            except Exception as e:  # the reference build is test infrastructure: its absence must not break the bench line
                sys.stderr.write("ik cpu_baseline unavailable: %s\n" % e)
                ikb = None
            if ikb is not None:
                ik["cpu_baseline"] = ikb
    print(json.dumps(line))


if __name__ == "__main__":
    main()
