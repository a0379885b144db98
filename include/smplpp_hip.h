/* smplpp_hip.h — C ABI of libsmplpp_hip.so, the MI355X (gfx950) engine behind the reference's smplpp::SMPL /
 * smplpp::IkTask API.  Plain pointers and sizes only; no torch / libtorch types.
 *
 * The reference has no FFI: its boundary is the C++ class API of libsmplpp.so
 * (/root/reference/include/smplpp/SMPL.h:241-269, include/smplpp/IkTask.h:20-84) and its one caller
 * node/node.cpp.  Each entry point below names the reference interface it stands in for; include/smplpp/SMPL.h and
 * include/smplpp/IkTask.h in this repository re-expose the reference's class names on top of it.
 *
 * Conventions
 *  - every function returns SMPLPP_OK (0) or an error code; smplpp_last_error() gives the message (thread-local).
 *    The C++ shim turns non-zero into `throw smplpp::Exception` to keep smpl_error semantics
 *    (include/smplpp/toolbox/Exception.h:48-49).
 *  - arrays are row-major with the reference's shapes.  `space` says where CALLER buffers live:
 *    SMPLPP_HOST (pageable/pinned host memory; the call stages and synchronises) or SMPLPP_DEVICE (HIP device
 *    memory on the model's device; the call only enqueues work on `stream` and returns).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *  - face ids are 0-based in this ABI (IkTask::faceIdx_ is 0-based; the model FILE is 1-based and is converted at
 *    create time like src/SMPL.cpp:520 does at every use).
 *  - one caller thread per handle (the reference is single-threaded, node/node.cpp:1414).
 */
#ifndef SMPLPP_HIP_H
#define SMPLPP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMPLPP_JOINT_NUM 24        /* include/smplpp/definition/def.h:10 */
#define SMPLPP_SHAPE_BASIS_DIM 10  /* def.h:11 */
#define SMPLPP_POSE_BASIS_DIM 207  /* def.h:12 */
#define SMPLPP_LATENT_DIM 32       /* def.h:14 */
#define SMPLPP_THETA_DIM 75        /* 3 * (JOINT_NUM + 1), node/node.cpp:787 */
#define SMPLPP_LATENT_POSE_DIM 44  /* LATENT_DIM + 12, node/node.cpp:42 */

enum
{
  SMPLPP_OK = 0,
  SMPLPP_ERR_INVALID = 1, /* bad shape / argument: the reference's smpl_error("...", "Cannot ...") cases */
  SMPLPP_ERR_HIP = 2,     /* HIP runtime failure, including "no GPU" */
  SMPLPP_ERR_NUMERIC = 3, /* "LLT has numerical issue!" node/node.cpp:934-937 */
  SMPLPP_ERR_STATE = 4    /* call order (e.g. a getter before any launch) */
};

enum
{
  SMPLPP_HOST = 0,
  SMPLPP_DEVICE = 1
};

typedef struct smplpp_model smplpp_model;   /* stands in for smplpp::SMPL            (SMPL.h:140-270) */
typedef struct smplpp_ik smplpp_ik;         /* the IK loop state of node/node.cpp:645-1002, batched over frames */
typedef struct smplpp_vposer smplpp_vposer; /* stands in for smplpp::VPoserDecoder   (VPoser.h:53-90) */

const char * smplpp_last_error(void);
/* Number of HIP devices visible; SMPLPP_ERR_HIP (and *count = 0) when there is none. */
int smplpp_device_count(int * count);

/* ------------------------------------------------------------------ model: SMPL::setDevice/setModelPath/init */
/* Replaces SMPL::init (src/SMPL.cpp:560-643) minus the JSON parse, which stays on the host side: the seven arrays
 * of scripts/preprocess.py:98-117, host pointers.  vertex_num may differ from 6890 (the reference makes VERTEX_NUM
 * a variable, def.h:9).  Re-lays the blend bases out for the fused kernel, folds the joint regressor, builds the
 * adjacent-face table (:620-640). */
int smplpp_model_create(int64_t vertex_num, int64_t face_num, const float * vertices_template /*[V,3]*/,
                        const float * shape_blend_shapes /*[V,3,10]*/, const float * pose_blend_shapes /*[V,3,207]*/,
                        const float * joint_regressor /*[24,V]*/, const float * weights /*[V,24]*/,
                        const int64_t * kinematic_tree /*[2,24]*/, const int32_t * face_indices_1based /*[F,3]*/,
                        int device, smplpp_model ** out);
int smplpp_model_destroy(smplpp_model * m);
/* vertex_num, face_num, and the number of skinning weights kept per vertex (4, 8 or 24 = dense). */
int smplpp_model_info(const smplpp_model * m, int64_t * vertex_num, int64_t * face_num, int * weights_per_vertex,
                      int * device);

/* ------------------------------------------------------------------ FK: SMPL::launch + getters */
/* SMPL::launch(beta [n,10], theta [n,25,3]) (src/SMPL.cpp:671-737): theta[:,0,:] is the root translation,
 * theta[:,1:,:] the 24 axis-angles.  Any output may be NULL:
 *   verts  [n,V,3]      SMPL::getVertex          (:492-506)
 *   joints [n,24,3]     SMPL::getRestJoint       (:457-471)
 *   xforms [n,24,4,4]   WorldTransformation::getTransformation (relative transforms G')
 *   rest   [n,V,3]      SMPL::getRestShape
 * Two kernels: pose/chain, then the fused blend-shape GEMM + linear blend skinning.  The fused kernel computes in the reference's
 * arithmetic (every fp32 operand of the contraction carried exactly, as three bf16 pieces on the matrix pipe; fp32 accumulate; the
 * skinning in fp32 FMAs: src/BlendShape.cpp:762-765, src/LinearBlendSkinning.cpp:463-467).  SMPLPP_SKIN in the environment of
 * smplpp_model_create selects another form for every launch on the model: h (fp16x2 operand pieces, 22 bits, 3e-7 m: the form
 * the IK loops' internal forward passes use by default), b (round 1's bf16x3 kernel), p / v (fp32 MFMA). */
int smplpp_fk(smplpp_model * m, int64_t n, const float * beta, const float * theta, float * verts, float * joints,
              float * xforms, float * rest, int space, void * stream);
/* Input range of the fp16x2 form (SMPLPP_SKIN=h): |beta| < 1023 and relative transforms whose
 * translations stay within 16 x the template's extent (65504 / sG).  Outside it the operand pieces overflow fp16 and the
 * vertices of the frame are not finite, where the reference and the default form (and SMPLPP_SKIN=b|p|v) stay finite.  A launch that
 * meets such an operand sets bit 0 of the model's status word: a host-space smplpp_fk returns SMPLPP_ERR_NUMERIC itself;
 * an enqueue-only (device-space) caller reads it here — the call synchronises `stream`, returns the word and clears it. */
int smplpp_fk_status(smplpp_model * m, int * bits, void * stream);

/* Measurement hook (bench.py): while enabled, every launch of the fused blend-shape + skinning kernel is bracketed by
 * HIP events on the stream it is launched on; smplpp_profile_read waits for them, returns the number of launches
 * and the mean kernel duration in milliseconds since the last read, and clears the record. */
int smplpp_profile_enable(smplpp_model * m, int enable);
int smplpp_profile_read(smplpp_model * m, int64_t * launches, double * mean_skin_kernel_ms);

/* Stage-level entry points with the reference's stage semantics on arbitrary inputs (the Tester.cpp KATs feed
 * non-rotation matrices and 4x4 transforms with a non-trivial last row).  Host or device pointers.
 *   BlendShape::blend            src/BlendShape.cpp:620-647     JointRegression::regress   src/JointRegression.cpp:507-532
 *   WorldTransformation::transform  src/WorldTransformation.cpp:421-468
 *   LinearBlendSkinning::skinning   src/LinearBlendSkinning.cpp:445-483 (root_pos may be NULL) */
int smplpp_stage_blend_shape(int device, int64_t vertex_num, int64_t n, const float * beta, const float * theta24,
                             const float * shape_basis, const float * pose_basis, float * shape_blend, float * pose_blend,
                             float * pose_rot, int space, void * stream);
int smplpp_stage_joint_regression(int device, int64_t vertex_num, int64_t n, const float * template_shape,
                                  const float * joint_regressor, const float * shape_blend, const float * pose_blend,
                                  float * rest_shape, float * joints, int space, void * stream);
int smplpp_stage_world_transformation(int device, int64_t n, const int64_t * kinematic_tree, const float * joints,
                                      const float * pose_rot, float * xforms, int space, void * stream);
int smplpp_stage_skinning(int device, int64_t vertex_num, int64_t n, const float * weights, const float * rest_shape,
                          const float * xforms, const float * root_pos, float * verts, int space, void * stream);

/* ------------------------------------------------------------------ mesh queries on posed vertices */
/* SMPL::calcNormal (src/SMPL.cpp:518-525) and SMPL::calcVertexNormal (:527-535) for lists of ids, on frame-major
 * vertices [n,V,3]; outputs [n,count,3].  Adjacent faces are summed in ascending face id. */
int smplpp_face_normals(smplpp_model * m, int64_t n, const float * verts, int64_t count, const int64_t * face_ids,
                        float * normals, int space, void * stream);
int smplpp_vertex_normals(smplpp_model * m, int64_t n, const float * verts, int64_t count, const int64_t * vertex_ids,
                          float * normals, int space, void * stream);
/* igl::point_mesh_squared_distance as called at node/node.cpp:982: for each of n frames, K query points against that
 * frame's posed mesh.  face [n,K] (0-based), closest [n,K,3], sqdist [n,K] (nullable). */
int smplpp_closest_points(smplpp_model * m, int64_t n, const float * verts, int64_t K, const float * points,
                          int64_t * face, float * closest, float * sqdist, int space, void * stream);
/* SMPL::calcVertexNormal (src/SMPL.cpp:527-535) for EVERY vertex of every frame: normals [n,V,3]. */
int smplpp_mesh_vertex_normals(smplpp_model * m, int64_t n, const float * verts, float * normals, int space, void * stream);
/* The sweep grid of node/node.cpp:1023-1073 for ONE frame of posed vertices [V,3]: cells of GRID_SCALE = 0.025 m
 * (toolbox/GridUtils.hpp:28) from getGridIdxFloor(min) to getGridIdxCeil(max) per axis (:46-60) -> grid_min [3] (cell
 * index of the first cell), grid_num [3]; cells are ordered x outermost, z innermost like the reference's loops (:1037-1048).
 * winding [cap] (nullable) = generalized winding number of the mesh at each cell position (igl::winding_number, :1052),
 * inside [cap] (nullable) = winding number > 0.5 on the REAL-valued number.  Deviation, deliberate: the reference stores
 * igl::winding_number's result in an Eigen::VectorXi (:1051-1057), which truncates towards zero before the `> 0.5` test, so
 * there a cell counts as inside only when the number reaches 1 (0.99999 of an interior point becomes 0); compare `winding`
 * with 1 - eps yourself to reproduce that list.  *cells = the grid's cell count; at most `cap` cells are evaluated (call with
 * cap = 0 to size the arrays).  Non-finite or absurd (beyond +-25 km) vertices: SMPLPP_ERR_NUMERIC. */
int smplpp_sweep_grid(smplpp_model * m, const float * verts, int32_t * grid_min, int32_t * grid_num, int64_t cap,
                      float * winding, uint8_t * inside, int64_t * cells, int space, void * stream);
/* SMPL::getAdjacentFaces (src/SMPL.cpp:537-540): host copy; returns the count in *count, fills up to cap. */
int smplpp_adjacent_faces(const smplpp_model * m, int64_t vertex, int64_t cap, int64_t * faces, float * weights,
                          int64_t * count);

/* ------------------------------------------------------------------ IK: IkTask + the loop of node/node.cpp */
/* A batch of n independent frames, each with K tasks in the caller's std::map order (node/node.cpp:47,798).
 * Unknown layout per frame: [theta (75, or 44 with a VPoser) | phi (2K) | beta (10 when optimised)] (:787-791). */
int smplpp_ik_create(smplpp_model * m, int64_t n, int64_t K, smplpp_vposer * vposer /*nullable*/, smplpp_ik ** out);
int smplpp_ik_destroy(smplpp_ik * s);
/* This solver holds frames [frame_base, frame_base + n) of a larger job (contiguous shards per GPU).  Everything a
 * frame computes is independent of the frames beside it; the one kernel that orders its fp32 sums by frame position (the VPoser
 * decoder's Jacobian) takes the position from the GLOBAL index, so a frame's trajectory has the same bits on 1, 2, 4 or 8 GPUs.
 * Default 0.  No reference counterpart (single device, node/node.cpp:372). */
int smplpp_ik_set_frame_base(smplpp_ik * s, int64_t frame_base);
/* IkTask public fields (IkTask.h:54-84), struct-of-arrays over [n,K]; any pointer may be NULL = keep current.
 * Defaults match the header: weights 1, phiLimit 0.04, normalOffset 0, vertexWeights 1/3, targetNormal +Z. */
int smplpp_ik_set_tasks(smplpp_ik * s, const int64_t * face_idx /*[n,K]*/, const float * vertex_weights /*[n,K,3]*/,
                        const float * target_pos /*[n,K,3]*/, const float * target_normal /*[n,K,3]*/,
                        const double * pos_task_weight /*[n,K]*/, const double * normal_task_weight /*[n,K]*/,
                        const double * phi_limit /*[n,K]*/, const double * normal_offset /*[n,K]*/, int space);
/* g_beta [n,10] and g_theta [n,theta_dim] (node/node.cpp:44-45, :377). */
int smplpp_ik_set_config(smplpp_ik * s, const float * beta, const float * theta, int space);
int smplpp_ik_get_config(smplpp_ik * s, float * beta, float * theta, int space);
int smplpp_ik_get_tasks(smplpp_ik * s, int64_t * face_idx, float * vertex_weights, float * tangents /*[n,K,3,2]*/,
                        float * actual_pos /*[n,K,3]*/, float * actual_normal /*[n,K,3]*/, int space);
/* One evaluation of node/node.cpp:750-877 for every frame: forward, tangents + vertex weights refresh (:803-804),
 * residual e [n,4K] and the analytic Jacobian J [n,4K,D] (fp64, row-major) that replaces the per-row autograd
 * backward() of :823-869.  D = theta_dim + 2K + (optimize_beta ? 10 : 0). */
int smplpp_ik_eval(smplpp_ik * s, int optimize_beta, double * e, double * J, int space, void * stream);
/* `iters` repetitions of the loop body :704-1001 on every frame: eval, A = J^T J + damping (:883-904), solve
 * (enable_qp: box QP :909-930, else LLT :931-939), config update (:945-968), mesh re-projection (:970-1001).
 * optimize_beta_from >= 0 mirrors solveMocapBody: beta optimised and phi limits live from that iteration (:655,:695).
 * Frames whose number of tasks with pos_task_weight > 0 is < min_valid skip the solve (:785).
 * e_sqnorm [n] (nullable) receives |e|^2 of the last evaluation. */
int smplpp_ik_iterate(smplpp_ik * s, int iters, int enable_qp, int optimize_beta_from, int64_t min_valid,
                      double * e_sqnorm, int space, void * stream);
/* The frame loop of solveMocapMotion (node/node.cpp:1369-1407 with the per-frame target switch of :681-700) for n chains
 * (sequences / restarts) in lock step, enqueued without a host round trip per frame: for t = 0..T-1 the marker targets of
 * frame t become the task targets (valid == 0: target 0 and posTaskWeight_ 0, else posTaskWeight_ 1), then `warmup_iters`
 * (t == 0; the reference advances once ikIter > 30) or `iters_per_frame` iterations of smplpp_ik_iterate's loop body run,
 * warm-started from the previous frame; theta_out[t] receives g_theta after frame t. Every other task field (faces,
 * weights, offsets, phi limits) is what smplpp_ik_set_tasks left. Layouts: target_pos [T,n,K,3], valid [T,n,K] (bytes),
 * theta_out [T,n,theta_dim]. */
int smplpp_ik_solve_sequence(smplpp_ik * s, int64_t T, const float * target_pos, const uint8_t * valid, int warmup_iters,
                             int iters_per_frame, int enable_qp, int64_t min_valid, float * theta_out, int space, void * stream);
/* The same loop when every chain fits the SAME capture — the multi-restart fit the reference runs by hand (one sample_walk.c3d, many
 * initial poses; node/node.cpp:1369-1407 once per restart): target_pos [T,K,3] and valid [T,K] are given once and handed to all n
 * chains by the frame switch on the device; theta_out [T,n,theta_dim] as above. Results are those of smplpp_ik_solve_sequence with
 * the targets repeated n times, bit for bit. */
int smplpp_ik_solve_sequence_shared(smplpp_ik * s, int64_t T, const float * target_pos, const uint8_t * valid, int warmup_iters,
                                    int iters_per_frame, int enable_qp, int64_t min_valid, float * theta_out, int space, void * stream);
/* Vertices of the last forward inside the solver [n,V,3] (SMPL::getVertex after the loop's launch). */
int smplpp_ik_get_vertices(smplpp_ik * s, float * verts, int space, void * stream);
/* Outcome of the solves so far, per frame: bit 0 = the LAST solve failed with the reference's "LLT has numerical issue!"
 * (node/node.cpp:934-937; that frame's update was skipped), bit 1 = some solve failed since the configuration was set /
 * the sequence started, bit 2 = an evaluation since the tasks were last set met a task WITH A NORMAL TERM (normal weight or normal
 * offset) on a vertex of more than 16 adjacent faces: the analytic Jacobian differentiates vertex normals through per-face tables
 * whose width smplpp_model_create takes from the topology — 12 faces per vertex, 16 when some vertex has more (SMPL's own mesh:
 * at most 9; src/SMPL.cpp:527-535 puts no bound on it) — so beyond 16 those rows are unsupported — the solve SKIPS the update of such a frame
 * (no caller moves on a truncated Jacobian) and smplpp_ik_set_tasks clears the bit (it belongs to the tasks; the next evaluation
 * raises it again where it still applies); position-only tasks are unaffected and any model gets its solver.  Bit 3 = a forward
 * pass inside a loop on this model met an operand outside the fp16x2 form's range since the last smplpp_ik_set_config (one word per
 * model: every frame of the batch carries it).  Host-space eval / iterate / solve_sequence calls return SMPLPP_ERR_NUMERIC
 * (bits 0, 1) or SMPLPP_ERR_INVALID (bit 2) themselves; enqueue-only (SMPLPP_DEVICE) callers have no return value to inspect and
 * read it here (waits for `stream` first). flags [n]. */
int smplpp_ik_get_status(smplpp_ik * s, int32_t * flags, int space, void * stream);

/* Streams and sharing.  A model owns ONE workspace (pose coefficients, relative transforms of the last forward pass) that
 * smplpp_fk and every smplpp_ik built on the model write: all work on one model handle must be issued in stream order on
 * ONE caller stream (or be separated by the caller's own synchronisation).  The setters (set_tasks / set_config) are
 * host-synchronous on the NULL stream; call them only when no enqueue-only call on the solver is still in flight. */

/* ------------------------------------------------------------------ multi-GPU: the final gather
 * Frames (FK, independent IK) and restarts (capture fitting) shard across GPUs with no exchange in the data path; the only
 * collective is the gather of result rows at the end.  `comm` is an ncclComm_t (RCCL) the HOST created — one rank per GPU,
 * ncclCommInitRank / ncclCommInitAll — passed as an opaque pointer; RCCL is resolved at run time (the symbols already in the
 * process, else librccl.so), so single-GPU users carry no dependency on it.  Every rank passes its block `send`
 * [rows_per_rank[rank], row_floats] and receives all blocks in rank order in `recv` [sum rows_per_rank, row_floats] (device
 * pointers; `send` may alias its own slot of `recv`).  Equal blocks travel as one ncclAllGather, ragged ones as grouped
 * broadcasts; enqueued on `stream`.  What it replaces: nothing in the single-GPU reference — this is the SURVEY's proposed
 * `smplpp_gather(comm, ...)`. */
int smplpp_gather(void * comm, const float * send, float * recv, const int64_t * rows_per_rank, int world, int rank,
                  int64_t row_floats, void * stream);
/* The same exchange to ONE rank (the usual consumer of a job's results): every other rank sends its block once, over its own
 * xGMI link, straight into its slot of root's `recv` (grouped ncclSend / ncclRecv; `recv` may be NULL elsewhere) — an eighth
 * of the all-gather's traffic at eight ranks.  In place on root when `send` is its own slot of `recv`. */
int smplpp_gather_to_root(void * comm, const float * send, float * recv, const int64_t * rows_per_rank, int world, int rank, int root,
                          int64_t row_floats, void * stream);
/* Start-up check of the RCCL binding (resolved at run time, see above): `rank` exchanges `count_floats` floats with ITSELF in one
 * ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd — the calls smplpp_gather_to_root makes for a peer — so a host learns before
 * the job, not at its end, whether the library it loaded speaks this ABI.  Device pointers, `send` != `recv`; enqueued on `stream`. */
int smplpp_gather_selfcheck(void * comm, int rank, const float * send, float * recv, int64_t count_floats, void * stream);
/* Where each rank's block lies in the gathered array: offsets[world + 1] in floats (the last entry is the total).  Host-only
 * arithmetic shared by the two collectives; callable without a GPU. */
int smplpp_gather_offsets(const int64_t * rows_per_rank, int world, int64_t row_floats, int64_t * offsets);

/* ------------------------------------------------------------------ VPoser decoder (src/VPoser.cpp) */
/* VPoserDecoderImpl (VPoser.h:53-90): Linear(32,512) LeakyReLU Dropout(eval) Linear(512,512) LeakyReLU
 * Linear(512,126) -> 6D -> rotation (Gram-Schmidt, :129-141) -> axis-angle (:25-120).  Weights are
 * torch::nn::Linear layout [out,in] as loadParamsFromJson stores them (:169-238), host pointers. */
int smplpp_vposer_create(int device, const float * w0 /*[512,32]*/, const float * b0 /*[512]*/,
                         const float * w1 /*[512,512]*/, const float * b1 /*[512]*/, const float * w2 /*[126,512]*/,
                         const float * b2 /*[126]*/, smplpp_vposer ** out);
int smplpp_vposer_destroy(smplpp_vposer * v);
/* VPoserDecoderImpl::forward (src/VPoser.cpp:163-167): z [n,32] -> axis-angles [n,21,3];
 * jac (nullable) [n,63,32] = d(out)/dz, the quantity autograd supplies in node/node.cpp:761-772. */
int smplpp_vposer_forward(smplpp_vposer * v, int64_t n, const float * z, float * out, float * jac, int space,
                          void * stream);
/* The same for a SHARD of a larger job: frame_base = the global index of this call's frame 0.  The Jacobian kernel orders its
 * fp32 sums by the frame's global index, so a latent decodes to the same bits whether the job runs on 1 GPU or is cut over
 * 2, 4 or 8 (smplpp_vposer_forward is frame_base = 0).  No reference counterpart: the reference is single-device
 * (node/node.cpp:372); SURVEY 8(e). */
int smplpp_vposer_forward_at(smplpp_vposer * v, int64_t n, int64_t frame_base, const float * z, float * out, float * jac, int space,
                             void * stream);
/* convertRotMatToAxisAngle (src/VPoser.cpp:25-120): rot [n,3,3] -> aa [n,3]. */
int smplpp_rotmat_to_axis_angle(int device, int64_t n, const float * rot, float * aa, int space, void * stream);

#ifdef __cplusplus
}
#endif
#endif /* SMPLPP_HIP_H */
