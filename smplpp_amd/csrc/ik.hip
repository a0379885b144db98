// The IK loop of the reference (node/node.cpp:645-1002) for a batch of independent frames, one workgroup per frame.
//
//  ik_eval_kernel    node.cpp:798-877.  The reference gets each Jacobian row by a full reverse-mode autograd pass
//                    through the 6890-vertex FK graph (3-4 passes per task); here the same derivative is the analytic
//                    forward-mode Jacobian of only the vertices the tasks touch: per frame the chain
//                    derivatives dG'_i/dtheta_{j,k} of all 72 rotation columns are built once in LDS (41 KB: three columns per
//                    ancestor depth), one tree level per step, then every
//                    task reads them for its face vertices (and their 1-rings when a normal is involved: per-face ring
//                    tables built with the model).  In the VPoser layout it also writes the latent rows (node.cpp:761-772).
//  ik_solve_kernel   node.cpp:883-968: A = J^T J + damping in fp64 built straight from J staged through LDS, right-looking
//                    Cholesky of the packed lower-triangular augmented system in LDS (fp64) or the box QP by a primal
//                    active set around it, config update, query points for the re-projection.
//  proj_scan/finish  node.cpp:970-1001: exact closest point on the posed mesh (all 13776 faces, sphere-culled against the
//                    distance to each task's current face; each face gathered once per frame for all K queries), new
//                    face id and area-ratio weights.
// All fp32 where the reference is fp32 (FK, task geometry, autograd gradients), fp64 where it is fp64 (Eigen).
#include "ik_types.h"
#include "ik_eval.h"
#include "ik_solve.h"
#include "ik_proj.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>

using namespace smplpp_hip;

struct smplpp_ik
{
  smplpp_model * m = nullptr;
  smplpp_vposer * vp = nullptr;
  int64_t n = 0, K = 0;
  int64_t frame_base = 0; // global index of frame 0 when this solver holds a shard of a larger job (smplpp_ik_set_frame_base)
  int theta_dim = TD75;
  TaskArrays ta{};
  float *theta = nullptr, *beta = nullptr, *theta25 = nullptr, *vjac = nullptr;
  float *verts = nullptr, *rest = nullptr, *joints = nullptr, *poserot = nullptr, *pts = nullptr;
  double *e = nullptr, *J = nullptr, *Jl = nullptr, *e2 = nullptr, *xout = nullptr;
  int *skip = nullptr, *status = nullptr, *sticky = nullptr, *list_cnt = nullptr, *list_f = nullptr;
  int * range_word = nullptr; // this solver's own "an operand left the fp16x2 form's range" word (status bit 3): its loops' forward passes report here
  int32_t * roles = nullptr; // [DMAX][EVAL_NT] the chain-derivative entries of every thread of ik_eval_kernel (slot u: row u)
  float * list_d = nullptr;
  std::vector<void *> owned;
  bool have_eval = false;
  // re-projection beside the solve: when no task's surface coordinates can move (phiLimit_ <= 0 everywhere, or the
  // motion stage's forced zero, node.cpp:699) the query points are the actual positions the evaluation already wrote,
  // so the face scan does not depend on the solve and runs on a second stream while the solve is in flight
  // The posed mesh is double-buffered so that the side stream can still read iteration i's mesh (scan + finish) while the
  // main stream already writes iteration i+1's; the join sits in front of iteration i+1's evaluation, the first kernel that
  // reads what the finish kernel wrote (face, weights). Both events ride on a kernel's own completion signal
  // (hipExtLaunchKernelGGL): a separate hipEventRecord costs the recording stream ~7 us per iteration.
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool phi_locked = false;
  bool side_pending = false; // a finish kernel is in flight on the side stream; ev_join / the join flag marks its end
  // hand-over between the two streams through device flags (wg_signal + hipStreamWaitValue32) instead of events: [0] fork
  // flag, [16] its workgroup counter, [32] join flag, [48] its counter (one 64-byte line each)
  int * dbg_buf = nullptr;
  unsigned * sig = nullptr;
  unsigned tick_fork = 0, tick_join = 0, tick_done = 0; // ([64] / [80]: the solve's "configuration final" flag and its counter)
  bool use_flags = false;
  // Latent layout with few frames (a capture fit's chains: one decoder workgroup per frame, most of the chip idle).  The decoder's
  // VALUE is all the pose step, the fused kernel and the evaluation's direct rows need; its Jacobian (two thirds of the kernel's
  // 29 us) only the pull-back behind them.  So when another iteration follows, the side stream — idle once scan + finish are done,
  // well before the solve ends — waits for the solve's "configuration final" flag and makes the NEXT iteration's Jacobian there
  // (vposer_jac2_kernel<NF, false>, `out` null, raising the join flag at its end), while the main stream decodes the value with the
  // kernel's value-only instantiation (<NF, true>: the same bits, 18 us), poses, skins and joins: the Jacobian is there when the
  // evaluation (which pulls its rows back through it) starts.  Same kernels' arithmetic, another schedule: bit-identical
  // (tests/test_mocap_gpu.py).  SMPLPP_IK_LATENT_SPLIT=0/1 (read at creation) overrides the n <= 128 rule.
  double last_enqueue_us = 0.0; // host time of the last smplpp_ik_solve_sequence's / smplpp_ik_iterate's enqueue loop
  bool latent_split = false;
  bool jac_ahead = false; // the decoder Jacobian of the CURRENT configuration is (being) made on the side stream; the join flag follows it
  // development switches, read ONCE at creation (never in the per-call path): SMPLPP_DEBUG_SYNC, SMPLPP_IK_DBG_STOP,
  // SMPLPP_IK_OVERLAP=0 (re-projection behind the solve on one stream), SMPLPP_SCAN_BLOCKS
  bool dbg_sync = false, overlap_ok = true;
  int dbg_stop = 0;
  int64_t scan_blocks = 1536;
  int scan_form = -1; // development switch SMPLPP_SCAN_FORM (read at creation): 0 forces the K > 8 instantiations of the face scan
  float * vbuf[2] = {nullptr, nullptr};
  int vcur = 0;
};

template<class T>
static hipError_t dalloc(smplpp_ik * s, T ** p, size_t count)
{
  hipError_t e = hipMalloc((void **)p, sizeof(T) * std::max<size_t>(count, 1));
  if(e == hipSuccess) s->owned.push_back(*p);
  return e;
}

static ModelView view_of(const smplpp_model * m)
{
  ModelView mv;
  mv.faces = m->faces;
  mv.adjOff = m->adjOff;
  mv.adjFace = m->adjFace;
  mv.parent = m->parent;
  mv.wIdx = m->wIdx;
  mv.wVal = m->wVal;
  mv.wSum = m->wSum;
  mv.Pvm = m->Pvm;
  mv.Svm = m->Svm;
  mv.JS = m->JS;
  mv.faceRing = m->faceRing;
  mv.faceMap = m->faceMap;
  mv.anc = m->anc;
  mv.nlev = m->nlev;
  mv.V = m->V;
  mv.maxw = m->maxw;
  return mv;
}

extern "C" int smplpp_ik_destroy(smplpp_ik * s)
{
  if(!s) return SMPLPP_OK;
  (void)hipSetDevice(s->m->device);
  if(s->side) (void)hipStreamSynchronize(s->side);
  if(s->ev_fork) (void)hipEventDestroy(s->ev_fork);
  if(s->ev_join) (void)hipEventDestroy(s->ev_join);
  if(s->side) (void)hipStreamDestroy(s->side);
  for(void * p : s->owned) (void)hipFree(p);
  delete s;
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_set_frame_base(smplpp_ik * s, int64_t frame_base)
{
  if(!s || frame_base < 0) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_set_frame_base: bad argument");
  s->frame_base = frame_base;
  s->jac_ahead = false;
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_create(smplpp_model * m, int64_t n, int64_t K, smplpp_vposer * vposer, smplpp_ik ** out)
{
  if(!m || !out || n <= 0 || K <= 0) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: bad argument");
  *out = nullptr;
  if(m->F <= 0) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: the model has no faces");
  if(m->nlev > DMAX) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: kinematic trees deeper than 12 levels are not supported");
  if(m->V > 65535) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: at most 65535 vertices are supported (ring tables hold 16-bit ids)");
  if(!m->faceRing || !m->faceMap || !m->anc) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: the model carries no ring tables");
  // the evaluation advances the chain derivatives one tree level per step with one thread per (joint of the level, ancestor
  // depth, axis, row): the per-thread entries of every level (ik_eval_kernel: role), and every level must fit the workgroup
  // (SMPL: at most 5 x 9 x 9 = 405 of 1024)
  std::vector<int32_t> roles((size_t)DMAX * EVAL_NT, -1);
  {
    std::vector<int> depth(NJ, 0);
    std::vector<std::vector<int>> at(NJ + 1);
    for(int i = 0; i < NJ; i++)
    {
      depth[i] = i ? depth[m->h_parent[i]] + 1 : 0;
      at[depth[i]].push_back(i);
    }
    // every (joint i, ancestor depth da <= depth(i), axis, row) once, dealt to the threads ROUND-ROBIN: entry e goes to thread
    // e % EVAL_NT as its e / EVAL_NT-th (SMPL: 1.2 k entries, two per thread at most).  (Rounds 1-3 filled row L with the entries
    // of the joints at tree level L, the order their level-by-level recurrence needed; the closed form has no order, and with
    // that filling the first wavefronts held nine entries each while the last held none.)
    size_t e = 0;
    for(int L = 0; L < DMAX; L++)
    {
      const int per = 9 * (L + 1);
      for(int t = 0; t < (int)at[L].size() * per; t++, e++)
      {
        if(e >= roles.size()) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: kinematic tree too wide for the evaluation kernel");
        const int ji = t / per, rem = t % per, da = rem / 9, a9 = rem % 9, i = at[L][ji];
        roles[e] = i | ((m->h_parent[i] & 31) << 5) | ((3 * da + a9 / 3) << 10) | ((a9 % 3) << 16) | ((da == L ? 1 : 0) << 18);
      }
    }
  }
  if(K > PROJ_MAXK) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: at most 48 tasks per frame are supported");
  if(TD75 + 2 * K + NB > MAXD)
    return fail(SMPLPP_ERR_INVALID, "smplpp_ik_create: too many tasks for the in-LDS solver (75 + 2K + 10 must be <= 181)");
  // (a vertex with more than 16 adjacent faces — the widest table the evaluation is instantiated for — does not stop the solver from being created: position-only tasks anywhere and
  // normal-term tasks away from such a vertex are unaffected; a normal-term task that touches one is reported when it is evaluated)
  HIP_TRY(hipSetDevice(m->device));
  smplpp_ik * s = new smplpp_ik();
  s->m = m;
  s->vp = vposer;
  s->n = n;
  s->K = K;
  s->theta_dim = vposer ? TD44 : TD75;
  {
    const char * e;
    s->dbg_sync = getenv("SMPLPP_DEBUG_SYNC") != nullptr;
    if((e = getenv("SMPLPP_IK_DBG_STOP"))) s->dbg_stop = atoi(e);
    if((e = getenv("SMPLPP_IK_OVERLAP"))) s->overlap_ok = e[0] != '0';
    if(s->dbg_sync) s->overlap_ok = false;
    // workgroups of the face scan.  Few frames: 1536 in all (a capture fit's 64 chains: 24 chunks of 574 faces per frame, measured
    // against 9 / 18 / 36 chunks).  256 frames: TWO chunks per frame — the scan then runs beside kernels that fill the chip
    // themselves (solve, pose, FK: one workgroup per frame or per CU), and fewer, longer scan workgroups take less from them than
    // many short ones: configs[2] 89.2 -> 85.0 us per iteration in three alternating pairs on one box (6 chunks before); 512 frames
    // keep their three (2 and 3 measured level).  SMPLPP_SCAN_BLOCKS overrides.
    s->scan_blocks = (n >= 256 && n < 512) ? 2 * n : 1536;
    if(n >= 512 && K <= 8 && (m->F + 767) / 768 <= 32) s->scan_blocks = n * ((m->F + 767) / 768); // (chunks of at most 768 faces: the 80-register instantiation, below)
    if((e = getenv("SMPLPP_SCAN_BLOCKS"))) s->scan_blocks = atoll(e);
    if((e = getenv("SMPLPP_SCAN_FORM"))) s->scan_form = atoi(e);
    s->latent_split = vposer != nullptr && n <= 128 && s->dbg_stop == 0;
    // (a debug stop ends the solve kernel in front of its "configuration final" flag: never beside the schedule that waits for it)
    if((e = getenv("SMPLPP_IK_LATENT_SPLIT"))) s->latent_split = vposer != nullptr && e[0] != '0' && s->dbg_stop == 0;
  }
  const size_t nk = (size_t)n * K;
  const size_t Dmax = TD75 + 2 * K + NB;
#define A_(field, count)                                         \
  do                                                             \
  {                                                              \
    hipError_t _e = dalloc(s, &s->field, (count));               \
    if(_e != hipSuccess)                                         \
    {                                                            \
      int _rc = hip_fail(_e, #field, __FILE__, __LINE__);        \
      smplpp_ik_destroy(s);                                      \
      return _rc;                                                \
    }                                                            \
  } while(0)
  A_(ta.face, nk);
  A_(ta.vw, nk * 3);
  A_(ta.tang, nk * 6);
  A_(ta.tpos, nk * 3);
  A_(ta.tnrm, nk * 3);
  A_(ta.posw, nk);
  A_(ta.nrmw, nk);
  A_(ta.philim, nk);
  A_(ta.noff, nk);
  A_(ta.apos, nk * 3);
  A_(ta.anrm, nk * 3);
  A_(ta.hint, nk);
  A_(ta.roww, nk * 2);
  A_(theta, (size_t)n * s->theta_dim);
  A_(beta, (size_t)n * NB);
  A_(theta25, (size_t)n * 75);
  A_(vbuf[0], (size_t)n * m->V * 3);
  A_(vbuf[1], (size_t)n * m->V * 3);
  A_(rest, (size_t)n * m->V * 3);
  A_(joints, (size_t)n * NJ * 3);
  A_(poserot, (size_t)n * NJ * 9);
  A_(pts, nk * 3);
  A_(e, nk * 4);
  A_(J, nk * 4 * Dmax);
  A_(e2, (size_t)n);
  A_(xout, (size_t)n * Dmax);
  A_(roles, roles.size());
  A_(list_cnt, nk);
  A_(list_d, nk * PROJ_LIST);
  A_(list_f, nk * PROJ_LIST);
  A_(skip, (size_t)n);
  A_(status, (size_t)n);
  A_(sticky, (size_t)n);
  A_(range_word, 1);
  s->ta.flags = s->sticky;
  if(vposer)
  {
    A_(Jl, nk * 4 * Dmax);
    A_(vjac, (size_t)n * 63 * 32);
  }
#undef A_
  // (from here on a failure releases the solver with everything it already owns: arrays, side stream, events)
#define S_TRY(expr)                                            \
  do                                                           \
  {                                                            \
    hipError_t _e = (expr);                                    \
    if(_e != hipSuccess)                                       \
    {                                                          \
      int _rc = hip_fail(_e, #expr, __FILE__, __LINE__);       \
      smplpp_ik_destroy(s);                                    \
      return _rc;                                              \
    }                                                          \
  } while(0)
  // IkTask defaults (include/smplpp/IkTask.h:54-84)
  auto grid = [](size_t c) { return dim3((unsigned)((c + 255) / 256)); };
  S_TRY(hipMemset(s->ta.face, 0, sizeof(int32_t) * nk));
  fill_f32_kernel<<<grid(nk * 3), 256>>>(s->ta.vw, 1.0f / 3.0f, nk * 3);
  fill_f32_kernel<<<grid(nk * 6), 256>>>(s->ta.tang, 0.0f, nk * 6);
  fill_f32_kernel<<<grid(nk * 3), 256>>>(s->ta.tpos, 0.0f, nk * 3);
  fill_nrm_kernel<<<grid(nk * 3), 256>>>(s->ta.tnrm, nk * 3);
  fill_f32_kernel<<<grid(nk), 256>>>(s->ta.posw, 1.0f, nk);
  fill_f32_kernel<<<grid(nk), 256>>>(s->ta.nrmw, 1.0f, nk);
  fill_f32_kernel<<<grid(nk), 256>>>(s->ta.philim, 0.04f, nk);
  fill_f32_kernel<<<grid(nk), 256>>>(s->ta.noff, 0.0f, nk);
  S_TRY(hipMemset(s->theta, 0, sizeof(float) * n * s->theta_dim));
  S_TRY(hipMemset(s->theta25, 0, sizeof(float) * n * 75));
  S_TRY(hipMemset(s->beta, 0, sizeof(float) * n * NB));
  S_TRY(hipMemset(s->skip, 0, sizeof(int) * n));
  S_TRY(hipMemcpy(s->roles, roles.data(), sizeof(int32_t) * roles.size(), hipMemcpyHostToDevice));
  S_TRY(hipMemset(s->list_cnt, 0, sizeof(int) * nk));
  S_TRY(hipMemset(s->status, 0, sizeof(int) * n));
  S_TRY(hipMemset(s->sticky, 0, sizeof(int) * n));
  S_TRY(hipMemset(s->range_word, 0, sizeof(int)));
  s->verts = s->vbuf[0];
  // (default priority: a lowest-priority side stream — tried against the scan being dispatched ahead of the solve — halved the
  // latent-IK leg of bench.py, where several solvers' streams exist; what fixes that order is the solve kernel's own "all my
  // workgroups run" flag, see ik_solve_kernel)
  S_TRY(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
  S_TRY(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
  S_TRY(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
  {
    S_TRY(dalloc(s, &s->sig, 128));
    S_TRY(hipMemset(s->sig, 0, sizeof(unsigned) * 128));
    // stream memory operations are optional in HIP: probe once (flag 0 >= 0 is satisfied at once); SMPLPP_IK_EVENTS=1 keeps events
    const char * e = getenv("SMPLPP_IK_EVENTS");
    if(!(e && e[0] != '0'))
    {
      s->use_flags = hipStreamWaitValue32(s->side, s->sig, 0u, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
      (void)hipGetLastError();
    }
    if(!s->use_flags) s->latent_split = false; // (the hand-overs of that schedule are flags)
  }
  S_TRY(hipDeviceSynchronize());
#undef S_TRY
  *out = s;
  return SMPLPP_OK;
}

// copy caller array -> solver array with optional conversion
template<class Src, class Dst, class Conv>
static int set_array(const Src * src, Dst * dst, size_t count, int space, Conv conv)
{
  if(!src) return SMPLPP_OK;
  In<Src> in;
  HIP_TRY(in.init(src, count, space, nullptr));
  conv(in.d, dst, (int64_t)count);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_set_tasks(smplpp_ik * s, const int64_t * face_idx, const float * vertex_weights, const float * target_pos,
                                   const float * target_normal, const double * pos_task_weight, const double * normal_task_weight,
                                   const double * phi_limit, const double * normal_offset, int space)
{
  if(!s) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_set_tasks: null solver");
  int rc = check_space(space, "smplpp_ik_set_tasks");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  const size_t nk = (size_t)s->n * s->K;
  if(face_idx && space == SMPLPP_HOST)
    for(size_t i = 0; i < nk; i++)
      if(face_idx[i] < 0 || face_idx[i] >= s->m->F) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_set_tasks: face index out of range");
  auto g = [](int64_t c) { return dim3((unsigned)((c + 255) / 256)); };
  auto cpf = [](const float * a, float * b, int64_t c) { (void)hipMemcpy(b, a, sizeof(float) * c, hipMemcpyDeviceToDevice); };
  auto cvd = [&](const double * a, float * b, int64_t c) { f64_to_f32_kernel<<<g(c), 256>>>(a, b, c); };
  auto cvi = [&](const int64_t * a, int32_t * b, int64_t c) { i64_to_i32_kernel<<<g(c), 256>>>(a, b, c); };
  // bit 2 of the status word (a normal term on a vertex beyond MAXADJ faces) belongs to the tasks the evaluation met: new faces,
  // weights or normal terms start clean, and the next evaluation raises it again where it still applies
  if(face_idx || normal_task_weight || normal_offset || vertex_weights)
  {
    clear_bits_kernel<<<g((int64_t)s->n), 256>>>(s->sticky, 4, (int64_t)s->n);
    HIP_TRY(hipGetLastError());
  }
  if((rc = set_array(face_idx, s->ta.face, nk, space, cvi))) return rc;
  if((rc = set_array(vertex_weights, s->ta.vw, nk * 3, space, cpf))) return rc;
  if((rc = set_array(target_pos, s->ta.tpos, nk * 3, space, cpf))) return rc;
  if((rc = set_array(target_normal, s->ta.tnrm, nk * 3, space, cpf))) return rc;
  if((rc = set_array(pos_task_weight, s->ta.posw, nk, space, cvd))) return rc;
  if((rc = set_array(normal_task_weight, s->ta.nrmw, nk, space, cvd))) return rc;
  if((rc = set_array(phi_limit, s->ta.philim, nk, space, cvd))) return rc;
  if((rc = set_array(normal_offset, s->ta.noff, nk, space, cvd))) return rc;
  if(phi_limit)
  {
    std::vector<float> h(nk);
    HIP_TRY(hipMemcpy(h.data(), s->ta.philim, sizeof(float) * nk, hipMemcpyDeviceToHost));
    bool locked = true;
    for(size_t i = 0; i < nk && locked; i++) locked = !(h[i] > 0.0f);
    s->phi_locked = locked;
  }
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_set_config(smplpp_ik * s, const float * beta, const float * theta, int space)
{
  if(!s) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_set_config: null solver");
  int rc = check_space(space, "smplpp_ik_set_config");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  hipMemcpyKind kind = space == SMPLPP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  if(beta) HIP_TRY(hipMemcpy(s->beta, beta, sizeof(float) * s->n * NB, kind));
  if(theta) HIP_TRY(hipMemcpy(s->theta, theta, sizeof(float) * s->n * s->theta_dim, kind));
  // status bit 1 (smplpp_ik_get_status) reports failures "since the configuration was set": a new configuration starts clean
  s->jac_ahead = false; // (a Jacobian made ahead belongs to the configuration it was made for)
  HIP_TRY(hipMemset(s->status, 0, sizeof(int) * s->n));
  HIP_TRY(hipMemset(s->sticky, 0, sizeof(int) * s->n));
  if(s->range_word) HIP_TRY(hipMemset(s->range_word, 0, sizeof(int))); // (status bit 3: same lifetime; this solver's own word)
  if(theta && s->vp) // latent layout: the entries that pass through to theta25 (the decoder fills the rest at every evaluation)
  {
    ik_splice_kernel<<<dim3((unsigned)((s->n * 75 + 255) / 256)), 256>>>(s->theta, nullptr, s->theta25, s->n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(nullptr));
  }
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_get_config(smplpp_ik * s, float * beta, float * theta, int space)
{
  if(!s) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_get_config: null solver");
  int rc = check_space(space, "smplpp_ik_get_config");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  HIP_TRY(hipDeviceSynchronize());
  hipMemcpyKind kind = space == SMPLPP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if(beta) HIP_TRY(hipMemcpy(beta, s->beta, sizeof(float) * s->n * NB, kind));
  if(theta) HIP_TRY(hipMemcpy(theta, s->theta, sizeof(float) * s->n * s->theta_dim, kind));
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_get_tasks(smplpp_ik * s, int64_t * face_idx, float * vertex_weights, float * tangents, float * actual_pos,
                                   float * actual_normal, int space)
{
  if(!s) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_get_tasks: null solver");
  int rc = check_space(space, "smplpp_ik_get_tasks");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t nk = (size_t)s->n * s->K;
  hipMemcpyKind kind = space == SMPLPP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if(face_idx)
  {
    Out<int64_t> o;
    HIP_TRY(o.init(face_idx, nk, space));
    i32_to_i64_kernel<<<dim3((unsigned)((nk + 255) / 256)), 256>>>(s->ta.face, o.d, (int64_t)nk);
    HIP_TRY(hipGetLastError());
    HIP_TRY(o.finish(nullptr));
    HIP_TRY(hipDeviceSynchronize());
  }
  if(vertex_weights) HIP_TRY(hipMemcpy(vertex_weights, s->ta.vw, sizeof(float) * nk * 3, kind));
  if(tangents) HIP_TRY(hipMemcpy(tangents, s->ta.tang, sizeof(float) * nk * 6, kind));
  if(actual_pos) HIP_TRY(hipMemcpy(actual_pos, s->ta.apos, sizeof(float) * nk * 3, kind));
  if(actual_normal)
  {
    // IkTask::calcActualNormal() evaluated on demand at the current task state (face, weights) and the last posed mesh
    ik_actual_normals_kernel<<<dim3((unsigned)((nk + 63) / 64)), 64>>>(view_of(s->m), s->ta, s->verts, (int)s->K, (int64_t)nk);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(actual_normal, s->ta.anrm, sizeof(float) * nk * 3, kind));
  }
  return SMPLPP_OK;
}

// forward + eval for all frames (enqueue only)
static int ik_forward_eval(smplpp_ik * s, int optimize_beta, int phi_live, int64_t min_valid, hipStream_t st, hipEvent_t eval_done = nullptr)
{
  smplpp_model * m = s->m;
  const int64_t n = s->n;
  const int K = (int)s->K;
  const float * th25 = s->theta;
  // latent_split: this configuration's decoder Jacobian is being made on the side stream (smplpp_ik::jac_ahead) — here only the value
  const bool jac_elsewhere = s->vp && s->jac_ahead;
  {
    TraceRange tr_fwd("forward SMPL"); // node.cpp:752-781 (the VPoser splice is inside that span there too)
    if(s->vp) // node.cpp:761-772
    {
      // the decoder writes its 63 angles straight into theta25[:, 6:69]; the pass-through entries (root translation / rotation,
      // joints 22-23) are kept current by whoever changes the configuration: smplpp_ik_set_config and the solve kernel's update
      int rc = jac_elsewhere ? vposer_forward_device(s->vp, n, s->theta + 6, TD44, s->theta25 + 6, 75, nullptr, st, s->frame_base, true)
                             : vposer_forward_device(s->vp, n, s->theta + 6, TD44, s->theta25 + 6, 75, s->vjac, st, s->frame_base);
      if(rc) return rc;
      th25 = s->theta25;
    }
    s->vcur ^= 1;
    s->verts = s->vbuf[s->vcur];
    int rc = fk_device(m, n, s->beta, th25, s->verts, s->joints, nullptr, s->rest, s->poserot, st, RANGE_INTERNAL, s->range_word); // node.cpp:777
    if(rc) return rc;
  }
  TraceRange tr_eval("calculate IK matrices"); // node.cpp:796-881
  if(s->side_pending) // the previous iteration's re-projection (side stream) wrote the faces / weights read from here on
  {
    if(s->use_flags)
      HIP_TRY(hipStreamWaitValue32(st, s->sig + 32, s->tick_join, hipStreamWaitValueGte, 0xffffffffu));
    else
      HIP_TRY(hipStreamWaitEvent(st, s->ev_join, 0));
    s->side_pending = false;
  }
  s->jac_ahead = false; // (consumed by the evaluation below: the join above covers the Jacobian kernel, which raised it)
  const bool deep = m->nlev > 9; // (ik_eval_kernel's instantiations: see EvalPlan)
  const bool wide = m->madj > MAXADJ; // a topology with a vertex of 13..16 faces: 16-face tables, fewer normal tasks per group
  const size_t shmem = sizeof(float) * (deep ? (wide ? EvalPlan<DMAX, 64, 3>::L_END : EvalPlan<DMAX, 64, 3>::L_END)
                                             : (wide ? EvalPlan<9, 76, 4>::L_END : EvalPlan<9, 76, 6>::L_END)) + L_ANC_BYTES;
  static PerDeviceOnce once_eval[4];
  const void * kfn = deep ? (wide ? reinterpret_cast<const void *>(&ik_eval_kernel<DMAX, 64, 3, MAXADJ_WIDE>) : reinterpret_cast<const void *>(&ik_eval_kernel<DMAX, 64, 3>))
                          : (wide ? reinterpret_cast<const void *>(&ik_eval_kernel<9, 76, 4, MAXADJ_WIDE>) : reinterpret_cast<const void *>(&ik_eval_kernel<9, 76, 6>));
  HIP_TRY(lds_opt_in(once_eval[(deep ? 1 : 0) + (wide ? 2 : 0)], m->device, kfn, (int)shmem));
  int tsplit = (n < 256) ? (int)(256 / n) : 1; // one round of workgroups (one per CU: its LDS is the evaluation's)
  if(tsplit > K) tsplit = K;
  if(tsplit < 1) tsplit = 1;
  if(s->use_flags) eval_done = nullptr; // (flags mode: the fork is the solve kernel's start flag; the evaluation's end is signalled in events mode only)
#define EVAL_(DM, RC, NG, MA)                                                                                                              \
  hipExtLaunchKernelGGL((ik_eval_kernel<DM, RC, NG, MA>), dim3((unsigned)(n * tsplit)), dim3(EVAL_NT), shmem, st, nullptr, eval_done, 0,     \
                        view_of(m), s->ta, th25, (const float *)s->verts, (const float *)s->rest, (const float *)m->ws.Gp.as<float>(),     \
                        (const float *)s->joints, (const float *)s->poserot, K, optimize_beta, phi_live, (int)min_valid, s->pts, s->e,     \
                        s->J, s->skip, s->dbg_stop, tsplit, s->roles, s->vp ? (const float *)s->vjac : (const float *)nullptr,             \
                        s->vp ? s->Jl : (double *)nullptr)
  if(deep && wide)
    EVAL_(DMAX, 64, 3, MAXADJ_WIDE);
  else if(deep)
    EVAL_(DMAX, 64, 3, MAXADJ);
  else if(wide)
    EVAL_(9, 76, 4, MAXADJ_WIDE);
  else
    EVAL_(9, 76, 6, MAXADJ);
#undef EVAL_
  HIP_TRY(hipGetLastError());
  s->have_eval = true;
  return SMPLPP_OK;
}

static int ik_check_valence(smplpp_ik * s);

extern "C" int smplpp_ik_eval(smplpp_ik * s, int optimize_beta, double * e, double * J, int space, void * stream)
{
  if(!s) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_eval: null solver");
  int rc = check_space(space, "smplpp_ik_eval");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  rc = ik_forward_eval(s, optimize_beta, 1, 0, st);
  if(rc) return rc;
  const int64_t D = s->theta_dim + 2 * s->K + (optimize_beta ? NB : 0);
  hipMemcpyKind kind = space == SMPLPP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if(e) HIP_TRY(hipMemcpyAsync(e, s->e, sizeof(double) * s->n * s->K * 4, kind, st));
  if(J) HIP_TRY(hipMemcpyAsync(J, s->vp ? s->Jl : s->J, sizeof(double) * s->n * s->K * 4 * D, kind, st));
  if(space == SMPLPP_HOST)
  {
    HIP_TRY(hipStreamSynchronize(st));
    if((rc = ik_check_valence(s))) return rc;
  }
  return SMPLPP_OK;
}

// `iters` iterations enqueued on st (+ the solver's side stream); leaves the last re-projection pending on the side
// stream (s->side_pending) — the caller joins (ik_join) before anything else may touch the task arrays or the mesh.
// What the sequence driver wants done around the LAST of the iterations: the configuration after it recorded (by the solve
// kernel itself) and the NEXT frame's targets put in place (by the re-projection's finish kernel, wherever it runs: the
// evaluation that read the old targets is over by then, nothing else reads them, and the next evaluation waits for it).
struct SeqHook
{
  float * theta_record = nullptr;        // [n][theta_dim]
  const float * next_tpos = nullptr;     // [n][K][3]
  const uint8_t * next_valid = nullptr;  // [n][K]
  int shared = 0;                        // next_tpos / next_valid are [K] / [K][3]: one capture for every chain
};

// more_follows: the caller enqueues another iteration right behind this call's last one (the sequence driver, frame after frame)
static int ik_iterate_enqueue(smplpp_ik * s, int iters, int enable_qp, int optimize_beta_from, int64_t min_valid, hipStream_t st,
                              const SeqHook * hook = nullptr, bool more_follows = false)
{
  int rc = SMPLPP_OK;
  smplpp_model * m = s->m;
  const int K = (int)s->K;
  static PerDeviceOnce once_solve[5];
  HIP_TRY(lds_opt_in(once_solve[0], m->device, reinterpret_cast<const void *>(&ik_solve_kernel<false>), (int)SOLVE_LDS_MAX));
  HIP_TRY(lds_opt_in(once_solve[1], m->device, reinterpret_cast<const void *>(&ik_solve_kernel<true>), (int)SOLVE_LDS_MAX));
  HIP_TRY(lds_opt_in(once_solve[2], m->device, reinterpret_cast<const void *>(&ik_solve_kernel<false, 11>), (int)SOLVE_LDS_MAX));
  HIP_TRY(lds_opt_in(once_solve[3], m->device, reinterpret_cast<const void *>(&ik_solve_kernel<false, 5>), (int)SOLVE_LDS_MAX));
  HIP_TRY(lds_opt_in(once_solve[4], m->device, reinterpret_cast<const void *>(&ik_solve_kernel<false, 3>), (int)SOLVE_LDS_MAX));
  if(s->use_flags && (s->tick_fork > 0x7fff0000u || s->tick_join > 0x7fff0000u || s->tick_done > 0x7fff0000u))
  {
    // the hand-over flags carry iteration numbers compared with >=: start over long before they could wrap
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipStreamSynchronize(s->side));
    HIP_TRY(hipMemset(s->sig, 0, sizeof(unsigned) * 128));
    s->tick_fork = s->tick_join = s->tick_done = 0;
  }
  const bool dbg = s->dbg_sync;
  const int dbg_stop = s->dbg_stop;
  const bool overlap_ok = s->overlap_ok;
  const int64_t scan_blocks = s->scan_blocks;
#define DBG_SYNC(tag)                                                            \
  if(dbg)                                                                        \
  {                                                                              \
    fprintf(stderr, "[smplpp dbg] it %d: %s ...\n", it, tag);                    \
    HIP_TRY(hipStreamSynchronize(st));                                           \
    fprintf(stderr, "[smplpp dbg] it %d: %s done\n", it, tag);                   \
  }
  for(int it = 0; it < iters; it++)
  {
    const int opt_beta = (optimize_beta_from >= 0 && it >= optimize_beta_from) ? 1 : 0; // node.cpp:655
    const int phi_live = (optimize_beta_from >= 0) ? (it >= optimize_beta_from ? 1 : 0) : 1; // :693-700
    // x_phi = 0 for every task (no task's surface coordinates can move): the query points are the actual positions the
    // evaluation wrote, so scan + finish run on the side stream beside the solve and the next iteration's pose / FK
    const bool beside = overlap_ok && (!phi_live || s->phi_locked);
    rc = ik_forward_eval(s, opt_beta, phi_live, min_valid, st, beside ? s->ev_fork : nullptr);
    if(rc) return rc;
    DBG_SYNC("forward+eval");
    const int beta_dim = opt_beta ? NB : 0;
    // LDS plan: packed system + vectors, the rest (up to a 150 KB total) for the J row chunk
    const int D = s->theta_dim + 2 * K + beta_dim, rows = 4 * K;
    // the packed system is sized for the unknowns that CAN be free: a pinned phi (zero limit, node.cpp:567,699) never is,
    // which leaves 75 of the 157 unknowns of a 41-marker motion solve and room for its 164 Jacobian rows in two chunks
    const int m_dim = D - ((!phi_live || s->phi_locked) ? 2 * K : 0);
    // the box of node.cpp:911-929 bounds phi and d beta only: with every phi pinned and beta fixed (each motion-stage solve) no
    // variable has a finite bound, the QP's optimum IS the LLT solution (x = 0 + 1.0 (x_llt - 0): the same bits), and the kernel
    // takes its LLT exit instead of a ratio test and a bound check that cannot find anything (six barriers)
    const int qp_k = (enable_qp && !((!phi_live || s->phi_locked) && beta_dim == 0)) ? 1 : 0;
    // tiles of 16 the register-tiled factorisation covers (176 < m_dim + 1: all-LDS path).  5 (round 4): the motion solve of a capture
    // fit has 75 unknowns that can be free (+ the rhs row = 76 <= 80): 15 register tiles per thread instead of 21 in every rank-4
    // update of its 19 column steps, its own instantiation like 11 (one tile count per instantiation: DESIGN.md §3.3)
    // (and the same fit in the 44-d latent layout has 44 + 1 <= 48: 6 register tiles per thread in its 11 steps)
    const int ntr_primal = (m_dim + 1 <= 48) ? 3 : (m_dim + 1 <= 80) ? 5 : ((m_dim + 1 <= 96 || m_dim + 1 > 176) ? 6 : 11);
    // theta is never bound, so the free set keeps at least theta_dim unknowns: with fewer residual rows than that every pass
    // (also every active-set pass of the QP) takes the dual form.  (Decided up here because the kernel's LDS plan depends on the
    // instantiation's tile count: ik_solve_kernel<true> carries the default, 6.)
    const bool dual_shape = rows < s->theta_dim && rows <= 63 && D <= 192 && dbg_stop != 9;
    const int ntr = dual_shape ? 6 : ntr_primal;
    const size_t fixed = sizeof(double) * ((size_t)(m_dim + 1) * (m_dim + 2) / 2 + 7 * (size_t)D + 2 * (size_t)rows + 128 * (size_t)ntr + 4) + sizeof(int) * 2 * (size_t)D;
    const size_t budget = SOLVE_LDS_MAX;
    int chunk_rows = (int)((budget - fixed) / (sizeof(double) * (size_t)D));
    if(chunk_rows > rows) chunk_rows = rows;
    if(chunk_rows < 4) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_iterate: system too large for the in-LDS solver");
    const size_t solve_shmem = fixed + sizeof(double) * (size_t)chunk_rows * D;
    const bool dual_only = dual_shape && chunk_rows >= rows;
    if(dual_shape && !dual_only) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_iterate: system too large for the in-LDS solver");
    const bool last = hook && it == iters - 1;
    float * theta_record = last ? hook->theta_record : nullptr;
    const bool go = beside && s->use_flags; // the side stream's fork: raised by the solve kernel once all its workgroups run
    if(go) s->tick_fork++;
    // latent_split: the NEXT iteration's decoder Jacobian on the side stream, behind this solve's "configuration final" flag
    const bool ahead = s->latent_split && go && !opt_beta && (it + 1 < iters || more_follows);
    if(ahead) s->tick_done++;
    unsigned * const done_flag = ahead ? s->sig + 64 : (unsigned *)nullptr;
    unsigned * const done_counter = ahead ? s->sig + 80 : (unsigned *)nullptr;
#define SOLVE_(DO) ik_solve_kernel<DO><<<dim3((unsigned)s->n), dim3(256), solve_shmem, st>>>(                                              \
    s->ta, s->e, s->vp ? s->Jl : s->J, s->theta, s->beta, beside ? nullptr : s->pts, K, s->theta_dim, beta_dim, phi_live, qp_k, \
    s->vp ? 1 : 0, chunk_rows, s->skip, s->e2, s->status, s->sticky, s->xout, dbg_stop, m_dim, s->vp ? s->theta25 : (float *)nullptr, theta_record, \
    go ? s->sig : (unsigned *)nullptr, go ? s->sig + 16 : (unsigned *)nullptr, s->tick_fork, done_flag, done_counter, s->tick_done)
#define SOLVE11_(NTR_) ik_solve_kernel<false, NTR_><<<dim3((unsigned)s->n), dim3(256), solve_shmem, st>>>(                                              \
    s->ta, s->e, s->vp ? s->Jl : s->J, s->theta, s->beta, beside ? nullptr : s->pts, K, s->theta_dim, beta_dim, phi_live, qp_k, \
    s->vp ? 1 : 0, chunk_rows, s->skip, s->e2, s->status, s->sticky, s->xout, dbg_stop, m_dim, s->vp ? s->theta25 : (float *)nullptr, theta_record, \
    go ? s->sig : (unsigned *)nullptr, go ? s->sig + 16 : (unsigned *)nullptr, s->tick_fork, done_flag, done_counter, s->tick_done)
    {
      TraceRange tr_solve("solve IK"); // node.cpp:907-943
      if(dual_only)
        SOLVE_(true);
      else if(ntr == 11)
        SOLVE11_(11);
      else if(ntr == 5)
        SOLVE11_(5);
      else if(ntr == 3)
        SOLVE11_(3);
      else
        SOLVE_(false);
    }
#undef SOLVE_
#undef SOLVE11_
    HIP_TRY(hipGetLastError());
    DBG_SYNC("solve");
    {
      TraceRange tr_proj("project point"); // node.cpp:974-988
      const float * qpts = beside ? s->ta.apos : s->pts;
      hipStream_t pst = st;
      if(beside)
      {
        if(s->use_flags)
          HIP_TRY(hipStreamWaitValue32(s->side, s->sig, s->tick_fork, hipStreamWaitValueGte, 0xffffffffu));
        else
          HIP_TRY(hipStreamWaitEvent(s->side, s->ev_fork, 0));
        pst = s->side;
      }
      int chunks = (int)(scan_blocks / s->n);
      chunks = chunks < 1 ? 1 : (chunks > 32 ? 32 : chunks);
      const float * hint = beside ? s->ta.hint : nullptr; // the evaluation's distance is to the ACTUAL position
      const dim3 sg((unsigned)(s->n * chunks));
#define SCAN_(KPR, NBT_) proj_scan_kernel<KPR, NBT_><<<sg, dim3(256), 0, pst>>>(view_of(m), s->ta, s->verts, qpts, hint, m->F, K, chunks, s->skip, \
                                                                   s->list_cnt, s->list_d, s->list_f, dbg_stop)
      const bool small_chunk = (m->F + chunks - 1) / chunks <= 3 * 256; // (a thread then meets at most three faces)
      // K <= 8 with 512 frames and more (configs[4]): the K > 8 instantiation on chunks of at most 768 faces — 80 registers, six
      // wavefronts per SIMD instead of three — is the faster one beside the decoder, whose workgroups wait for the scan's to drain
      // (44.8 against 51.6 us, the latent loop -4 %); at 256 frames the queries-in-registers form stays ahead (81.5 against 84.2 us)
      const bool many = s->scan_form < 0 ? (s->n >= 512 && small_chunk) : s->scan_form == 0;
      if(K <= 8 && many && small_chunk) SCAN_(0, 3);
      else if(K <= 8 && many) SCAN_(0, CP_BATCH);
      else if(K <= 4) SCAN_(2, CP_BATCH);
      else if(K <= 8) SCAN_(4, CP_BATCH);
      else if(small_chunk) SCAN_(0, 3);
      else SCAN_(0, CP_BATCH);
#undef SCAN_
      HIP_TRY(hipGetLastError());
      int *& dbg_buf = s->dbg_buf; // (SMPLPP_DEBUG_SYNC only; owned by the solver, on its device)
      if(dbg && !dbg_buf) HIP_TRY(dalloc(s, &dbg_buf, 8));
      if(dbg) HIP_TRY(hipMemsetAsync(dbg_buf, 0, sizeof(int) * 8, st));
      int fsplit = (s->n < 256) ? (int)(256 / s->n) : 1;
      if(fsplit > K) fsplit = K;
      if(fsplit < 1) fsplit = 1;
      const bool join_flag = beside && s->use_flags && !ahead; // (ahead: the Jacobian kernel behind the finish kernel raises the join)
      if(beside && s->use_flags) s->tick_join++;
      hipExtLaunchKernelGGL(proj_finish_kernel, dim3((unsigned)(s->n * fsplit)), dim3(256), 0, pst, nullptr,
                            (beside && !s->use_flags) ? s->ev_join : nullptr, 0,
                            view_of(m), s->ta, (const float *)s->verts, qpts, m->F, K, (const int *)s->skip, s->list_cnt, s->list_d,
                            s->list_f, dbg ? dbg_buf : (int *)nullptr, fsplit, join_flag ? s->sig + 32 : (unsigned *)nullptr,
                            join_flag ? s->sig + 48 : (unsigned *)nullptr, s->tick_join, last ? hook->next_tpos : (const float *)nullptr,
                            last ? hook->next_valid : (const uint8_t *)nullptr, last ? hook->shared : 0);
      HIP_TRY(hipGetLastError());
      if(ahead)
      {
        HIP_TRY(hipStreamWaitValue32(s->side, s->sig + 64, s->tick_done, hipStreamWaitValueGte, 0xffffffffu));
        rc = vposer_forward_device(s->vp, s->n, s->theta + 6, TD44, nullptr, 75, s->vjac, s->side, s->frame_base, false, s->sig + 32,
                                   s->sig + 48, s->tick_join);
        if(rc) return rc;
        s->jac_ahead = true;
      }
      if(beside) s->side_pending = true;
      if(dbg)
      {
        int h[8];
        HIP_TRY(hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost));
        fprintf(stderr, "[smplpp dbg] it %d: project lists: tasks %d, empty %d, overflow %d, nan %d, max cnt %d\n", it, h[0], h[1], h[2], h[3], h[4]);
      }
    }
    DBG_SYNC("project");
  }
#undef DBG_SYNC
  return SMPLPP_OK;
}

static int ik_join(smplpp_ik * s, hipStream_t st)
{
  if(s->side_pending) // everything the caller does next on its stream is ordered behind the last re-projection
  {
    if(s->use_flags)
      HIP_TRY(hipStreamWaitValue32(st, s->sig + 32, s->tick_join, hipStreamWaitValueGte, 0xffffffffu));
    else
      HIP_TRY(hipStreamWaitEvent(st, s->ev_join, 0));
    s->side_pending = false;
  }
  return SMPLPP_OK;
}

static int ik_check_status(smplpp_ik * s, const int * flags)
{
  std::vector<int> h((size_t)s->n);
  HIP_TRY(hipMemcpy(h.data(), flags, sizeof(int) * s->n, hipMemcpyDeviceToHost));
  for(int64_t f = 0; f < s->n; f++)
    if(h[f] & 1) return fail(SMPLPP_ERR_NUMERIC, "LLT has numerical issue!"); // node.cpp:934-937
  return SMPLPP_OK;
}

// bit 2 of the sticky word: raised by the evaluation (see TaskArrays::flags)
static int ik_check_valence(smplpp_ik * s)
{
  std::vector<int> h((size_t)s->n);
  HIP_TRY(hipMemcpy(h.data(), s->sticky, sizeof(int) * s->n, hipMemcpyDeviceToHost));
  for(int64_t f = 0; f < s->n; f++)
    if(h[f] & 4)
      return fail(SMPLPP_ERR_INVALID, "a task with a normal term (normal weight or normal offset) touches a vertex with more than 12 adjacent "
                                      "faces: the Jacobian of such a term is not supported");
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_iterate(smplpp_ik * s, int iters, int enable_qp, int optimize_beta_from, int64_t min_valid,
                                 double * e_sqnorm, int space, void * stream)
{
  if(!s || iters < 0) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_iterate: bad argument");
  int rc = check_space(space, "smplpp_ik_iterate");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const auto enq_t0 = std::chrono::steady_clock::now();
  rc = ik_iterate_enqueue(s, iters, enable_qp, optimize_beta_from, min_valid, st);
  s->last_enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - enq_t0).count();
  if(rc)
  {
    (void)ik_join(s, st);
    return rc;
  }
  if((rc = ik_join(s, st))) return rc;
  if(e_sqnorm)
  {
    hipMemcpyKind kind = space == SMPLPP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    HIP_TRY(hipMemcpyAsync(e_sqnorm, s->e2, sizeof(double) * s->n, kind, st));
  }
  if(space == SMPLPP_HOST)
  {
    HIP_TRY(hipStreamSynchronize(st));
    if((rc = ik_check_valence(s))) return rc; // (first: such a frame's solve reports itself as skipped too)
    if((rc = ik_check_status(s, s->status))) return rc;
  }
  return SMPLPP_OK;
}

// node/node.cpp:1369-1407 with :681-700 — the frame loop of solveMocapMotion on the device: frame t's marker targets
// replace the task targets (a missing marker: target 0, weight 0), `warmup_iters` iterations on the first frame and
// `iters_per_frame` on every later one, warm-started; nothing returns to the host between frames.
__global__ void ik_seq_frame_kernel(const float * __restrict__ tpos_t, const uint8_t * __restrict__ valid_t, float * __restrict__ tpos,
                                    float * __restrict__ posw, int64_t nk, const float * __restrict__ theta, float * __restrict__ theta_prev_out,
                                    int64_t ntheta, int64_t shared_K)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if(tpos_t && i < nk)
  {
    const int64_t j = shared_K > 0 ? i % shared_K : i; // (shared_K: the targets are one capture's [K] for every chain)
    const bool v = valid_t[j] != 0;
    posw[i] = v ? 1.0f : 0.0f;
    for(int x = 0; x < 3; x++) tpos[i * 3 + x] = v ? tpos_t[j * 3 + x] : 0.0f;
  }
  if(theta_prev_out && i < ntheta) theta_prev_out[i] = theta[i];
}

// Development hook (not part of include/smplpp_hip.h): host microseconds the last smplpp_ik_solve_sequence / smplpp_ik_iterate spent enqueueing.
extern "C" int smplpp_debug_ik_enqueue_us(smplpp_ik * s, double * out)
{
  if(!s || !out) return fail(SMPLPP_ERR_INVALID, "smplpp_debug_ik_enqueue_us: bad argument");
  *out = s->last_enqueue_us;
  return SMPLPP_OK;
}

static int ik_solve_sequence_impl(smplpp_ik * s, int64_t T, const float * target_pos, const uint8_t * valid, bool shared, int warmup_iters,
                                  int iters_per_frame, int enable_qp, int64_t min_valid, float * theta_out, int space, void * stream)
{
  if(!s || T <= 0 || !target_pos || !valid || !theta_out || warmup_iters < 0 || iters_per_frame < 0)
    return fail(SMPLPP_ERR_INVALID, "smplpp_ik_solve_sequence: bad argument");
  int rc = check_space(space, "smplpp_ik_solve_sequence");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t nk = s->n * s->K, ntheta = s->n * s->theta_dim;
  const int64_t tk = shared ? s->K : nk; // targets per frame of the sequence as the caller holds them
  In<float> tp;
  In<uint8_t> vl;
  Out<float> th;
  HIP_TRY(tp.init(target_pos, (size_t)(T * tk * 3), space, st));
  HIP_TRY(vl.init(valid, (size_t)(T * tk), space, st));
  HIP_TRY(th.init(theta_out, (size_t)(T * ntheta), space));
  HIP_TRY(hipMemsetAsync(s->sticky, 0, sizeof(int) * s->n, st));
  const int64_t cnt = nk > ntheta ? nk : ntheta;
  const dim3 grid((unsigned)((cnt + 255) / 256));
  // frame 0's targets go in here; every later switch and every frame's record ride on the iterations themselves (SeqHook): no
  // kernel of its own between one frame's solve and the next frame's pose step
  ik_seq_frame_kernel<<<grid, 256, 0, st>>>(tp.d, vl.d, s->ta.tpos, s->ta.posw, nk, nullptr, nullptr, 0, shared ? s->K : 0);
  HIP_TRY(hipGetLastError());
  const auto enq_t0 = std::chrono::steady_clock::now();
  for(int64_t t = 0; t < T; t++)
  {
    const int iters = t == 0 ? warmup_iters : iters_per_frame;
    SeqHook hook;
    hook.theta_record = th.d + t * ntheta;
    if(t + 1 < T)
    {
      hook.next_tpos = tp.d + (t + 1) * tk * 3;
      hook.next_valid = vl.d + (t + 1) * tk;
      hook.shared = shared ? 1 : 0;
    }
    if(iters > 0)
      rc = ik_iterate_enqueue(s, iters, enable_qp, -1, min_valid, st, &hook, /*more_follows=*/t + 1 < T && iters_per_frame > 0);
    else // (no iteration to carry the hook)
    {
      ik_seq_frame_kernel<<<grid, 256, 0, st>>>(hook.next_tpos, hook.next_valid, s->ta.tpos, s->ta.posw, nk, s->theta, hook.theta_record, ntheta,
                                                shared ? s->K : 0);
      HIP_TRY(hipGetLastError());
    }
    if(rc) break;
  }
  const int jrc = ik_join(s, st);
  // (development figure, smplpp_debug_ik_enqueue_us: what the HOST spent handing the T frames' launches to the two streams — when
  // it approaches the frames' time on the GPU, the chains wait for the host)
  s->last_enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - enq_t0).count();
  if(rc) return rc;
  if(jrc) return jrc;
  HIP_TRY(th.finish(st));
  if(space == SMPLPP_HOST)
  {
    HIP_TRY(hipStreamSynchronize(st));
    if((rc = ik_check_valence(s))) return rc;
    if((rc = ik_check_status(s, s->sticky))) return rc;
  }
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_solve_sequence(smplpp_ik * s, int64_t T, const float * target_pos, const uint8_t * valid, int warmup_iters,
                                        int iters_per_frame, int enable_qp, int64_t min_valid, float * theta_out, int space, void * stream)
{
  return ik_solve_sequence_impl(s, T, target_pos, valid, false, warmup_iters, iters_per_frame, enable_qp, min_valid, theta_out, space, stream);
}

// The same loop when every chain fits the SAME capture (the multi-restart fit: BASELINE configs[3], 64 restarts x one sequence):
// target_pos [T,K,3] and valid [T,K] once, handed to all n chains by the frame switch on the device — the caller neither builds nor
// uploads n copies (100 MB for 64 restarts of sample_walk.c3d; 83 ms of host work in front of 0.39 s of GPU work).
extern "C" int smplpp_ik_solve_sequence_shared(smplpp_ik * s, int64_t T, const float * target_pos, const uint8_t * valid, int warmup_iters,
                                               int iters_per_frame, int enable_qp, int64_t min_valid, float * theta_out, int space,
                                               void * stream)
{
  return ik_solve_sequence_impl(s, T, target_pos, valid, true, warmup_iters, iters_per_frame, enable_qp, min_valid, theta_out, space, stream);
}

// Per-frame outcome of the solves so far: flags[f] bit 0 = the last solve of frame f failed ("LLT has numerical issue!",
// node/node.cpp:934-937: the update of that frame was skipped), bit 1 = some solve since the last set_config /
// solve_sequence start failed, bit 2 = an evaluation since the tasks were last set (smplpp_ik_set_tasks clears it; so do set_config
// and solve_sequence) met a task with a normal term on a vertex of more than 12 adjacent faces: its Jacobian rows are not supported,
// and the solve skips that frame's update (bit 0 then reads 1 as for any skipped update).  SMPLPP_HOST calls of iterate / solve_sequence report the same condition as an error; a
// SMPLPP_DEVICE (enqueue-only) caller reads it here once its stream has reached the point of interest.
extern "C" int smplpp_ik_get_status(smplpp_ik * s, int32_t * flags, int space, void * stream)
{
  if(!s || !flags) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_get_status: bad argument");
  int rc = check_space(space, "smplpp_ik_get_status");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::vector<int> a((size_t)s->n), b((size_t)s->n);
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipMemcpy(a.data(), s->status, sizeof(int) * s->n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(b.data(), s->sticky, sizeof(int) * s->n, hipMemcpyDeviceToHost));
  // bit 3: a forward pass INSIDE this solver's loops met an operand outside the fp16x2 form's range since the last set_config
  // (one word per solver — which frame is not recorded, so every frame of the batch carries it; such a frame's vertices are not
  // finite and its solve then fails on its own)
  int internal = 0;
  if(s->range_word && s->m->form_ik == 'h') HIP_TRY(hipMemcpy(&internal, s->range_word, sizeof(int), hipMemcpyDeviceToHost));
  std::vector<int32_t> h((size_t)s->n);
  for(int64_t f = 0; f < s->n; f++)
    h[(size_t)f] = (a[(size_t)f] == 1 ? 1 : 0) | ((b[(size_t)f] & 1) ? 2 : 0) | (b[(size_t)f] & 4) | ((internal & 1) ? 8 : 0);
  if(space == SMPLPP_HOST)
    memcpy(flags, h.data(), sizeof(int32_t) * (size_t)s->n);
  else
    HIP_TRY(hipMemcpy(flags, h.data(), sizeof(int32_t) * (size_t)s->n, hipMemcpyHostToDevice));
  return SMPLPP_OK;
}

extern "C" int smplpp_ik_get_vertices(smplpp_ik * s, float * verts, int space, void * stream)
{
  if(!s || !verts) return fail(SMPLPP_ERR_INVALID, "smplpp_ik_get_vertices: bad argument");
  if(!s->have_eval) return fail(SMPLPP_ERR_STATE, "Failed to get vertices of new pose!");
  int rc = check_space(space, "smplpp_ik_get_vertices");
  if(rc) return rc;
  HIP_TRY(hipSetDevice(s->m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipMemcpyKind kind = space == SMPLPP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  HIP_TRY(hipMemcpyAsync(verts, s->verts, sizeof(float) * s->n * s->m->V * 3, kind, st));
  if(space == SMPLPP_HOST) HIP_TRY(hipStreamSynchronize(st));
  return SMPLPP_OK;
}
