// smplpp::IkTask (reference include/smplpp/IkTask.h:20-84) and the batched IK loop of node/node.cpp:645-1002 over the
// C ABI.  Same public field names and defaults as the reference class.
#ifndef SMPLPP_SHIM_IK_TASK_H
#define SMPLPP_SHIM_IK_TASK_H

#include <cmath>

#include "SMPL.h"
#include "VPoser.h"

namespace smplpp
{
class IkTask
{
public:
  // include/smplpp/IkTask.h:20-34 of the reference: (smpl, faceIdx) and (smpl, faceIdx, targetPos, targetNormal) with
  // tensors of three elements
  IkTask(const std::shared_ptr<smplpp::SMPL> & smpl, int64_t faceIdx) : smpl_(smpl), faceIdx_(faceIdx)
  {
    targetNormal_.index_put_({2}, 1.0); // src/IkTask.cpp:11-15: target position 0, target normal +Z
  }
  IkTask(const std::shared_ptr<smplpp::SMPL> & smpl, int64_t faceIdx, Tensor targetPos, Tensor targetNormal)
  : smpl_(smpl), faceIdx_(faceIdx), targetPos_(std::move(targetPos)), targetNormal_(std::move(targetNormal))
  {
    if(targetPos_.numel() != 3 || targetNormal_.numel() != 3) throw Exception("IkTask", "targetPos / targetNormal must hold three elements");
  }

  // ---- the four methods of the reference class (include/smplpp/IkTask.h:33-49, src/IkTask.cpp:33-86), on the vertices of
  // batch 0 of the last SMPL::launch.  In the reference they are also the autograd seam; here they are plain values (the
  // Jacobian comes from IkSolver::eval).  fp32 like the reference's tensors.
  // Tangent vectors of the focused face (src/IkTask.cpp:33-48)
  void calcTangents()
  {
    float v[3][3];
    faceVertices(v);
    float t1[3], b[3], nrm[3], t2[3];
    for(int x = 0; x < 3; x++)
    {
      t1[x] = v[1][x] - v[0][x];
      b[x] = v[2][x] - v[0][x];
    }
    cross(t1, b, nrm);
    cross(nrm, t1, t2);
    normalize(t1);
    normalize(t2);
    for(int x = 0; x < 3; x++)
    {
      tangents_.data[(size_t)x * 2 + 0] = t1[x];
      tangents_.data[(size_t)x * 2 + 1] = t2[x];
    }
  }
  // Vertex weights such that actualPos + tangents . phi is the weighted sum of the face vertices (src/IkTask.cpp:50-58,
  // calcTriangleVertexWeights: toolbox/GeometryUtils.h:42-52)
  void calcVertexWeights(const Tensor & actualPos)
  {
    if(actualPos.numel() != 3) throw Exception("IkTask", "calcVertexWeights: a point of three elements");
    float v[3][3], pos[3], w[3];
    faceVertices(v);
    for(int x = 0; x < 3; x++)
      pos[x] = (float)actualPos.at(x) + tangents_.data[(size_t)x * 2] * phi_.data[0] + tangents_.data[(size_t)x * 2 + 1] * phi_.data[1];
    for(int i = 0; i < 3; i++)
    {
      float a[3], b[3], c[3];
      for(int x = 0; x < 3; x++)
      {
        a[x] = v[(i + 1) % 3][x] - pos[x];
        b[x] = v[(i + 2) % 3][x] - pos[x];
      }
      cross(a, b, c);
      w[i] = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    }
    const float sum = w[0] + w[1] + w[2];
    for(int i = 0; i < 3; i++) vertexWeights_.data[(size_t)i] = w[i] / sum;
  }
  // Position of the task point (src/IkTask.cpp:60-72)
  Tensor calcActualPos() const
  {
    float v[3][3];
    faceVertices(v);
    Tensor p({3});
    for(int x = 0; x < 3; x++)
      for(int i = 0; i < 3; i++) p.data[(size_t)x] += v[i][x] * vertexWeights_.data[(size_t)i];
    if(normalOffset_ > 0.0)
    {
      const Tensor n = calcActualNormal();
      for(int x = 0; x < 3; x++) p.data[(size_t)x] += (float)normalOffset_ * n.data[(size_t)x];
    }
    return p;
  }
  // Unit normal of the task point: the weighted vertex normals of the face (src/IkTask.cpp:74-86)
  Tensor calcActualNormal() const
  {
    const Tensor fv = smpl_->getFaceIndexRaw(faceIdx_).to(kCPU) - 1;
    float n[3] = {0.f, 0.f, 0.f};
    for(int i = 0; i < 3; i++)
    {
      const Tensor vn = smpl_->calcVertexNormal(fv.idata[(size_t)i]);
      for(int x = 0; x < 3; x++) n[x] += vertexWeights_.data[(size_t)i] * vn.data[(size_t)x];
    }
    normalize(n);
    Tensor r({3});
    for(int x = 0; x < 3; x++) r.data[(size_t)x] = n[x];
    return r;
  }

  // public fields of the reference class, same names, types and defaults (include/smplpp/IkTask.h:54-84)
  std::shared_ptr<smplpp::SMPL> smpl_;
  int64_t faceIdx_;
  double posTaskWeight_ = 1.0;
  double normalTaskWeight_ = 1.0;
  double phiLimit_ = 0.04;
  double normalOffset_ = 0.0;
  Tensor targetPos_ = Tensor({3});
  Tensor targetNormal_ = Tensor({3});
  Tensor vertexWeights_ = Tensor({3}, 1.0f / 3.0f);
  Tensor tangents_ = Tensor({3, 2});
  Tensor phi_ = Tensor({2});

private:
  void faceVertices(float (&v)[3][3]) const
  {
    const Tensor fv = smpl_->getFaceIndexRaw(faceIdx_).to(kCPU) - 1;        // src/IkTask.cpp:35
    const Tensor t = smpl_->getVertexRaw(fv.to(kInt64)).to(kCPU).clone().detach(); // :37
    for(int i = 0; i < 3; i++)
      for(int x = 0; x < 3; x++) v[i][x] = t.data[(size_t)i * 3 + x];
  }
  static void cross(const float * a, const float * b, float * c)
  {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
  }
  static void normalize(float * a) // torch::nn::functional::normalize: x / max(|x|, 1e-12)
  {
    const float n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const float d = n > 1e-12f ? n : 1e-12f;
    for(int x = 0; x < 3; x++) a[x] /= d;
  }
};

// g_ikTaskList (node/node.cpp:47): std::map order fixes the rows of e/J and the phi column blocks (:798).
using IkTaskList = std::map<std::string, IkTask>;

// The loop body of node/node.cpp:704-1001 for n frames that share one task list layout.
class IkSolver
{
public:
  // vposer (nullable): the 44-d configuration layout of node/node.cpp:761-772 with the decoder inside the loop
  IkSolver(const std::shared_ptr<SMPL> & smpl, int64_t n, int64_t K, const std::shared_ptr<VPoserDecoder> & vposer = nullptr)
  : smpl_(smpl), vposer_(vposer), n_(n), K_(K)
  {
    check(smplpp_ik_create(smpl->handle(), n, K, vposer ? vposer->handle() : nullptr, &s_), "node");
  }
  int64_t thetaDim() const { return vposer_ ? (int64_t)(LATENT_DIM + 12) : (int64_t)SMPLPP_THETA_DIM; }
  ~IkSolver() { smplpp_ik_destroy(s_); }
  IkSolver(const IkSolver &) = delete;
  IkSolver & operator=(const IkSolver &) = delete;

  // every frame gets the same task list
  void setTaskList(const IkTaskList & tasks)
  {
    if((int64_t)tasks.size() != K_) throw Exception("node", "task count mismatch");
    std::vector<int64_t> face((size_t)(n_ * K_));
    std::vector<float> vw((size_t)(n_ * K_ * 3)), tp(vw.size()), tn(vw.size());
    std::vector<double> pw((size_t)(n_ * K_)), nw(pw.size()), pl(pw.size()), no(pw.size());
    for(int64_t f = 0; f < n_; f++)
    {
      int64_t k = 0;
      for(const auto & kv : tasks)
      {
        const IkTask & t = kv.second;
        const size_t i = (size_t)(f * K_ + k);
        face[i] = t.faceIdx_;
        pw[i] = t.posTaskWeight_;
        nw[i] = t.normalTaskWeight_;
        pl[i] = t.phiLimit_;
        no[i] = t.normalOffset_;
        for(int x = 0; x < 3; x++)
        {
          vw[i * 3 + x] = (float)t.vertexWeights_.at(x);
          tp[i * 3 + x] = (float)t.targetPos_.at(x);
          tn[i * 3 + x] = (float)t.targetNormal_.at(x);
        }
        k++;
      }
    }
    check(smplpp_ik_set_tasks(s_, face.data(), vw.data(), tp.data(), tn.data(), pw.data(), nw.data(), pl.data(), no.data(), SMPLPP_HOST),
          "node");
  }
  void setConfig(const Tensor & beta /*[n,10]*/, const Tensor & theta /*[n,25,3], or [n,44] with a VPoser*/)
  {
    if(theta.numel() != n_ * thetaDim()) throw Exception("node", "setConfig: theta must hold n x thetaDim values");
    check(smplpp_ik_set_config(s_, beta.ptr(), theta.ptr(), SMPLPP_HOST), "node");
  }
  void getConfig(Tensor & beta, Tensor & theta)
  {
    beta = Tensor({n_, SHAPE_BASIS_DIM});
    theta = vposer_ ? Tensor({n_, thetaDim()}) : Tensor({n_, JOINT_NUM + 1, 3});
    check(smplpp_ik_get_config(s_, beta.ptr(), theta.ptr(), SMPLPP_HOST), "node");
  }
  // node.cpp:798-877 in one call: e [n,4K], J [n,4K,D] (row-major, fp64)
  void eval(bool optimizeBeta, std::vector<double> & e, std::vector<double> & J)
  {
    const int64_t D = thetaDim() + 2 * K_ + (optimizeBeta ? SHAPE_BASIS_DIM : 0);
    e.resize((size_t)(n_ * 4 * K_));
    J.resize((size_t)(n_ * 4 * K_ * D));
    check(smplpp_ik_eval(s_, optimizeBeta ? 1 : 0, e.data(), J.data(), SMPLPP_HOST, nullptr), "node");
  }
  // node.cpp:704-1001 x iters; returns |e|^2 per frame of the last evaluation
  std::vector<double> iterate(int iters, bool enableQp = false, int optimizeBetaFrom = -1, int64_t minValid = 0)
  {
    std::vector<double> e2((size_t)n_);
    check(smplpp_ik_iterate(s_, iters, enableQp ? 1 : 0, optimizeBetaFrom, minValid, e2.data(), SMPLPP_HOST, nullptr), "node");
    return e2;
  }
  // the frame loop of solveMocapMotion (node.cpp:1369-1407, targets per frame :681-700) without a host round trip per
  // frame: targetPos [T,n,K,3], valid [T,n,K]; returns g_theta after every frame [T,n,thetaDim]
  std::vector<float> solveSequence(int64_t T, const std::vector<float> & targetPos, const std::vector<uint8_t> & valid,
                                   int warmupIters = 32, int itersPerFrame = 1, bool enableQp = true, int64_t minValid = 0)
  {
    if((int64_t)targetPos.size() != T * n_ * K_ * 3 || (int64_t)valid.size() != T * n_ * K_)
      throw Exception("node", "solveSequence: targetPos must be [T,n,K,3] and valid [T,n,K]");
    std::vector<float> theta((size_t)(T * n_ * thetaDim()));
    check(smplpp_ik_solve_sequence(s_, T, targetPos.data(), valid.data(), warmupIters, itersPerFrame, enableQp ? 1 : 0, minValid,
                                   theta.data(), SMPLPP_HOST, nullptr),
          "node");
    return theta;
  }

  // the same loop for ONE capture shared by all n chains (restarts): targetPos [T,K,3], valid [T,K] (smplpp_ik_solve_sequence_shared)
  std::vector<float> solveSequenceShared(int64_t T, const std::vector<float> & targetPos, const std::vector<uint8_t> & valid,
                                         int warmupIters = 32, int itersPerFrame = 1, bool enableQp = true, int64_t minValid = 0)
  {
    if((int64_t)targetPos.size() != T * K_ * 3 || (int64_t)valid.size() != T * K_)
      throw Exception("node", "solveSequenceShared: targetPos must be [T,K,3] and valid [T,K]");
    std::vector<float> theta((size_t)(T * n_ * thetaDim()));
    check(smplpp_ik_solve_sequence_shared(s_, T, targetPos.data(), valid.data(), warmupIters, itersPerFrame, enableQp ? 1 : 0, minValid,
                                          theta.data(), SMPLPP_HOST, nullptr),
          "node");
    return theta;
  }

  // per-task state after the last evaluation / re-projection (what the reference reads back through IkTask fields and
  // calcActualPos / calcActualNormal, node/node.cpp:803-814, 958, 997-998): faceIdx [n,K], vertexWeights [n,K,3],
  // tangents [n,K,3,2], actualPos [n,K,3], actualNormal [n,K,3]
  void getTasks(std::vector<int64_t> & faceIdx, std::vector<float> & vertexWeights, std::vector<float> & tangents,
                std::vector<float> & actualPos, std::vector<float> & actualNormal)
  {
    const size_t nk = (size_t)(n_ * K_);
    faceIdx.resize(nk);
    vertexWeights.resize(nk * 3);
    tangents.resize(nk * 6);
    actualPos.resize(nk * 3);
    actualNormal.resize(nk * 3);
    check(smplpp_ik_get_tasks(s_, faceIdx.data(), vertexWeights.data(), tangents.data(), actualPos.data(), actualNormal.data(), SMPLPP_HOST),
          "node");
  }

private:
  std::shared_ptr<SMPL> smpl_;
  std::shared_ptr<VPoserDecoder> vposer_;
  int64_t n_, K_;
  smplpp_ik * s_ = nullptr;
};
} // namespace smplpp
#endif
