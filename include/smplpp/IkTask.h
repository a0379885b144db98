// smplpp::IkTask (reference include/smplpp/IkTask.h:20-84) and the batched IK loop of node/node.cpp:645-1002 over the
// C ABI.  Same public field names and defaults as the reference class.
#ifndef SMPLPP_SHIM_IK_TASK_H
#define SMPLPP_SHIM_IK_TASK_H

#include "SMPL.h"

namespace smplpp
{
class IkTask
{
public:
  IkTask(const std::shared_ptr<smplpp::SMPL> & smpl, int64_t faceIdx) : smpl_(smpl), faceIdx_(faceIdx) {}
  IkTask(const std::shared_ptr<smplpp::SMPL> & smpl, int64_t faceIdx, const std::vector<float> & targetPos,
         const std::vector<float> & targetNormal)
  : smpl_(smpl), faceIdx_(faceIdx), targetPos_(targetPos), targetNormal_(targetNormal)
  {
  }
  std::shared_ptr<smplpp::SMPL> smpl_;
  int64_t faceIdx_;
  double posTaskWeight_ = 1.0;
  double normalTaskWeight_ = 1.0;
  double phiLimit_ = 0.04;
  double normalOffset_ = 0.0;
  std::vector<float> targetPos_{0.f, 0.f, 0.f};
  std::vector<float> targetNormal_{0.f, 0.f, 1.f};
  std::vector<float> vertexWeights_{1.f / 3, 1.f / 3, 1.f / 3};
  std::vector<float> tangents_ = std::vector<float>(6, 0.f); // [3,2]
  std::vector<float> phi_{0.f, 0.f};
};

// g_ikTaskList (node/node.cpp:47): std::map order fixes the rows of e/J and the phi column blocks (:798).
using IkTaskList = std::map<std::string, IkTask>;

// The loop body of node/node.cpp:704-1001 for n frames that share one task list layout.
class IkSolver
{
public:
  IkSolver(const std::shared_ptr<SMPL> & smpl, int64_t n, int64_t K) : smpl_(smpl), n_(n), K_(K)
  {
    check(smplpp_ik_create(smpl->handle(), n, K, nullptr, &s_), "node");
  }
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
          vw[i * 3 + x] = t.vertexWeights_[x];
          tp[i * 3 + x] = t.targetPos_[x];
          tn[i * 3 + x] = t.targetNormal_[x];
        }
        k++;
      }
    }
    check(smplpp_ik_set_tasks(s_, face.data(), vw.data(), tp.data(), tn.data(), pw.data(), nw.data(), pl.data(), no.data(), SMPLPP_HOST),
          "node");
  }
  void setConfig(const Tensor & beta /*[n,10]*/, const Tensor & theta /*[n,25,3]*/)
  {
    check(smplpp_ik_set_config(s_, beta.ptr(), theta.ptr(), SMPLPP_HOST), "node");
  }
  void getConfig(Tensor & beta, Tensor & theta)
  {
    beta = Tensor({n_, SHAPE_BASIS_DIM});
    theta = Tensor({n_, JOINT_NUM + 1, 3});
    check(smplpp_ik_get_config(s_, beta.ptr(), theta.ptr(), SMPLPP_HOST), "node");
  }
  // node.cpp:798-877 in one call: e [n,4K], J [n,4K,D] (row-major, fp64)
  void eval(bool optimizeBeta, std::vector<double> & e, std::vector<double> & J)
  {
    const int64_t D = SMPLPP_THETA_DIM + 2 * K_ + (optimizeBeta ? SHAPE_BASIS_DIM : 0);
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
    std::vector<float> theta((size_t)(T * n_ * SMPLPP_THETA_DIM));
    check(smplpp_ik_solve_sequence(s_, T, targetPos.data(), valid.data(), warmupIters, itersPerFrame, enableQp ? 1 : 0, minValid,
                                   theta.data(), SMPLPP_HOST, nullptr),
          "node");
    return theta;
  }

private:
  std::shared_ptr<SMPL> smpl_;
  int64_t n_, K_;
  smplpp_ik * s_ = nullptr;
};
} // namespace smplpp
#endif
