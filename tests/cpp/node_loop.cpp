// The reference's one caller, re-hosted on the shim without ROS: the loop body of
// /root/reference/node/node.cpp:645-1001 and the frame advance of :1369-1407, written against smplpp::SMPL / smplpp::IkTask /
// smplpp::IkSolver exactly as INTEGRATION.md section 3 prescribes — same globals (g_smpl, g_ikTaskList, g_theta, g_beta), same
// std::map task order, same per-iteration target switch, same skip rule — with `namespace torch = smplpp::torchlike`, so that the
// tensor expressions of the reference's call sites stand as they are written there.  What changes is the autograd seam alone:
// node.cpp:798-943 (per-row backward() + Eigen normal equations + LLT / QLD) is ONE call, smplpp::IkSolver::iterate.
// Driven by tests/test_cpp_shim.py, which checks what this program writes against the reference-generated golden trajectory
// (tests/golden/ik_traj50.npz) and against the CPU oracle, never against the Python mirror of the same library.
//   usage: node_loop <model.json> <in.bin> <out.bin>
//   in.bin : int64 mode (0 = direct IK from given states, 1 = capture window), then the arrays documented at `Input` below
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include <smplpp/IkTask.h>
#include <smplpp/SMPL.h>

namespace torch = smplpp::torchlike;
namespace at = smplpp::torchlike::at;

// ---- the node's globals (node/node.cpp:40-52)
static std::shared_ptr<smplpp::SMPL> g_smpl;
static std::map<std::string, smplpp::IkTask> g_ikTaskList;
static torch::Tensor g_theta = torch::zeros({smplpp::JOINT_NUM + 1, 3});
static torch::Tensor g_beta = torch::zeros({smplpp::SHAPE_BASIS_DIM});

struct Reader
{
  std::ifstream f;
  explicit Reader(const char * p) : f(p, std::ios::binary) {}
  int64_t i64()
  {
    int64_t v = 0;
    f.read(reinterpret_cast<char *>(&v), 8);
    return v;
  }
  std::vector<float> f32(int64_t n)
  {
    std::vector<float> v((size_t)n);
    f.read(reinterpret_cast<char *>(v.data()), n * 4);
    return v;
  }
  std::vector<int64_t> i64s(int64_t n)
  {
    std::vector<int64_t> v((size_t)n);
    f.read(reinterpret_cast<char *>(v.data()), n * 8);
    return v;
  }
  std::string str()
  {
    const int64_t n = i64();
    std::string s((size_t)n, ' ');
    f.read(&s[0], n);
    return s;
  }
};
struct Writer
{
  std::ofstream f;
  explicit Writer(const char * p) : f(p, std::ios::binary) {}
  void f32(const float * p, int64_t n) { f.write(reinterpret_cast<const char *>(p), n * 4); }
  void f32(const torch::Tensor & t)
  {
    const std::vector<float> v = t.toVector<float>();
    f32(v.data(), (int64_t)v.size());
  }
  void f64(double v) { f.write(reinterpret_cast<const char *>(&v), 8); }
  void i64(int64_t v) { f.write(reinterpret_cast<const char *>(&v), 8); }
};

// What one pass of the loop body leaves behind for the test: the state the solve started from and the state it produced
static void dumpState(Writer & w)
{
  w.f32(g_theta);
  for(const auto & ikTaskKV : g_ikTaskList) w.i64(ikTaskKV.second.faceIdx_);
  for(const auto & ikTaskKV : g_ikTaskList) w.f32(ikTaskKV.second.vertexWeights_);
}

// node/node.cpp:645-1001 for one value of ikIter.  `points` / `valid`: the capture frame the iteration fits (solveMocap), or
// null.  Returns |e|^2 of the evaluation (what the damping term adds to the diagonal, :893), -1 when the solve was skipped (:785).
static double loopBody(int64_t ikIter, smplpp::IkSolver & solver, bool solveMocapBody, bool solveMocapMotion, bool enableIk, bool enableQp,
                       const float * points, const uint8_t * valid)
{
  const bool solveMocap = solveMocapBody || solveMocapMotion;
  bool optimizeBeta = false; // :652-656
  if(solveMocapBody)
  {
    optimizeBeta = (ikIter >= 25);
  }

  // Update IK target from mocap (:664-700)
  int32_t validMocapMarkerNum = 0;
  if(solveMocap)
  {
    int32_t mocapMarkerIdx = 0;
    for(auto & ikTaskKV : g_ikTaskList)
    {
      auto & ikTask = ikTaskKV.second;
      if(!valid[mocapMarkerIdx]) // point.isEmpty()
      {
        if(solveMocapBody)
        {
          throw smplpp::smpl_error("node", "All mocap markers must be found to solve mocap body: " + ikTaskKV.first + " not found.");
        }
        ikTask.posTaskWeight_ = 0.0;
        ikTask.targetPos_.zero_();
      }
      else
      {
        validMocapMarkerNum++;
        ikTask.posTaskWeight_ = 1.0;
        ikTask.targetPos_.index_put_({0}, points[mocapMarkerIdx * 3 + 0]);
        ikTask.targetPos_.index_put_({1}, points[mocapMarkerIdx * 3 + 1]);
        ikTask.targetPos_.index_put_({2}, points[mocapMarkerIdx * 3 + 2]);
      }

      if(solveMocapBody)
      {
        ikTask.phiLimit_ = ikIter < 25 ? 0.0 : 0.04;
      }
      else if(solveMocapMotion)
      {
        ikTask.phiLimit_ = 0.0;
      }
      mocapMarkerIdx++;
    }
  }

  // Setup gradient (:702-743): nothing to set up — there is no autograd behind smplpp::Tensor

  // Forward SMPL model (:745-781)
  {
    torch::Tensor theta;
    theta = g_theta;
    g_smpl->launch(g_beta.view({1, -1}), theta.view({1, theta.size(0), theta.size(1)}));
  }

  // Solve IK (:783-1001)
  double eSquaredNorm = -1.0;
  if(enableIk && !(solveMocapMotion && validMocapMarkerNum < (int32_t)(g_ikTaskList.size() / 2)))
  {
    // :798-943 — residual, Jacobian (the reference: one backward() per row), normal equations, damping, LLT / box QP — and the
    // configuration update of :945-968 are the engine's: the task list and the configuration go in, one iteration runs
    solver.setTaskList(g_ikTaskList);
    solver.setConfig(g_beta.view({1, -1}), g_theta.view({1, smplpp::JOINT_NUM + 1, 3}));
    eSquaredNorm = solver.iterate(1, enableQp, optimizeBeta ? 0 : -1, 0)[0];

    // Update config (:945-968)
    torch::Tensor beta, theta;
    solver.getConfig(beta, theta);
    g_theta = theta.index({0});
    if(optimizeBeta)
    {
      g_beta = beta.index({0});
    }

    // Project point onto mesh + update face and vertex weights (:970-1001): igl::point_mesh_squared_distance ran on the device
    std::vector<int64_t> faceIdx;
    std::vector<float> vertexWeights, tangents, actualPos, actualNormal;
    solver.getTasks(faceIdx, vertexWeights, tangents, actualPos, actualNormal);
    int32_t ikTaskIdx = 0;
    for(auto & ikTaskKV : g_ikTaskList)
    {
      auto & ikTask = ikTaskKV.second;
      ikTask.faceIdx_ = faceIdx[(size_t)ikTaskIdx];
      for(int i = 0; i < 3; i++) ikTask.vertexWeights_.index_put_({i}, vertexWeights[(size_t)(ikTaskIdx * 3 + i)]);
      for(int i = 0; i < 6; i++) ikTask.tangents_.data[(size_t)i] = tangents[(size_t)(ikTaskIdx * 6 + i)];
      ikTaskIdx++;
    }
  }
  return eSquaredNorm;
}

int main(int argc, char ** argv)
{
  if(argc < 4)
  {
    std::printf("usage: node_loop <model.json> <in.bin> <out.bin>\n");
    return 64;
  }
  try
  {
    // ---- model setup (node/node.cpp:360-372, :412-415)
    std::unique_ptr<torch::Device> device;
    {
      std::string deviceType = "CUDA";
      if(deviceType == "CPU")
      {
        device = std::make_unique<torch::Device>(torch::kCPU);
      }
      else if(deviceType == "CUDA")
      {
        device = std::make_unique<torch::Device>(torch::kCUDA);
      }
      device->set_index(0);
    }
    g_smpl = std::make_shared<smplpp::SMPL>();
    g_smpl->setDevice(*device);
    g_smpl->setModelPath(argv[1]);
    g_smpl->init();

    Reader in(argv[2]);
    Writer out(argv[3]);
    const int64_t mode = in.i64();
    // Input (both modes): int64 K, then per task: name, int64 faceIdx, float targetPos[3], float targetNormal[3]
    const int64_t K = in.i64();
    for(int64_t k = 0; k < K; k++)
    {
      const std::string name = in.str();
      const int64_t faceIdx = in.i64();
      const std::vector<float> tp = in.f32(3), tn = in.f32(3);
      // (node/node.cpp:455-550: emplace by name; the std::map's order fixes the rows of e / J)
      g_ikTaskList.emplace(name, smplpp::IkTask(g_smpl, faceIdx, torch::tensor({tp[0], tp[1], tp[2]}), torch::tensor({tn[0], tn[1], tn[2]})));
    }
    smplpp::IkSolver solver(g_smpl, 1, K);

    if(mode == 0)
    {
      // ---- direct IK: int64 S states, each theta[75], faces[K], weights[3K] (in std::map order).  For every state: set the
      // node's globals to it, run ONE pass of the loop body, write the state it produced; then a free run from state 0.
      const int64_t S = in.i64();
      for(auto & ikTaskKV : g_ikTaskList) ikTaskKV.second.phiLimit_ = 0.0; // node/node.cpp:567
      std::vector<std::vector<float>> th, wt;
      std::vector<std::vector<int64_t>> fc;
      for(int64_t s = 0; s < S; s++)
      {
        th.push_back(in.f32(75));
        fc.push_back(in.i64s(K));
        wt.push_back(in.f32(3 * K));
      }
      auto setState = [&](int64_t s) {
        for(int i = 0; i < 75; i++) g_theta.data[(size_t)i] = th[(size_t)s][(size_t)i];
        int64_t k = 0;
        for(auto & ikTaskKV : g_ikTaskList)
        {
          ikTaskKV.second.faceIdx_ = fc[(size_t)s][(size_t)k];
          for(int i = 0; i < 3; i++) ikTaskKV.second.vertexWeights_.data[(size_t)i] = wt[(size_t)s][(size_t)(k * 3 + i)];
          k++;
        }
      };
      out.i64(S);
      for(int64_t s = 0; s < S; s++)
      {
        setState(s);
        const double e2 = loopBody(s, solver, false, false, true, false, nullptr, nullptr);
        dumpState(out);
        out.f64(e2);
      }
      setState(0);
      for(int64_t it = 0; it < S; it++)
      {
        const double e2 = loopBody(it, solver, false, false, true, false, nullptr, nullptr);
        dumpState(out);
        out.f64(e2);
      }
      // ---- the getters the node's other call sites use (node/node.cpp:121-123, 183-186, 214-220, 976-978), on the last pose
      torch::Tensor vertexTensor = g_smpl->getVertex().index({0}).to(torch::kCPU);       // :121 / :976 / :1028 / :1114
      torch::Tensor faceIdxTensor = g_smpl->getFaceIndex().to(torch::kCPU) - 1;           // :123 / :978 / :1030 / :1116
      const int64_t faceIdx = g_ikTaskList.begin()->second.faceIdx_;
      torch::Tensor faceVertexIdxs = g_smpl->getFaceIndexRaw(faceIdx).to(torch::kCPU) - 1; // :183
      torch::Tensor faceVertices = g_smpl->getVertexRaw(faceVertexIdxs.to(torch::kInt64)).to(torch::kCPU).clone().detach(); // :186
      out.i64(vertexTensor.size(0));
      out.i64(faceIdxTensor.size(0));
      out.i64(faceIdx);
      for(int32_t i = 0; i < 3; i++)
        for(int32_t j = 0; j < 3; j++) out.f64((double)faceVertices.index({i, j}).item<float>()); // (:193-195 reads them like this)
      const int32_t * facePtr = faceIdxTensor.data_ptr<int32_t>(); // (toEigenMatrix<int>(faceIdxTensor), :124)
      for(int j = 0; j < 3; j++) out.i64((int64_t)facePtr[faceIdx * 3 + j]);
      const int64_t vertexIdx = faceVertexIdxs.index({0}).item<int64_t>();
      int64_t adjacentSum = 0, adjacentNum = 0;
      for(const auto & adjacentFaceKV : g_smpl->getAdjacentFaces(vertexIdx)) // :214
      {
        adjacentSum += adjacentFaceKV.first;
        adjacentNum++;
      }
      out.i64(vertexIdx);
      out.i64(adjacentNum);
      out.i64(adjacentSum);
      // IkTask's own methods on the same pose (src/IkTask.cpp:33-86; the node calls them at :803-814)
      smplpp::IkTask & ikTask = g_ikTaskList.begin()->second;
      ikTask.calcTangents();
      out.f32(ikTask.tangents_);
      out.f32(ikTask.calcActualPos().to(torch::kCPU).clone().detach());
      out.f32(ikTask.calcActualNormal());
      torch::Tensor normalError = ikTask.normalTaskWeight_ * (at::dot(ikTask.calcActualNormal(), ikTask.targetNormal_).to(torch::kCPU) + 1.0); // :811
      out.f64((double)normalError.item<float>());
      // SMPL::out (src/SMPL.cpp:757-790) into the path of setVertPath, through a COPY of the model object (src/SMPL.cpp:160)
      smplpp::SMPL smplCopy(*g_smpl);
      smplCopy.setVertPath(std::string(argv[3]) + ".obj");
      smplCopy.out(0);
    }
    else
    {
      // ---- capture window (solveMocapMotion, node/node.cpp:553-567): int64 T frames, float points[T,K,3], int64 valid[T,K] (std::map
      // order), int64 warm-up iterations (the reference: frames advance once ikIter > 30), float theta0[75]
      const int64_t T = in.i64();
      const std::vector<float> points = in.f32(T * K * 3);
      const std::vector<int64_t> valid64 = in.i64s(T * K);
      const int64_t warm = in.i64();
      const std::vector<float> th0 = in.f32(75);
      std::vector<uint8_t> valid(valid64.begin(), valid64.end());
      g_theta = torch::from_blob(th0.data(), {75}).clone().view({smplpp::JOINT_NUM + 1, 3}); // (toTorchTensor<float>(..., true), :390-391)
      for(auto & ikTaskKV : g_ikTaskList) // :553-567
      {
        ikTaskKV.second.normalTaskWeight_ = 0.0;
        ikTaskKV.second.normalOffset_ = 0.015;
        ikTaskKV.second.phiLimit_ = 0.0;
      }
      int64_t mocapFrameIdx = 0;
      const int64_t mocapFrameInterval = 1;
      std::vector<float> motion; // motionMsg.data_list (:1389-1398): frame index + theta per stored instant
      int64_t iterations = 0;
      for(int64_t ikIter = 0;; ikIter++)
      {
        out.i64(mocapFrameIdx);
        dumpState(out);
        const double e2 = loopBody(ikIter, solver, false, true, true, true, points.data() + mocapFrameIdx * K * 3, valid.data() + mocapFrameIdx * K);
        dumpState(out);
        out.f64(e2);
        iterations++;
        // :1369-1407
        if(ikIter > warm - 2)
        {
          motion.push_back((float)mocapFrameIdx);
          for(int i = 0; i < 75; i++) motion.push_back(g_theta.data[(size_t)i]);
          mocapFrameIdx += mocapFrameInterval;
        }
        if(mocapFrameIdx >= T)
        {
          break;
        }
      }
      out.i64(-1);
      out.i64(iterations);
      out.i64((int64_t)motion.size() / 76);
      out.f32(motion.data(), (int64_t)motion.size());
    }
    std::printf("OK\n");
    return 0;
  }
  catch(const smplpp::Exception & ex)
  {
    std::printf("smplpp::Exception: %s\n", ex.what());
    return 1;
  }
}
