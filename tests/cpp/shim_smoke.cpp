// Compiles against the header-only C++ shim and libsmplpp_hip.so; driven by tests/test_cpp_shim.py.
// usage: shim_smoke <model.json> [--expect-no-gpu]
#include <cmath>
#include <cstdio>
#include <cstring>

#include <smplpp/IkTask.h>
#include <smplpp/SMPL.h>

int main(int argc, char ** argv)
{
  const bool expect_no_gpu = argc > 2 && !std::strcmp(argv[2], "--expect-no-gpu");
  try
  {
    auto smpl = std::make_shared<smplpp::SMPL>();
    smpl->setDevice(smplpp::Device("CUDA", 0)); // node/node.cpp:360-372
    smpl->setModelPath(argv[1]);
    smpl->init();
    smplpp::Tensor beta({1, 10}), theta({1, 25, 3});
    theta.data[2] = 0.25f; // root translation z
    smpl->launch(beta, theta);
    smplpp::Tensor v = smpl->getVertex();
    smplpp::Tensor r = smpl->getRestShape();
    double worst = 0.0;
    for(int64_t i = 0; i < smpl->vertexNum(); i++)
      for(int x = 0; x < 3; x++)
        worst = std::fmax(worst, std::fabs((double)v.data[i * 3 + x] - (double)r.data[i * 3 + x] - (x == 2 ? 0.25 : 0.0)));
    std::printf("zero pose: max |v - (rest + t)| = %.3g\n", worst);
    if(worst > 1e-6) return 2;
    // one IK iteration with two tasks (std::map order)
    smplpp::IkTaskList tasks;
    tasks.emplace("LeftHand", smplpp::IkTask(smpl, 5));
    tasks.emplace("RightHand", smplpp::IkTask(smpl, 9));
    for(auto & kv : tasks)
    {
      kv.second.phiLimit_ = 0.0; // node.cpp:567
      kv.second.targetPos_ = {0.1f, 0.2f, 0.3f};
    }
    smplpp::IkSolver solver(smpl, 1, 2);
    solver.setTaskList(tasks);
    solver.setConfig(beta, theta);
    std::vector<double> e, J;
    solver.eval(false, e, J);
    auto e2 = solver.iterate(3);
    std::printf("ik: |e|^2 after 3 iterations = %.3g (rows %zu, J %zu)\n", e2[0], e.size(), J.size());
    // the capture-fitting frame loop on the device: 3 frames, the second with a missing marker
    const int64_t T = 3;
    std::vector<float> tp((size_t)(T * 1 * 2 * 3));
    std::vector<uint8_t> valid((size_t)(T * 1 * 2), 1);
    for(int64_t t = 0; t < T; t++)
      for(int k = 0; k < 2; k++)
      {
        tp[(size_t)((t * 2 + k) * 3 + 0)] = 0.1f + 0.01f * (float)t;
        tp[(size_t)((t * 2 + k) * 3 + 1)] = 0.2f;
        tp[(size_t)((t * 2 + k) * 3 + 2)] = 0.3f;
      }
    valid[3] = 0;
    auto seq = solver.solveSequence(T, tp, valid, 4, 1, true, 0);
    bool finite = true;
    for(float x : seq) finite = finite && std::isfinite(x);
    std::printf("sequence: %zu values, finite %d\n", seq.size(), finite ? 1 : 0);
    if(!finite || seq.size() != (size_t)(T * 75)) return 3;
    std::printf("OK\n");
    return 0;
  }
  catch(const smplpp::Exception & ex)
  {
    std::printf("smplpp::Exception: %s\n", ex.what());
    return expect_no_gpu ? 0 : 1;
  }
}
