// Compiles against the header-only C++ shim and libsmplpp_hip.so; driven by tests/test_cpp_shim.py.
// usage: shim_smoke <model.json> [<vposer.json>] [--expect-no-gpu]
// Prints "KEY v0 v1 ..." lines that tests/test_cpp_shim.py compares with the Python mirror's numbers.
#include <cmath>
#include <cstdio>
#include <cstring>

#include <smplpp/IkTask.h>
#include <smplpp/SMPL.h>
#include <smplpp/VPoser.h>

static void dump(const char * key, const std::vector<float> & v)
{
  std::printf("%s", key);
  for(float x : v) std::printf(" %.9g", (double)x);
  std::printf("\n");
}
static void dump(const char * key, const smplpp::Tensor & t)
{
  dump(key, t.toVector<float>());
}

int main(int argc, char ** argv)
{
  const bool expect_no_gpu = !std::strcmp(argv[argc - 1], "--expect-no-gpu");
  const char * vposer_json = (argc > 2 && std::strcmp(argv[2], "--expect-no-gpu")) ? argv[2] : nullptr;
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
      kv.second.targetPos_ = smplpp::torchlike::tensor({0.1f, 0.2f, 0.3f});
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
    // ---- the reference's IkTask methods (include/smplpp/IkTask.h:33-49) on a posed frame, and the index-tensor getVertexRaw
    smplpp::Tensor th2({1, 25, 3});
    for(int i = 3; i < 75; i++) th2.data[(size_t)i] = 0.05f * std::sin(0.7f * (float)i);
    smpl->launch(beta, th2);
    smplpp::IkTask task(smpl, 7, smplpp::torchlike::tensor({0.1f, 0.0f, 0.2f}), smplpp::torchlike::tensor({0.f, 0.f, 1.f}));
    task.normalOffset_ = 0.015;
    task.phi_ = smplpp::torchlike::tensor({0.002f, -0.001f});
    task.calcTangents();
    dump("TANGENTS", task.tangents_);
    const smplpp::Tensor fv = smpl->getFaceIndexRaw(7).to(smplpp::kCPU) - 1; // node/node.cpp:183
    smplpp::Tensor tri = smpl->getVertexRaw(fv.to(smplpp::kInt64)).to(smplpp::kCPU).clone().detach(); // :186
    dump("FACEVERTS", tri.data);
    smplpp::Tensor centroid({3});
    for(int i = 0; i < 3; i++)
      for(int x = 0; x < 3; x++) centroid.data[(size_t)x] += tri.data[(size_t)i * 3 + x] * (i == 0 ? 0.5f : 0.25f);
    task.calcVertexWeights(centroid);
    dump("WEIGHTS", task.vertexWeights_);
    dump("ACTUALPOS", task.calcActualPos());
    dump("ACTUALNORMAL", task.calcActualNormal());
    if(vposer_json)
    {
      // ---- smplpp::VPoserDecoder (include/smplpp/VPoser.h:53-90) and the latent IK layout
      auto vposer = std::make_shared<smplpp::VPoserDecoder>(0);
      vposer->loadParamsFromJson(vposer_json);
      (*vposer)->eval();                              // node/node.cpp:437-438: the holder's calls
      (*vposer)->to(smplpp::Device("CUDA", 0));
      smplpp::Tensor z({2, 32});
      for(int i = 0; i < 64; i++) z.data[(size_t)i] = 0.1f * std::cos(0.37f * (float)i);
      smplpp::Tensor jac;
      smplpp::Tensor aa = vposer->forward(z, &jac);
      dump("VPOSER", aa.data);
      double js = 0.0;
      for(float x : jac.data) js += std::fabs((double)x);
      std::printf("VPOSERJACABS %.9g\n", js);
      smplpp::Tensor rot({1, 3, 3});
      const float c = std::cos(0.3f), sn = std::sin(0.3f);
      rot.data = {c, -sn, 0.f, sn, c, 0.f, 0.f, 0.f, 1.f};
      dump("ROT2AA", smplpp::convertRotMatToAxisAngle(rot).data);
      smplpp::IkSolver ls(smpl, 1, 2, vposer);
      ls.setTaskList(tasks);
      smplpp::Tensor g({1, 44});
      ls.setConfig(beta, g);
      auto le2 = ls.iterate(3);
      smplpp::Tensor b2, g2;
      ls.getConfig(b2, g2);
      std::printf("LATENTIK %.9g %zu\n", le2[0], g2.data.size());
      std::vector<int64_t> tf;
      std::vector<float> tw, tt, tpv, tnv;
      ls.getTasks(tf, tw, tt, tpv, tnv);
      dump("LATENTPOS", tpv);
    }
    std::printf("OK\n");
    return 0;
  }
  catch(const smplpp::Exception & ex)
  {
    std::printf("smplpp::Exception: %s\n", ex.what());
    return expect_no_gpu ? 0 : 1;
  }
}
