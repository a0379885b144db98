// The C++ side of the capture path's on-disk formats (include/smplpp/Mocap.h) — no GPU needed.  Driven by tests/test_mocap_cpu.py:
//   mocap_formats <in.c3d> <dump.txt> <body.yaml> <motion.txt>
// reads the C3D file, writes "labels / rate / frames / points" and a checksum of every frame to dump.txt (the Python reader of the
// same file must agree), matches the Baseline-41 names against the labels, writes a MocapBody.yaml and a motion text, and reads
// the yaml back.
#include <cstdio>
#include <fstream>

#include <smplpp/Mocap.h>

int main(int argc, char ** argv)
{
  if(argc < 5) return 64;
  try
  {
    const smplpp::C3dPoints c = smplpp::readC3d(argv[1]);
    std::ofstream d(argv[2]);
    d << "rate " << c.rate << " frames " << c.frames << " points " << c.points << " first " << c.firstFrame << " units " << c.units << "\n";
    for(const auto & l : c.labels) d << "label " << l << "\n";
    for(int64_t t = 0; t < c.frames; t++)
    {
      double sum = 0.0;
      int64_t nvalid = 0;
      for(int64_t p = 0; p < c.points; p++)
      {
        if(c.isEmpty(t, p)) continue;
        nvalid++;
        const float * x = c.point(t, p);
        sum += (double)x[0] + 2.0 * (double)x[1] + 3.0 * (double)x[2];
      }
      char b[96];
      std::snprintf(b, sizeof b, "frame %lld %lld %.9g\n", (long long)t, (long long)nvalid, sum);
      d << b;
    }
    std::vector<std::string> names;
    for(const auto & kv : smplpp::baseline41()) names.push_back(kv.first); // std::map order, as g_ikTaskList iterates (node.cpp:47)
    const std::vector<int64_t> idx = smplpp::matchMarkers(c.labels, names);
    for(size_t i = 0; i < names.size(); i++) d << "match " << names[i] << " " << idx[i] << "\n";
    // MocapBody.yaml round trip
    smplpp::Tensor beta({10});
    for(int i = 0; i < 10; i++) beta.data[(size_t)i] = 0.1f * (float)(i - 4) + 1e-7f * (float)i;
    std::vector<smplpp::MocapBodyTask> tasks;
    for(const auto & kv : smplpp::baseline41())
    {
      smplpp::MocapBodyTask t;
      t.name = kv.first;
      t.faceIdx = kv.second;
      t.vertexWeights[0] = 0.2f + 0.001f * (float)tasks.size();
      t.vertexWeights[1] = 0.3f;
      t.vertexWeights[2] = 1.0f - t.vertexWeights[0] - t.vertexWeights[1];
      tasks.push_back(t);
    }
    smplpp::writeMocapBodyYaml(argv[3], beta, tasks);
    smplpp::Tensor beta2;
    std::vector<smplpp::MocapBodyTask> tasks2;
    smplpp::readMocapBodyYaml(argv[3], beta2, tasks2);
    bool same = tasks2.size() == tasks.size();
    for(int i = 0; i < 10 && same; i++) same = beta2.data[(size_t)i] == beta.data[(size_t)i];
    for(size_t i = 0; i < tasks.size() && same; i++)
      same = tasks2[i].name == tasks[i].name && tasks2[i].faceIdx == tasks[i].faceIdx && tasks2[i].vertexWeights[0] == tasks[i].vertexWeights[0]
             && tasks2[i].vertexWeights[2] == tasks[i].vertexWeights[2];
    d << "yaml_roundtrip " << (same ? 1 : 0) << "\n";
    std::vector<smplpp::Tensor> motion;
    for(int t = 0; t < 3; t++)
    {
      smplpp::Tensor th({25, 3});
      for(int i = 0; i < 75; i++) th.data[(size_t)i] = 0.01f * (float)(i + 75 * t) + 1e-6f;
      motion.push_back(th);
    }
    smplpp::writeMotionText(argv[4], motion);
    std::printf("OK\n");
    return 0;
  }
  catch(const smplpp::Exception & ex)
  {
    std::printf("smplpp::Exception: %s\n", ex.what());
    return 1;
  }
}
