// The on-disk formats either side of the capture-fitting path, C++ side — header-only, no dependency beyond the
// shim's Tensor.  What /root/reference/node/node.cpp reads and writes around its two mocap modes:
//   readC3d               the point data of a C3D file — what the node takes from ezc3d (node/node.cpp:580-594: pointNames;
//                         :667-690: frame(i).points().point(j), isEmpty()).  Intel byte order, float or scaled-integer points,
//                         POINT:LABELS (+ LABELS2 ...), RATE, DATA_START; a negative residual marks a missing marker.
//   matchMarkers          the suffix match of task names against point labels (:583-594)
//   baseline41            the OptiTrack Baseline-41 marker -> SMPL face table (:455-500)
//   writeMocapBodyYaml /  /tmp/MocapBody.yaml: beta + per task {name, faceIdx, vertexWeights} (:1418-1441 writes, :509-534 reads)
//   readMocapBodyYaml
//   writeMotionText       one line per stored instant: the 75 numbers of theta, 25 x 3 row-major
//                         (scripts/convertRosbagToText.py:18-19 of the reference, from the motion the node stores at :1389-1398)
// The Python side of the same formats is smplpp_amd/mocap.py; tests/test_mocap_cpu.py reads one file with both.
#ifndef SMPLPP_SHIM_MOCAP_H
#define SMPLPP_SHIM_MOCAP_H

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "Tensor.h"

namespace smplpp
{
struct C3dPoints
{
  std::vector<std::string> labels; // [P]
  double rate = 0.0;
  int64_t frames = 0, points = 0, firstFrame = 0;
  std::vector<float> xyz;          // [T][P][3], file units
  std::vector<uint8_t> valid;      // [T][P]: residual >= 0 (ezc3d: !isEmpty())
  std::string units;
  const float * point(int64_t t, int64_t p) const { return xyz.data() + (t * points + p) * 3; }
  bool isEmpty(int64_t t, int64_t p) const { return !valid[(size_t)(t * points + p)]; }
};

namespace detail
{
template<class T>
inline T rd(const std::vector<unsigned char> & d, size_t off)
{
  if(off + sizeof(T) > d.size()) throw Exception("c3d", "file truncated");
  T v;
  std::memcpy(&v, d.data() + off, sizeof(T));
  return v;
}
inline std::string trimmed(const unsigned char * p, size_t n)
{
  std::string s(reinterpret_cast<const char *>(p), n);
  while(!s.empty() && (s.back() == ' ' || s.back() == '\0')) s.pop_back();
  size_t b = 0;
  while(b < s.size() && s[b] == ' ') b++;
  return s.substr(b);
}
} // namespace detail

inline C3dPoints readC3d(const std::string & path)
{
  std::ifstream f(path, std::ios::binary);
  if(!f) throw Exception("c3d", "cannot open " + path);
  std::vector<unsigned char> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  if(d.size() < 512 || d[1] != 0x50) throw Exception("c3d", "not a C3D file: " + path);
  using detail::rd;
  const int paramBlock = d[0];
  const int npoints = rd<uint16_t>(d, 2), nanalog = rd<uint16_t>(d, 4), first = rd<uint16_t>(d, 6), last = rd<uint16_t>(d, 8);
  const float scale = rd<float>(d, 12);
  int dataBlock = rd<uint16_t>(d, 16);
  double rate = rd<float>(d, 20);
  const size_t p0 = (size_t)(paramBlock - 1) * 512;
  if(rd<uint8_t>(d, p0 + 3) != 84) throw Exception("c3d", "only Intel (little-endian IEEE) C3D files are supported");
  // parameter section: groups (negative id) and parameters (positive id) chained by a 16-bit "next" offset
  std::map<int, std::string> groups;
  struct Param
  {
    int type = 0;
    std::vector<int> dims;
    size_t data = 0;
  };
  std::map<std::string, Param> params; // "<group id>:<NAME>"
  size_t pos = p0 + 4;
  for(;;)
  {
    const int nameLen = rd<int8_t>(d, pos), gid = rd<int8_t>(d, pos + 1);
    const size_t n = (size_t)(nameLen < 0 ? -nameLen : nameLen);
    if(n == 0) break;
    std::string name(reinterpret_cast<const char *>(d.data() + pos + 2), n);
    for(auto & c : name) c = (char)std::toupper((unsigned char)c);
    const size_t nxtAt = pos + 2 + n;
    const int nxt = rd<int16_t>(d, nxtAt);
    if(gid < 0)
      groups[-gid] = name;
    else
    {
      Param p;
      size_t q = nxtAt + 2;
      p.type = rd<int8_t>(d, q);
      const int ndim = d.at(q + 1);
      for(int i = 0; i < ndim; i++) p.dims.push_back(d.at(q + 2 + i));
      p.data = q + 2 + ndim;
      params[std::to_string(gid) + ":" + name] = p;
    }
    if(nxt == 0) break;
    pos = nxtAt + (size_t)nxt;
  }
  int pointGroup = -1;
  for(const auto & g : groups)
    if(g.second == "POINT") pointGroup = g.first;
  auto find = [&](const std::string & name) -> const Param * {
    auto it = params.find(std::to_string(pointGroup) + ":" + name);
    return it == params.end() ? nullptr : &it->second;
  };
  C3dPoints out;
  for(int k = 1;; k++) // files with more than 255 points continue in LABELS2, ...
  {
    const Param * p = find(k == 1 ? std::string("LABELS") : "LABELS" + std::to_string(k));
    if(!p) break;
    if(p->type != -1 || p->dims.size() != 2) break;
    for(int i = 0; i < p->dims[1]; i++) out.labels.push_back(detail::trimmed(d.data() + p->data + (size_t)i * p->dims[0], (size_t)p->dims[0]));
  }
  for(int i = (int)out.labels.size(); i < npoints; i++) out.labels.push_back("*" + std::to_string(i));
  out.labels.resize((size_t)npoints);
  if(const Param * p = find("RATE")) rate = p->type == 4 ? rd<float>(d, p->data) : rate;
  if(const Param * p = find("DATA_START")) dataBlock = rd<uint16_t>(d, p->data);
  if(const Param * p = find("UNITS"))
    if(p->type == -1) out.units = detail::trimmed(d.data() + p->data, p->dims.empty() ? 1 : (size_t)p->dims[0]);
  int64_t frames = (int64_t)last - first + 1;
  if(frames <= 0)
    if(const Param * p = find("FRAMES")) frames = rd<uint16_t>(d, p->data);
  const size_t off = (size_t)(dataBlock - 1) * 512;
  const int64_t words = (int64_t)npoints * 4 + nanalog;
  out.rate = rate;
  out.frames = frames;
  out.points = npoints;
  out.firstFrame = first;
  out.xyz.resize((size_t)(frames * npoints * 3));
  out.valid.resize((size_t)(frames * npoints));
  for(int64_t t = 0; t < frames; t++)
    for(int64_t p = 0; p < npoints; p++)
    {
      float v[4];
      if(scale < 0) // float data
        for(int c = 0; c < 4; c++) v[c] = rd<float>(d, off + (size_t)((t * words + p * 4 + c) * 4));
      else // scaled 16-bit integers; the residual in the fourth word
        for(int c = 0; c < 4; c++) v[c] = (float)rd<int16_t>(d, off + (size_t)((t * words + p * 4 + c) * 2)) * (c < 3 ? scale : 1.0f);
      for(int c = 0; c < 3; c++) out.xyz[(size_t)((t * npoints + p) * 3 + c)] = v[c];
      out.valid[(size_t)(t * npoints + p)] = v[3] >= 0.0f ? 1 : 0;
    }
  return out;
}

// node/node.cpp:583-594: for every task name the first point label that ENDS with it (labels carry a "Skeleton:"-style prefix)
inline std::vector<int64_t> matchMarkers(const std::vector<std::string> & pointLabels, const std::vector<std::string> & taskNames)
{
  std::vector<int64_t> out;
  for(const auto & name : taskNames)
  {
    int64_t idx = -1;
    for(size_t i = 0; i < pointLabels.size() && idx < 0; i++)
    {
      const std::string & s = pointLabels[i];
      if(s.size() >= name.size() && s.compare(s.size() - name.size(), name.size(), name) == 0) idx = (int64_t)i;
    }
    if(idx < 0) throw Exception("node", "mocap marker " + name + " not found");
    out.push_back(idx);
  }
  return out;
}

// node/node.cpp:455-500 (https://docs.optitrack.com/markersets/full-body/baseline-41): marker name -> SMPL face index
inline const std::map<std::string, int64_t> & baseline41()
{
  static const std::map<std::string, int64_t> t = {
      {"HeadTop", 7324}, {"HeadFront", 7194}, {"HeadSide", 13450}, {"Chest", 6842}, {"WaistLFront", 2162}, {"WaistRFront", 13026},
      {"WaistLBack", 5117}, {"WaistRBack", 12007}, {"BackTop", 8914}, {"BackRight", 11433}, {"BackLeft", 4309}, {"LShoulderTop", 2261},
      {"LShoulderBack", 4599}, {"LUArmHigh", 4249}, {"LElbowOut", 4913}, {"LWristIn", 4091}, {"LWristOut", 2567}, {"LHandOut", 2636},
      {"RShoulderTop", 13583}, {"RShoulderBack", 11491}, {"RUArmHigh", 11137}, {"RElbowOut", 11802}, {"RWristIn", 9712}, {"RWristOut", 9590},
      {"RHandOut", 9733}, {"LThigh", 1122}, {"LKneeOut", 1165}, {"LShin", 1247}, {"LAnkleOut", 5742}, {"LToeIn", 5758}, {"LToeOut", 6000},
      {"LToeTip", 5591}, {"LHeel", 5815}, {"RThigh", 8523}, {"RKneeOut", 8053}, {"RShin", 12108}, {"RAnkleOut", 12630}, {"RToeIn", 12896},
      {"RToeOut", 12889}, {"RToeTip", 12478}, {"RHeel", 12705}};
  return t;
}

struct MocapBodyTask
{
  std::string name;
  int64_t faceIdx = 0;
  float vertexWeights[3] = {1.f / 3, 1.f / 3, 1.f / 3};
};

// /tmp/MocapBody.yaml as node/node.cpp:1426-1441 writes it (Eigen::FullPrecision, ", " separators, one bracketed row)
inline void writeMocapBodyYaml(const std::string & path, const Tensor & beta, const std::vector<MocapBodyTask> & tasks)
{
  if(beta.numel() != 10) throw Exception("node", "Size of beta must be 10 but " + std::to_string(beta.numel())); // :513-517
  std::ofstream f(path);
  if(!f) throw Exception("node", "cannot write " + path);
  auto num = [](double x) {
    char b[40];
    std::snprintf(b, sizeof b, "%.9g", x);
    return std::string(b);
  };
  f << "beta: [";
  for(int i = 0; i < 10; i++) f << (i ? ", " : "") << num(beta.at(i));
  f << "]\nikTaskList:\n";
  for(const auto & t : tasks)
    f << "  - name: " << t.name << "\n    faceIdx: " << t.faceIdx << "\n    vertexWeights: [" << num(t.vertexWeights[0]) << ", " << num(t.vertexWeights[1])
      << ", " << num(t.vertexWeights[2]) << "]\n";
}

// node/node.cpp:509-534: the subset of YAML that file uses (two top-level keys; flow sequences of numbers; a block sequence of maps)
inline void readMocapBodyYaml(const std::string & path, Tensor & beta, std::vector<MocapBodyTask> & tasks)
{
  std::ifstream f(path);
  if(!f) throw Exception("node", "cannot open " + path);
  auto numbers = [](const std::string & s) {
    std::vector<double> v;
    const size_t a = s.find('['), b = s.rfind(']');
    if(a == std::string::npos || b == std::string::npos) throw Exception("node", "MocapBody.yaml: expected a [ ... ] sequence");
    std::stringstream ss(s.substr(a + 1, b - a - 1));
    std::string tok;
    while(std::getline(ss, tok, ',')) v.push_back(std::stod(tok));
    return v;
  };
  beta = Tensor({10});
  tasks.clear();
  std::string ln;
  while(std::getline(f, ln))
  {
    const size_t c = ln.find(':');
    if(c == std::string::npos) continue;
    std::string key = ln.substr(0, c), val = ln.substr(c + 1);
    size_t b = key.find_first_not_of(" -");
    key = b == std::string::npos ? "" : key.substr(b);
    const size_t vb = val.find_first_not_of(' ');
    val = vb == std::string::npos ? "" : val.substr(vb);
    if(key == "beta")
    {
      const std::vector<double> v = numbers(val);
      if(v.size() != 10) throw Exception("node", "Size of beta must be 10 but " + std::to_string(v.size())); // node.cpp:513-517
      for(int i = 0; i < 10; i++) beta.data[(size_t)i] = (float)v[(size_t)i];
    }
    else if(key == "name")
    {
      tasks.emplace_back();
      tasks.back().name = val;
    }
    else if(key == "faceIdx" && !tasks.empty())
      tasks.back().faceIdx = std::stoll(val);
    else if(key == "vertexWeights" && !tasks.empty())
    {
      const std::vector<double> v = numbers(val);
      if(v.size() != 3) throw Exception("node", "vertexWeights must hold three numbers");
      for(int i = 0; i < 3; i++) tasks.back().vertexWeights[i] = (float)v[(size_t)i];
    }
  }
}

// One line per instant: the 75 numbers of theta (25 x 3 row-major), as scripts/convertRosbagToText.py:18-19 prints the stored motion
inline void writeMotionText(const std::string & path, const std::vector<Tensor> & thetaPerInstant)
{
  std::ofstream f(path);
  if(!f) throw Exception("node", "cannot write " + path);
  for(const auto & th : thetaPerInstant)
  {
    if(th.numel() != 75) throw Exception("node", "theta must hold 25 x 3 numbers");
    for(int i = 0; i < 75; i++)
    {
      char b[40];
      std::snprintf(b, sizeof b, "%.17g", th.at(i));
      f << (i ? " " : "") << b;
    }
    f << "\n";
  }
}
} // namespace smplpp
#endif
