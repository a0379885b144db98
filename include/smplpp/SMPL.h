// smplpp::SMPL / smplpp::IkTask over the C ABI of libsmplpp_hip.so — header-only C++ shim.
//
// Keeps the reference's class names, method names and argument meaning
// (/root/reference/include/smplpp/SMPL.h:241-269, include/smplpp/IkTask.h:20-84) so that node/node.cpp-style callers
// re-link against the MI355X engine.  `torch::Tensor` is replaced by the minimal owning host container `smplpp::Tensor`
// (Tensor.h: the methods node.cpp applies to these results, under torch's names); errors become `smplpp::Exception` like
// smpl_error (include/smplpp/toolbox/Exception.h:48-49).  Signatures follow the reference header line by line: the
// three-argument constructor, copy construction / assignment, setVertPath + out(index), getFaceIndex / getFaceIndexRaw as tensors
// of kInt32 (1-based), getVertexRaw(index tensor), getAdjacentFaces as a reference to an unordered_map.
// The one semantic change is the autograd seam: there is no backward(); use IkSolver::eval()/iterate().
#ifndef SMPLPP_SHIM_SMPL_H
#define SMPLPP_SHIM_SMPL_H

#include <array>
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../smplpp_hip.h"
#include "Tensor.h"

namespace smplpp
{
constexpr int64_t JOINT_NUM = SMPLPP_JOINT_NUM;
constexpr int64_t SHAPE_BASIS_DIM = SMPLPP_SHAPE_BASIS_DIM;
constexpr int64_t POSE_BASIS_DIM = SMPLPP_POSE_BASIS_DIM;
constexpr int64_t LATENT_DIM = SMPLPP_LATENT_DIM;

inline void check(int rc, const char * module)
{
  if(rc != SMPLPP_OK) throw Exception(module, smplpp_last_error());
}
// the reference's name for its exception (include/smplpp/toolbox/Exception.h:48-49: smpl_error(module, message))
using smpl_error = Exception;

// An index list as a tensor of kInt64 (`getVertexRaw(faceVertexIdxs.to(torch::kInt64))`, node/node.cpp:186)
struct IndexTensor : Tensor
{
  IndexTensor() : Tensor({0}, kInt64) {}
  IndexTensor(std::initializer_list<int64_t> v) : Tensor({(int64_t)v.size()}, kInt64) { idata.assign(v.begin(), v.end()); }
  explicit IndexTensor(const std::vector<int64_t> & v) : Tensor({(int64_t)v.size()}, kInt64) { idata = v; }
};

namespace detail
{
// Just enough JSON for the model files scripts/preprocess.py writes: an object of (nested) numeric arrays.
struct JsonArrays
{
  std::map<std::string, std::vector<double>> values;
  std::map<std::string, std::vector<int64_t>> shapes;
  explicit JsonArrays(const std::string & path)
  {
    std::ifstream f(path, std::ios::binary);
    if(!f) throw Exception("SMPL", "Cannot initialize a SMPL model!"); // src/SMPL.cpp:616
    std::stringstream ss;
    ss << f.rdbuf();
    s_ = ss.str();
    skip();
    expect('{');
    for(;;)
    {
      skip();
      if(peek() == '}') break;
      std::string key = str();
      skip();
      expect(':');
      std::vector<int64_t> shape;
      std::vector<double> vals;
      array(vals, shape, 0);
      values[key] = std::move(vals);
      shapes[key] = std::move(shape);
      skip();
      if(peek() == ',') p_++;
    }
  }

private:
  std::string s_;
  size_t p_ = 0;
  char peek() const { return p_ < s_.size() ? s_[p_] : '\0'; }
  void skip()
  {
    while(p_ < s_.size() && std::isspace((unsigned char)s_[p_])) p_++;
  }
  void expect(char c)
  {
    if(peek() != c) throw Exception("SMPL", std::string("model json: expected '") + c + "'");
    p_++;
  }
  std::string str()
  {
    expect('"');
    size_t b = p_;
    while(p_ < s_.size() && s_[p_] != '"') p_++;
    std::string r = s_.substr(b, p_ - b);
    p_++;
    return r;
  }
  void array(std::vector<double> & vals, std::vector<int64_t> & shape, size_t depth)
  {
    skip();
    if(peek() != '[')
    {
      char * end = nullptr;
      vals.push_back(std::strtod(s_.c_str() + p_, &end));
      p_ = (size_t)(end - s_.c_str());
      return;
    }
    p_++;
    int64_t n = 0;
    for(;;)
    {
      skip();
      if(peek() == ']')
      {
        p_++;
        break;
      }
      array(vals, shape, depth + 1);
      n++;
      skip();
      if(peek() == ',') p_++;
    }
    if(shape.size() <= depth) shape.resize(depth + 1, 0);
    shape[depth] = n;
  }
};
} // namespace detail

class SMPL
{
public:
  // %% Constructor and Destructor %% (reference include/smplpp/SMPL.h:241-247, src/SMPL.cpp:100-262)
  SMPL() = default;
  SMPL(const std::string & modelPath, const std::string & vertPath, const Device & device) : vertPath_(vertPath)
  {
    setDevice(device);
    setModelPath(modelPath);
  }
  // A copy shares the engine's model (immutable once created) and takes a copy of the outputs of the last launch — the
  // reference deep-copies its tensors (src/SMPL.cpp:160-262); std::map<std::string, IkTask> and the node's globals need no more.
  SMPL(const SMPL &) = default;
  SMPL & operator=(const SMPL &) = default;
  ~SMPL() = default;

  // %% Setter and Getter %%
  void setDevice(const Device & device)
  {
    if(!device.has_index()) throw Exception("SMPL", "Failed to fetch device index!"); // src/SMPL.cpp:289-297
    if(device.type == "CPU" || device.type == "cpu") throw Exception("SMPL", "libsmplpp_hip has no CPU engine");
    device_ = device;
  }
  Device getDevice() const { return device_; }
  void setModelPath(const std::string & modelPath) { path_ = modelPath; }
  void setVertPath(const std::string & vertexPath) { vertPath_ = vertexPath; } // src/SMPL.cpp:361

  // SMPL::init (src/SMPL.cpp:560-643): parse the .json written by scripts/preprocess.py and create the engine model.
  void init()
  {
    detail::JsonArrays j(path_);
    auto f32 = [&](const char * k) {
      auto it = j.values.find(k);
      if(it == j.values.end()) throw Exception("SMPL", std::string("model json lacks ") + k);
      return std::vector<float>(it->second.begin(), it->second.end());
    };
    std::vector<float> vt = f32("vertices_template"), S = f32("shape_blend_shapes"), P = f32("pose_blend_shapes"),
                       Jr = f32("joint_regressor"), W = f32("weights");
    const auto & kv = j.values.at("kinematic_tree");
    const auto & fv = j.values.at("face_indices");
    std::vector<int64_t> kin(kv.begin(), kv.end());
    std::vector<int32_t> faces(fv.begin(), fv.end());
    initFromArrays((int64_t)vt.size() / 3, (int64_t)faces.size() / 3, vt.data(), S.data(), P.data(), Jr.data(), W.data(),
                   kin.data(), faces.data());
  }
  void initFromArrays(int64_t V, int64_t F, const float * vt, const float * S, const float * P, const float * Jreg,
                      const float * W, const int64_t * kintree, const int32_t * faces1)
  {
    smplpp_model * m = nullptr;
    check(smplpp_model_create(V, F, vt, S, P, Jreg, W, kintree, faces1, device_.index, &m), "SMPL");
    m_ = std::shared_ptr<smplpp_model>(m, [](smplpp_model * p) { smplpp_model_destroy(p); });
    V_ = V;
    F_ = F;
    faces1_ = Tensor({F, 3}, kInt32); // 1-based like the model file (scripts/preprocess.py:91)
    faces1_.idata.assign(faces1, faces1 + F * 3);
    // adjacent faces per vertex with uniform weights 1 / degree (src/SMPL.cpp:620-640), kept for getAdjacentFaces
    auto adj = std::make_shared<std::vector<std::unordered_map<int64_t, float>>>((size_t)V);
    std::vector<int64_t> fbuf(256);
    std::vector<float> wbuf(256);
    for(int64_t v = 0; v < V; v++)
    {
      int64_t cnt = 0;
      check(smplpp_adjacent_faces(m, v, (int64_t)fbuf.size(), fbuf.data(), wbuf.data(), &cnt), "SMPL");
      if(cnt > (int64_t)fbuf.size())
      {
        fbuf.resize((size_t)cnt);
        wbuf.resize((size_t)cnt);
        check(smplpp_adjacent_faces(m, v, cnt, fbuf.data(), wbuf.data(), &cnt), "SMPL");
      }
      for(int64_t i = 0; i < cnt; i++) (*adj)[(size_t)v][fbuf[(size_t)i]] = wbuf[(size_t)i];
    }
    adjacent_ = adj;
  }

  // SMPL::launch (src/SMPL.cpp:671-737): beta [N,10], theta [N,25,3] (row 0 = root translation)
  void launch(const Tensor & beta, const Tensor & theta)
  {
    if(!m_ || beta.shape.size() != 2 || beta.size(1) != SHAPE_BASIS_DIM || theta.shape.size() != 3
       || theta.size(0) != beta.size(0) || theta.size(1) != JOINT_NUM + 1 || theta.size(2) != 3 || beta.dtype != kFloat32 || theta.dtype != kFloat32)
      throw Exception("SMPL", "Cannot launch a SMPL model!");
    const int64_t n = beta.size(0);
    verts_ = Tensor({n, V_, 3});
    rest_ = Tensor({n, V_, 3});
    joints_ = Tensor({n, JOINT_NUM, 3});
    xforms_ = Tensor({n, JOINT_NUM, 4, 4});
    check(smplpp_fk(m_.get(), n, beta.ptr(), theta.ptr(), verts_.ptr(), joints_.ptr(), xforms_.ptr(), rest_.ptr(), SMPLPP_HOST, nullptr),
          "SMPL");
  }

  Tensor getVertex() const { return need(verts_); }       // [N,6890,3] copy (src/SMPL.cpp:492-506)
  Tensor getRestShape() const { return need(rest_); }
  Tensor getRestJoint() const { return need(joints_); }   // [N,24,3]
  Tensor getTransformation() const { return need(xforms_); }
  // batch 0 only, like src/LinearBlendSkinning.cpp:419-427
  Tensor getVertexRaw(int64_t idx) const
  {
    need(verts_);
    if(idx < 0 || idx >= V_) throw Exception("LinearBlendSknning", "vertex index out of range");
    Tensor t({3});
    for(int x = 0; x < 3; x++) t.data[x] = verts_.data[(size_t)idx * 3 + x];
    return t;
  }
  // index-tensor overload (include/smplpp/SMPL.h:257 of the reference; src/LinearBlendSkinning.cpp:424-427): rows of
  // batch 0 for a list of vertex ids -> [len, 3]
  Tensor getVertexRaw(const Tensor & idx) const
  {
    need(verts_);
    if(idx.dtype != kInt64 && idx.dtype != kInt32) throw Exception("LinearBlendSknning", "vertex indices must be an integer tensor");
    Tensor t({idx.numel(), 3});
    for(int64_t i = 0; i < idx.numel(); i++)
    {
      const int64_t v = idx.idata[(size_t)i];
      if(v < 0 || v >= V_) throw Exception("LinearBlendSknning", "vertex index out of range");
      for(int x = 0; x < 3; x++) t.data[(size_t)i * 3 + x] = verts_.data[(size_t)v * 3 + x];
    }
    return t;
  }
  Tensor getFaceIndex() const { return faces1_; } // [F,3] kInt32, 1-based (src/SMPL.cpp:418-433)
  Tensor getFaceIndexRaw(int64_t idx) const        // [3] kInt32, 1-based (:435-438)
  {
    if(idx < 0 || idx >= F_) throw Exception("SMPL", "face index out of range");
    return faces1_.index({idx});
  }
  Tensor calcNormal(int64_t faceIdx) const // src/SMPL.cpp:518-525 (batch 0)
  {
    need(verts_);
    Tensor t({3});
    check(smplpp_face_normals(m_.get(), 1, verts_.ptr(), 1, &faceIdx, t.ptr(), SMPLPP_HOST, nullptr), "SMPL");
    return t;
  }
  Tensor calcVertexNormal(int64_t idx) const // src/SMPL.cpp:527-535 (batch 0)
  {
    need(verts_);
    Tensor t({3});
    check(smplpp_vertex_normals(m_.get(), 1, verts_.ptr(), 1, &idx, t.ptr(), SMPLPP_HOST, nullptr), "SMPL");
    return t;
  }
  const std::unordered_map<int64_t, float> & getAdjacentFaces(int64_t idx) const // src/SMPL.cpp:537-540
  {
    if(!adjacent_ || idx < 0 || idx >= V_) throw Exception("SMPL", "Cannot get adjacent faces!");
    return (*adjacent_)[(size_t)idx];
  }
  // SMPL::calcVertexNormal for every vertex of every frame of the last launch: [N,V,3]
  Tensor calcMeshVertexNormals() const
  {
    need(verts_);
    Tensor t(verts_.shape);
    check(smplpp_mesh_vertex_normals(m_.get(), verts_.size(0), verts_.ptr(), t.ptr(), SMPLPP_HOST, nullptr), "SMPL");
    return t;
  }
  // The sweep grid of node/node.cpp:1023-1073 for frame `index`: the grid indices (cell position = 0.025 m x index) whose
  // winding number exceeds 0.5 — the keys the reference enters into g_sweepGridList
  std::vector<std::array<int32_t, 3>> calcSweepGrid(int64_t index = 0) const
  {
    need(verts_);
    const float * v = verts_.ptr() + (size_t)index * V_ * 3;
    int32_t g0[3], gn[3];
    int64_t cells = 0;
    check(smplpp_sweep_grid(m_.get(), v, g0, gn, 0, nullptr, nullptr, &cells, SMPLPP_HOST, nullptr), "SMPL");
    std::vector<uint8_t> inside((size_t)cells);
    check(smplpp_sweep_grid(m_.get(), v, g0, gn, cells, nullptr, inside.data(), &cells, SMPLPP_HOST, nullptr), "SMPL");
    std::vector<std::array<int32_t, 3>> out;
    int64_t i = 0;
    for(int32_t x = 0; x < gn[0]; x++)
      for(int32_t y = 0; y < gn[1]; y++)
        for(int32_t z = 0; z < gn[2]; z++, i++)
          if(inside[(size_t)i]) out.push_back({g0[0] + x, g0[1] + y, g0[2] + z});
    return out;
  }
  // SMPL::out (src/SMPL.cpp:757-790): Wavefront OBJ of frame `index` of the last launch into the path of setVertPath
  void out(int64_t index) const
  {
    if(verts_.data.empty() || index < 0 || index >= verts_.size(0) || vertPath_.empty()) throw Exception("SMPL", "Cannot export the deformed mesh!");
    std::ofstream f(vertPath_);
    if(!f) throw Exception("SMPL", "Cannot export the deformed mesh!");
    for(int64_t v = 0; v < V_; v++)
      f << 'v' << ' ' << verts_.data[((size_t)index * V_ + v) * 3] << ' ' << verts_.data[((size_t)index * V_ + v) * 3 + 1] << ' '
        << verts_.data[((size_t)index * V_ + v) * 3 + 2] << '\n';
    for(int64_t t = 0; t < F_; t++)
      f << 'f' << ' ' << faces1_.idata[(size_t)t * 3] << ' ' << faces1_.idata[(size_t)t * 3 + 1] << ' ' << faces1_.idata[(size_t)t * 3 + 2] << '\n';
  }

  smplpp_model * handle() const { return m_.get(); }
  int64_t vertexNum() const { return V_; }
  int64_t faceNum() const { return F_; }

private:
  static const Tensor & need(const Tensor & t)
  {
    if(t.data.empty()) throw Exception("LinearBlendSknning", "Failed to get vertices of new pose!"); // LinearBlendSkinning.cpp:413
    return t;
  }
  std::shared_ptr<smplpp_model> m_;
  std::shared_ptr<const std::vector<std::unordered_map<int64_t, float>>> adjacent_;
  Device device_;
  std::string path_, vertPath_;
  int64_t V_ = 0, F_ = 0;
  Tensor faces1_;
  Tensor verts_, rest_, joints_, xforms_;
};
} // namespace smplpp
#endif
