// smplpp::Tensor — the container that stands in for torch::Tensor at the boundary of the C++ shim ("a minimal
// owning array type").  Host memory, row-major, one of four element types, and exactly the methods the reference's one caller
// applies to what smplpp::SMPL / smplpp::IkTask hand out (/root/reference/node/node.cpp:106-123, 183-220, 681-699, 752-777, 803-814,
// 947-998; src/IkTask.cpp:33-86): index / index_put_ / view / to / clone / detach / zero_ / item / data_ptr, the few arithmetic
// operators, torch::zeros / empty / tensor / matmul and at::dot.  It computes nothing on its own account: every number it
// holds was produced by libsmplpp_hip.so or by the caller.  `namespace torch = smplpp::torchlike; namespace at =
// smplpp::torchlike::at;` makes node.cpp-style call sites compile as they are written (tests/cpp/node_loop.cpp).
#ifndef SMPLPP_SHIM_TENSOR_H
#define SMPLPP_SHIM_TENSOR_H

#include <cmath>
#include <cstdint>
#include <initializer_list>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

namespace smplpp
{
class Exception : public std::runtime_error
{
public:
  Exception(const std::string & module, const std::string & msg) : std::runtime_error("[" + module + "] " + msg) {}
};

enum ScalarType
{
  kFloat32 = 0,
  kInt32 = 1,
  kInt64 = 2,
  kFloat64 = 3
};
struct DeviceType // torch::kCPU / torch::kCUDA as values (`tensor.to(torch::kCPU)` is the identity here: the container is host memory)
{
  int id;
};
constexpr DeviceType kCPU{0}, kCUDA{1};

// Device naming follows node/node.cpp:360-371: "CUDA"/"HIP" with an explicit index selects the GPU engine.
struct Device
{
  std::string type = "CUDA";
  int index = 0;
  Device() = default;
  Device(const std::string & t, int i) : type(t), index(i) {}
  Device(DeviceType t, int i) : type(t.id == 0 ? "CPU" : "CUDA"), index(i) {}
  explicit Device(DeviceType t) : type(t.id == 0 ? "CPU" : "CUDA"), index(-1) {} // node/node.cpp:362-366, followed by set_index(0) (:372)
  void set_index(int i) { index = i; }
  bool has_index() const { return index >= 0; }
};

struct Slice // at::indexing::Slice(start, stop)
{
  int64_t start, stop;
  Slice(int64_t a, int64_t b) : start(a), stop(b) {}
};

class Tensor
{
public:
  std::vector<int64_t> shape;
  ScalarType dtype = kFloat32;
  std::vector<float> data;    // kFloat32 values
  std::vector<int64_t> idata; // kInt32 / kInt64 values
  std::vector<double> ddata;  // kFloat64 values

  Tensor() = default;
  explicit Tensor(std::vector<int64_t> s, float fill = 0.0f) : shape(std::move(s)), data((size_t)numel_of(shape), fill) {}
  Tensor(std::vector<int64_t> s, ScalarType t) : shape(std::move(s)), dtype(t)
  {
    const size_t n = (size_t)numel_of(shape);
    if(t == kFloat32)
      data.assign(n, 0.0f);
    else if(t == kFloat64)
      ddata.assign(n, 0.0);
    else
      idata.assign(n, 0);
  }
  // a 0-d tensor holding one value (what at::dot returns; the right-hand side of index_put_)
  static Tensor scalar(float v)
  {
    Tensor t;
    t.data.assign(1, v);
    return t;
  }

  static int64_t numel_of(const std::vector<int64_t> & s)
  {
    int64_t n = 1;
    for(auto d : s) n *= d;
    return n;
  }
  bool defined() const { return !(data.empty() && idata.empty() && ddata.empty()); }
  int64_t numel() const { return numel_of(shape); }
  int64_t dim() const { return (int64_t)shape.size(); }
  int64_t size(int64_t i) const { return shape.at((size_t)(i < 0 ? i + (int64_t)shape.size() : i)); }
  const std::vector<int64_t> & sizes() const { return shape; }
  ScalarType scalar_type() const { return dtype; }
  float * ptr() { return data.data(); }
  const float * ptr() const { return data.data(); }

  // element i as a double, whatever the type
  double at(int64_t i) const
  {
    return dtype == kFloat32 ? (double)data[(size_t)i] : (dtype == kFloat64 ? ddata[(size_t)i] : (double)idata[(size_t)i]);
  }
  void set(int64_t i, double v)
  {
    if(dtype == kFloat32)
      data[(size_t)i] = (float)v;
    else if(dtype == kFloat64)
      ddata[(size_t)i] = v;
    else
      idata[(size_t)i] = (int64_t)v;
  }

  // ---- torch::Tensor's names
  Tensor clone() const { return *this; }
  Tensor detach() const { return *this; }
  Tensor contiguous() const { return *this; }
  Tensor to(DeviceType) const { return *this; }
  Tensor to(const Device &) const { return *this; }
  Tensor to(ScalarType t) const
  {
    if(t == dtype) return *this;
    Tensor r(shape, t);
    for(int64_t i = 0; i < numel(); i++) r.set(i, at(i));
    return r;
  }
  Tensor & zero_()
  {
    for(auto & x : data) x = 0.0f;
    for(auto & x : idata) x = 0;
    for(auto & x : ddata) x = 0.0;
    return *this;
  }
  Tensor & fill_(double v)
  {
    for(int64_t i = 0; i < numel(); i++) set(i, v);
    return *this;
  }
  Tensor & set_requires_grad(bool) { return *this; } // (no autograd behind this container: smplpp::IkSolver::eval returns the Jacobian)
  Tensor view(std::initializer_list<int64_t> s) const
  {
    std::vector<int64_t> ns(s);
    int64_t known = 1, infer = -1;
    for(size_t i = 0; i < ns.size(); i++)
      if(ns[i] == -1)
        infer = (int64_t)i;
      else
        known *= ns[i];
    if(infer >= 0) ns[(size_t)infer] = known ? numel() / known : 0;
    if(numel_of(ns) != numel()) throw Exception("Tensor", "view: shape does not hold the tensor's elements");
    Tensor r = *this;
    r.shape = ns;
    return r;
  }
  Tensor view_as(const Tensor & o) const
  {
    Tensor r = *this;
    if(o.numel() != numel()) throw Exception("Tensor", "view_as: element counts differ");
    r.shape = o.shape;
    return r;
  }
  // index({i, j, ...}): successive selection along the leading dimensions
  Tensor index(std::initializer_list<int64_t> idx) const
  {
    if(idx.size() > shape.size()) throw Exception("Tensor", "index: too many indices");
    int64_t off = 0, block = numel();
    size_t d = 0;
    for(int64_t i : idx)
    {
      block /= shape[d];
      if(i < 0) i += shape[d];
      if(i < 0 || i >= shape[d]) throw Exception("Tensor", "index out of range");
      off += i * block;
      d++;
    }
    Tensor r(std::vector<int64_t>(shape.begin() + (long)d, shape.end()), dtype);
    for(int64_t k = 0; k < block; k++) r.set(k, at(off + k));
    return r;
  }
  // index({Slice(a, b)}): rows [a, b) of the leading dimension
  Tensor index(std::initializer_list<Slice> sl) const
  {
    if(sl.size() != 1 || shape.empty()) throw Exception("Tensor", "index: one slice of the leading dimension");
    const Slice s = *sl.begin();
    const int64_t block = numel() / shape[0];
    if(s.start < 0 || s.stop > shape[0] || s.start > s.stop) throw Exception("Tensor", "slice out of range");
    std::vector<int64_t> ns = shape;
    ns[0] = s.stop - s.start;
    Tensor r(ns, dtype);
    for(int64_t k = 0; k < r.numel(); k++) r.set(k, at(s.start * block + k));
    return r;
  }
  Tensor & index_put_(std::initializer_list<int64_t> idx, const Tensor & v)
  {
    int64_t off = 0, block = numel();
    size_t d = 0;
    for(int64_t i : idx)
    {
      block /= shape.at(d);
      if(i < 0) i += shape[d];
      if(i < 0 || i >= shape[d]) throw Exception("Tensor", "index out of range");
      off += i * block;
      d++;
    }
    if(v.numel() != block && v.numel() != 1) throw Exception("Tensor", "index_put_: value does not fit");
    for(int64_t k = 0; k < block; k++) set(off + k, v.at(v.numel() == 1 ? 0 : k));
    return *this;
  }
  Tensor & index_put_(std::initializer_list<int64_t> idx, double v) { return index_put_(idx, scalar((float)v)); }
  Tensor & index_put_(std::initializer_list<Slice> sl, const Tensor & v)
  {
    const Slice s = *sl.begin();
    const int64_t block = numel() / shape.at(0);
    if(s.start < 0 || s.stop > shape[0] || v.numel() != (s.stop - s.start) * block) throw Exception("Tensor", "index_put_: slice does not fit");
    for(int64_t k = 0; k < v.numel(); k++) set(s.start * block + k, v.at(k));
    return *this;
  }
  template<class T>
  T item() const
  {
    if(numel() != 1) throw Exception("Tensor", "item: not a single element");
    return (T)at(0);
  }
  // typed view of the storage (float, int64_t, double: the storage itself; int32_t: a converted copy kept beside it, refreshed on
  // every call, like the reference's face tensors of kInt32)
  template<class T>
  const T * data_ptr() const
  {
    if constexpr(std::is_same<T, float>::value)
    {
      if(dtype != kFloat32) throw Exception("Tensor", "data_ptr<float> of a tensor of another type");
      return data.data();
    }
    else if constexpr(std::is_same<T, double>::value)
    {
      if(dtype != kFloat64) throw Exception("Tensor", "data_ptr<double> of a tensor of another type");
      return ddata.data();
    }
    else if constexpr(std::is_same<T, int64_t>::value)
    {
      if(dtype != kInt64 && dtype != kInt32) throw Exception("Tensor", "data_ptr<int64_t> of a tensor of another type");
      return idata.data();
    }
    else
    {
      static_assert(std::is_same<T, int32_t>::value, "data_ptr<T>: float, double, int32_t or int64_t");
      if(dtype != kInt32 && dtype != kInt64) throw Exception("Tensor", "data_ptr<int32_t> of a tensor of another type");
      i32_.assign(idata.begin(), idata.end());
      return i32_.data();
    }
  }
  template<class T>
  std::vector<T> toVector() const
  {
    std::vector<T> v((size_t)numel());
    for(int64_t i = 0; i < numel(); i++) v[(size_t)i] = (T)at(i);
    return v;
  }

  Tensor & operator+=(const Tensor & o)
  {
    if(o.numel() != numel()) throw Exception("Tensor", "+=: element counts differ");
    if(dtype == kFloat32 && o.dtype == kFloat32)
      for(size_t i = 0; i < data.size(); i++) data[i] = data[i] + o.data[i]; // fp32 like the reference's update (node/node.cpp:947)
    else
      for(int64_t i = 0; i < numel(); i++) set(i, at(i) + o.at(i));
    return *this;
  }

private:
  mutable std::vector<int32_t> i32_;
};

namespace detail
{
inline Tensor binary(const Tensor & a, const Tensor & b, int op)
{
  if(a.numel() != b.numel() && b.numel() != 1) throw Exception("Tensor", "operator: element counts differ");
  const ScalarType t = (a.dtype == kFloat32 || b.dtype == kFloat32) ? kFloat32 : (a.dtype == kFloat64 || b.dtype == kFloat64 ? kFloat64 : a.dtype);
  Tensor r(a.shape, t);
  for(int64_t i = 0; i < a.numel(); i++)
  {
    const double x = a.at(i), y = b.at(b.numel() == 1 ? 0 : i);
    if(t == kFloat32) // fp32 arithmetic on fp32 tensors, like libtorch's
      r.data[(size_t)i] = op == 0 ? (float)x + (float)y : (op == 1 ? (float)x - (float)y : (float)x * (float)y);
    else
      r.set(i, op == 0 ? x + y : (op == 1 ? x - y : x * y));
  }
  return r;
}
} // namespace detail
inline Tensor operator+(const Tensor & a, const Tensor & b) { return detail::binary(a, b, 0); }
inline Tensor operator-(const Tensor & a, const Tensor & b) { return detail::binary(a, b, 1); }
inline Tensor operator-(const Tensor & a, double s)
{
  Tensor r = a;
  for(int64_t i = 0; i < a.numel(); i++) r.set(i, a.dtype == kFloat32 ? (double)((float)a.at(i) - (float)s) : a.at(i) - s);
  return r;
}
inline Tensor operator+(const Tensor & a, double s) { return a - (-s); }
inline Tensor operator*(double s, const Tensor & a)
{
  Tensor r = a;
  for(int64_t i = 0; i < a.numel(); i++) r.set(i, a.dtype == kFloat32 ? (double)((float)s * (float)a.at(i)) : s * a.at(i));
  return r;
}
inline Tensor operator*(const Tensor & a, double s) { return s * a; }

// ---- the torch:: / at:: free functions the caller uses on these tensors
namespace torchlike
{
using Tensor = smplpp::Tensor;
using Device = smplpp::Device;
using DeviceType = smplpp::DeviceType;
using ScalarType = smplpp::ScalarType;
using smplpp::kCPU;
using smplpp::kCUDA;
using smplpp::kFloat32;
using smplpp::kFloat64;
using smplpp::kInt32;
using smplpp::kInt64;
inline Tensor zeros(std::initializer_list<int64_t> s) { return Tensor(std::vector<int64_t>(s), 0.0f); }
inline Tensor empty(std::initializer_list<int64_t> s) { return Tensor(std::vector<int64_t>(s), 0.0f); }
inline Tensor tensor(std::initializer_list<float> v)
{
  Tensor t({(int64_t)v.size()});
  size_t i = 0;
  for(float x : v) t.data[i++] = x;
  return t;
}
// torch::from_blob(ptr, sizes).clone() (toolbox/TorchEigenUtils.hpp:26-30, toTorchTensor): a copy of caller memory
inline Tensor from_blob(const float * p, std::initializer_list<int64_t> s)
{
  Tensor t{std::vector<int64_t>(s), 0.0f};
  for(int64_t i = 0; i < t.numel(); i++) t.data[(size_t)i] = p[i];
  return t;
}
inline Tensor from_blob(const double * p, std::initializer_list<int64_t> s)
{
  Tensor t(std::vector<int64_t>(s), kFloat64);
  for(int64_t i = 0; i < t.numel(); i++) t.ddata[(size_t)i] = p[i];
  return t;
}
// [m,k] x [k] -> [m] and [m,k] x [k,n] -> [m,n], fp32 (node/node.cpp:958: tangents_ [3,2] x phi [2])
inline Tensor matmul(const Tensor & a, const Tensor & b)
{
  if(a.dim() != 2 || (b.dim() != 1 && b.dim() != 2) || a.size(1) != b.size(0)) throw Exception("Tensor", "matmul: shapes");
  const int64_t m = a.size(0), k = a.size(1), n = b.dim() == 2 ? b.size(1) : 1;
  Tensor r(b.dim() == 2 ? std::vector<int64_t>{m, n} : std::vector<int64_t>{m});
  for(int64_t i = 0; i < m; i++)
    for(int64_t j = 0; j < n; j++)
    {
      float s = 0.0f;
      for(int64_t q = 0; q < k; q++) s += (float)a.at(i * k + q) * (float)b.at(q * n + j);
      r.data[(size_t)(i * n + j)] = s;
    }
  return r;
}
namespace at
{
inline Tensor dot(const Tensor & a, const Tensor & b)
{
  if(a.numel() != b.numel()) throw Exception("Tensor", "dot: element counts differ");
  float s = 0.0f;
  for(int64_t i = 0; i < a.numel(); i++) s += (float)a.at(i) * (float)b.at(i);
  return Tensor::scalar(s);
}
namespace indexing
{
using Slice = smplpp::Slice;
}
} // namespace at
} // namespace torchlike
} // namespace smplpp
#endif
