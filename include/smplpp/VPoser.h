// smplpp::VPoserDecoder (reference include/smplpp/VPoser.h:53-90, src/VPoser.cpp:129-238) and
// smplpp::convertRotMatToAxisAngle (VPoser.h:18, src/VPoser.cpp:25-120) over the C ABI — header-only C++ shim.
// The decoder is Linear(32,512) LeakyReLU Dropout(eval) Linear(512,512) LeakyReLU Linear(512,126) -> 6D -> rotation ->
// axis-angle; `forward` keeps the reference's name and meaning.  The gradient autograd supplies in node/node.cpp:761-772
// is returned explicitly (`forward(latent, &jac)`), and an IkSolver given the decoder runs it inside the loop.
#ifndef SMPLPP_SHIM_VPOSER_H
#define SMPLPP_SHIM_VPOSER_H

#include "SMPL.h"

namespace smplpp
{
// rotMat [N,3,3] -> axis-angle [N,3] (src/VPoser.cpp:25-120)
inline Tensor convertRotMatToAxisAngle(const Tensor & rotMat, int device = 0)
{
  if(rotMat.numel() % 9 != 0) throw Exception("VPoser", "convertRotMatToAxisAngle: rotMat must be [N,3,3]");
  const int64_t n = rotMat.numel() / 9;
  Tensor aa({n, 3});
  check(smplpp_rotmat_to_axis_angle(device, n, rotMat.ptr(), aa.ptr(), SMPLPP_HOST, nullptr), "VPoser");
  return aa;
}

class VPoserDecoder
{
public:
  explicit VPoserDecoder(int device = 0) : device_(device) {}
  // the reference's VPoserDecoder is a torch module HOLDER: node/node.cpp:425-439 writes `smplpp::VPoserDecoder vposer;
  // vposer->loadParamsFromJson(path); vposer->eval(); vposer->to(*device);`
  VPoserDecoder * operator->() { return this; }
  void eval() {} // (dropout is the identity in this decoder: src/VPoser.cpp:146-155 in eval mode)
  void to(const Device & device)
  {
    if(v_ && device.index != device_) throw Exception("VPoser", "VPoserDecoder::to: move the decoder before its parameters are loaded");
    device_ = device.index < 0 ? 0 : device.index;
  }
  ~VPoserDecoder() { smplpp_vposer_destroy(v_); }
  VPoserDecoder(const VPoserDecoder &) = delete;
  VPoserDecoder & operator=(const VPoserDecoder &) = delete;

  //! Dimension of hidden variables / number of joints (VPoser.h:79-82)
  const int64_t hiddenDim_ = 512;
  const int64_t jointNum_ = 21;

  // VPoserDecoderImpl::loadParamsFromJson (src/VPoser.cpp:169-238): the file scripts/preprocess_vposer.py:42-52 writes,
  // torch::nn::Linear layout [out,in]
  void loadParamsFromJson(const std::string & jsonPath)
  {
    detail::JsonArrays j(jsonPath);
    static const char * keys[6] = {"decoder_net.0.weight", "decoder_net.0.bias", "decoder_net.3.weight",
                                   "decoder_net.3.bias",   "decoder_net.5.weight", "decoder_net.5.bias"};
    static const int64_t sizes[6] = {512 * 32, 512, 512 * 512, 512, 126 * 512, 126};
    std::vector<float> p[6];
    for(int i = 0; i < 6; i++)
    {
      auto it = j.values.find(keys[i]);
      if(it == j.values.end() || (int64_t)it->second.size() != sizes[i])
        throw Exception("VPoser", std::string("VPoser json lacks ") + keys[i] + " (or it has the wrong size)");
      p[i].assign(it->second.begin(), it->second.end());
    }
    setParams(p[0].data(), p[1].data(), p[2].data(), p[3].data(), p[4].data(), p[5].data());
  }
  void setParams(const float * w0, const float * b0, const float * w1, const float * b1, const float * w2, const float * b2)
  {
    smplpp_vposer_destroy(v_);
    v_ = nullptr;
    check(smplpp_vposer_create(device_, w0, b0, w1, b1, w2, b2, &v_), "VPoser");
  }
  // VPoserDecoderImpl::forward (src/VPoser.cpp:163-167): latent [B,32] -> joint angles [B,21,3];
  // jac (optional) [B,63,32] = d(angles)/d(latent)
  Tensor forward(const Tensor & latent, Tensor * jac = nullptr)
  {
    if(!v_) throw Exception("VPoser", "VPoserDecoder: parameters not loaded");
    if(latent.numel() % LATENT_DIM != 0) throw Exception("VPoser", "forward: latent must be [B,32]");
    const int64_t n = latent.numel() / LATENT_DIM;
    Tensor out({n, jointNum_, 3});
    if(jac) *jac = Tensor({n, 3 * jointNum_, LATENT_DIM});
    check(smplpp_vposer_forward(v_, n, latent.ptr(), out.ptr(), jac ? jac->ptr() : nullptr, SMPLPP_HOST, nullptr), "VPoser");
    return out;
  }
  smplpp_vposer * handle() const { return v_; }

private:
  int device_;
  smplpp_vposer * v_ = nullptr;
};
} // namespace smplpp
#endif
