// kernels.h -- launchers of the non-MFMA kernels (sean_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace chk {
hipError_t label_downsample(const uint8_t* in, uint8_t* out, int B, int S, int r, hipStream_t s);
hipError_t onehot_conv3x3(const uint8_t* lab, const float* table, const float* bias, float* out, int B, int H, int W,
                          int K, int relu, hipStream_t s);
hipError_t fc_mu(const float* codes, const float* Wt, const float* bias, float* mu_img, int B, int Npad, hipStream_t s);
hipError_t conv_img_tanh(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int H, int W,
                         hipStream_t s);
hipError_t gen_noise(float* out, long long n, uint64_t seed, hipStream_t s);
}  // namespace chk
