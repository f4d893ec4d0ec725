// Implicit-GEMM convolution kernel (conv_igemm_kernel.h), fp32 storage + v_mfma_f32_32x32x2_f32: the instantiations of vt_dtype VT_F32.
#include "conv_igemm_kernel.h"

extern "C" __attribute__((visibility("hidden"))) int vt_igemm_dispatch_f32(const void* args, int nbatch, void* stream) {
  return dispatch_tile<float, float>(*reinterpret_cast<const ConvArgs*>(args), nbatch, reinterpret_cast<hipStream_t>(stream));
}
