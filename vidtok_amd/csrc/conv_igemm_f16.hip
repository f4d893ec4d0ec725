// Implicit-GEMM convolution kernel (conv_igemm_kernel.h), fp16 storage + v_mfma_f32_32x32x16_f16: the instantiations of vt_dtype VT_F16
// (what the reference computes in under its README's torch.autocast(dtype=torch.float16); results in fp16, or in fp32: NCTHW outputs,
// attention scores, split-K partials).
#include "conv_igemm_kernel.h"

extern "C" __attribute__((visibility("hidden"))) int vt_igemm_dispatch_f16(const void* args, int nbatch, int out_f32, void* stream) {
  const ConvArgs& a = *reinterpret_cast<const ConvArgs*>(args);
  return out_f32 ? dispatch_tile<f16_t, float>(a, nbatch, reinterpret_cast<hipStream_t>(stream))
                 : dispatch_tile<f16_t, f16_t>(a, nbatch, reinterpret_cast<hipStream_t>(stream));
}
