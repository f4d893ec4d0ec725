// Implicit-GEMM convolution kernel (conv_igemm_kernel.h), bf16 storage + v_mfma_f32_32x32x16_bf16: the instantiations of vt_dtype VT_BF16
// (results in bf16, or in fp32: NCTHW outputs, attention scores, split-K partials).
#include "conv_igemm_kernel.h"

extern "C" __attribute__((visibility("hidden"))) int vt_igemm_dispatch_bf16(const void* args, int nbatch, int out_f32, void* stream) {
  const ConvArgs& a = *reinterpret_cast<const ConvArgs*>(args);
  return out_f32 ? dispatch_tile<bf16_t, float>(a, nbatch, reinterpret_cast<hipStream_t>(stream))
                 : dispatch_tile<bf16_t, bf16_t>(a, nbatch, reinterpret_cast<hipStream_t>(stream));
}
