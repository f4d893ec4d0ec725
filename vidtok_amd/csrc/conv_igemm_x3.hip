// Implicit-GEMM convolution kernel (conv_igemm_kernel.h), split-bf16 arithmetic on fp32 storage: the instantiations of vt_dtype VT_BF16X3.
#include "conv_igemm_kernel.h"

extern "C" __attribute__((visibility("hidden"))) int vt_igemm_dispatch_x3(const void* args, int nbatch, void* stream) {
  return dispatch_tile<split3_t, float>(*reinterpret_cast<const ConvArgs*>(args), nbatch, reinterpret_cast<hipStream_t>(stream));
}
