/* The model handle of libvidtok_amd.so from plain C: vidtok_kl_causal_488_4chn, encode -> KL mode -> decode of one synthetic
 * clip, with pseudo-random weights (a real host reads the reference checkpoint key by key instead).  With a fourth argument
 * t_chunk_enc: vidtok_kl_causal_488_16chn_v1_1 run with TEMPORAL TILING (chunks of t_chunk_enc frames, decoder look-ahead --
 * BASELINE.json configs[4]'s protocol: ./examples/roundtrip 129 256 256 16) next to the one-pass result of the same clip.
 *   cc -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/roundtrip.c -o examples/roundtrip \
 *      -L vidtok_amd -lvidtok_amd -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/vidtok_amd -Wl,-rpath,/opt/rocm/lib
 *   ./examples/roundtrip [T H W [t_chunk_enc]]        (default 17 128 128; needs an MI355X) */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vidtok_amd.h"

#define CHECK(expr)                                                        \
  do {                                                                     \
    if ((expr) != 0) {                                                     \
      fprintf(stderr, "%s failed: %s\n", #expr, vt_last_error());          \
      return 1;                                                            \
    }                                                                      \
  } while (0)
#define HIP(expr)                                                          \
  do {                                                                     \
    hipError_t e_ = (expr);                                                \
    if (e_ != hipSuccess) {                                                \
      fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_));           \
      return 1;                                                            \
    }                                                                      \
  } while (0)

static unsigned long long rng = 0x9E3779B97F4A7C15ull;
static float uniform(void) {                       /* [-1, 1) */
  rng = rng * 6364136223846793005ull + 1442695040888963407ull;
  return (float)((rng >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

int main(int argc, char** argv) {
  const int B = 1, T = argc > 3 ? atoi(argv[1]) : 17, H = argc > 3 ? atoi(argv[2]) : 128, W = argc > 3 ? atoi(argv[3]) : 128;
  const int t_chunk = argc > 4 ? atoi(argv[4]) : 0;                 /* > 0: the v1.1 model, tiled */
  vt_model_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.ch = 128; cfg.num_res_blocks = 2; cfg.in_channels = 3; cfg.out_ch = 3; cfg.z_channels = t_chunk > 0 ? 16 : 4; cfg.double_z = 1;
  if (t_chunk > 0) { cfg.version = 1; cfg.interpolation_mode = 1; }  /* v1.1: replicate padding, trilinear time up-sampling */
  cfg.num_resolutions = 4;
  const int mult[4] = {1, 2, 4, 4};
  memcpy(cfg.ch_mult, mult, sizeof mult);
  cfg.n_spatial_ds = 3; cfg.spatial_ds[0] = 0; cfg.spatial_ds[1] = 1; cfg.spatial_ds[2] = 2;
  cfg.n_tempo_ds = 2; cfg.tempo_ds[0] = 2; cfg.tempo_ds[1] = 1;
  cfg.n_spatial_us = 3; cfg.spatial_us[0] = 1; cfg.spatial_us[1] = 2; cfg.spatial_us[2] = 3;
  cfg.n_tempo_us = 2; cfg.tempo_us[0] = 1; cfg.tempo_us[1] = 2;
  cfg.time_downsample_factor = 4;
  cfg.regularizer = 0;
  if (vt_model_config_size() != (int)sizeof cfg) {
    fprintf(stderr, "header / library mismatch\n");
    return 1;
  }
  vt_model* m = NULL;
  CHECK(vt_create(&cfg, VT_BF16, &m));
  long long nparam = 0;
  for (int i = 0; i < vt_weight_count(m); ++i) {
    int64_t shape[5];
    int32_t nd;
    CHECK(vt_weight_shape(m, i, shape, &nd));
    long long n = 1, fan = 1;
    for (int k = 0; k < nd; ++k) n *= shape[k];
    for (int k = 1; k < nd; ++k) fan *= shape[k];
    float* w = (float*)malloc((size_t)n * sizeof(float));
    const char* key = vt_weight_name(m, i);
    const int is_norm_w = strstr(key, ".norm.weight") != NULL, is_vec = nd == 1;
    for (long long j = 0; j < n; ++j) w[j] = is_norm_w ? 1.0f + 0.1f * uniform() : (is_vec ? 0.05f * uniform() : uniform() * sqrtf(3.0f / (float)fan));
    CHECK(vt_load_weight(m, key, w, shape, nd));
    free(w);
    nparam += n;
  }
  int32_t ld[4];
  CHECK(vt_latent_dims(m, T, H, W, ld));
  int64_t ws_bytes = vt_workspace_bytes(m, B, T, H, W);
  if (t_chunk > 0) {
    const int64_t tb = vt_tile_workspace_bytes(m, B, T, H, W, t_chunk, 1);
    if (tb < 0) { fprintf(stderr, "vt_tile_workspace_bytes: %s\n", vt_last_error()); return 1; }
    if (tb > ws_bytes) ws_bytes = tb;
  }
  const size_t nx = (size_t)B * 3 * T * H * W, nh = (size_t)B * ld[0] * ld[1] * ld[2] * ld[3], nz = nh / 2;
  printf("%d parameters tensors, %lld values; latent %d x %d x %d x %d; workspace %.1f MiB\n", vt_weight_count(m), nparam, ld[0], ld[1], ld[2],
         ld[3], (double)ws_bytes / (1 << 20));
  float* x_host = (float*)malloc(nx * sizeof(float));
  for (size_t i = 0; i < nx; ++i) x_host[i] = uniform();
  float *x, *h, *z, *kl, *xh;
  void* ws;
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));
  HIP(hipMalloc((void**)&x, nx * 4)); HIP(hipMalloc((void**)&h, nh * 4)); HIP(hipMalloc((void**)&z, nz * 4));
  const size_t nxh = t_chunk > 0 ? (size_t)B * 3 * ld[1] * cfg.time_downsample_factor * H * W : nx;   /* v1.1 decodes the front padding too */
  HIP(hipMalloc((void**)&kl, 4)); HIP(hipMalloc((void**)&xh, nxh * 4)); HIP(hipMalloc(&ws, (size_t)ws_bytes));
  HIP(hipMemcpy(x, x_host, nx * 4, hipMemcpyHostToDevice));
  for (int pass = 0; pass < 2; ++pass) {           /* the second pass runs allocation-free on the same workspace */
    CHECK(vt_encode(m, x, B, T, H, W, h, ws, ws_bytes, stream));
    CHECK(vt_regularize_kl(m, h, NULL, z, kl, B, ld[1], ld[2], ld[3], stream));
    CHECK(vt_decode(m, z, B, ld[1], ld[2], ld[3], xh, ws, ws_bytes, stream));
  }
  HIP(hipStreamSynchronize(stream));
  if (t_chunk > 0) {
    /* the same clip as chunks [0,1), [1,1+c), ...: the handle keeps every module's causal cache between the chunks */
    const int f = cfg.time_downsample_factor, tz = vt_tile_latent_frames(m, T, t_chunk);
    const size_t nht = (size_t)B * ld[0] * tz * ld[2] * ld[3], nxt = (size_t)B * 3 * tz * f * H * W;
    float *ht, *zt, *xt;
    HIP(hipMalloc((void**)&ht, nht * 4)); HIP(hipMalloc((void**)&zt, nht * 2)); HIP(hipMalloc((void**)&xt, nxt * 4));
    CHECK(vt_tile_encode(m, x, B, T, H, W, t_chunk, ht, ws, ws_bytes, stream));
    CHECK(vt_regularize_kl(m, ht, NULL, zt, kl, B, tz, ld[2], ld[3], stream));
    CHECK(vt_tile_decode(m, zt, B, tz, ld[2], ld[3], t_chunk / f, 1, xt, ws, ws_bytes, stream));
    HIP(hipStreamSynchronize(stream));
    /* both passes end with the clip's T frames: the tiled one decodes tz * f >= T frames, the one-pass ld[1] * f */
    float* a = (float*)malloc((size_t)T * H * W * sizeof(float));
    float* b = (float*)malloc((size_t)T * H * W * sizeof(float));
    double num = 0.0, den = 0.0;
    for (int c = 0; c < 3; ++c) {
      HIP(hipMemcpy(a, xt + ((size_t)c * tz * f + (tz * f - T)) * H * W, (size_t)T * H * W * 4, hipMemcpyDeviceToHost));
      HIP(hipMemcpy(b, xh + ((size_t)c * ld[1] * f + (ld[1] * f - T)) * H * W, (size_t)T * H * W * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < (size_t)T * H * W; ++i) {
        const double d = fabs((double)a[i] - b[i]);
        if (d > num) num = d;
        if (fabs((double)b[i]) > den) den = fabs((double)b[i]);
      }
    }
    printf("tiled (t_chunk_enc %d, look-ahead) vs one pass: %d latent frames, max |diff| / max |one pass| = %.3e\n", t_chunk, tz, num / den);
    CHECK(vt_reset_cache(m));
    if (!(num / den < 0.1)) return 3;
  }
  float* out = (float*)malloc(nx * sizeof(float));
  float klv;
  HIP(hipMemcpy(out, xh, nx * 4, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(&klv, kl, 4, hipMemcpyDeviceToHost));
  double sum = 0.0, sq = 0.0;
  int finite = 1;
  for (size_t i = 0; i < nx; ++i) {
    finite = finite && isfinite(out[i]);
    sum += out[i];
    sq += (double)out[i] * out[i];
  }
  printf("reconstruction [%d, 3, %d, %d, %d]: finite %d, mean %.6f, rms %.6f, kl %.4f\n", B, T, H, W, finite, sum / (double)nx, sqrt(sq / (double)nx), klv);
  CHECK(vt_destroy(m));
  return finite ? 0 : 2;
}
