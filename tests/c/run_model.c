/* A caller WITHOUT Python: load a model file written by coma_amd (sd_model_save), feed raw input files, run the network through
 * the C ABI of libcoma_hip.so and write the raw output.  Built and run by tests/test_sd_model_gpu.py:
 *     gcc tests/c/run_model.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Lcoma_amd -lcoma_hip -L/opt/rocm/lib -lamdhip64 \
 *         -Wl,-rpath,$PWD/coma_amd -Wl,-rpath,/opt/rocm/lib -o run_model
 *     run_model unet   model.sdm ctx.bin x_in.bin timesteps.bin eps_out.bin
 *     run_model decode model.sdm z.bin image_out.bin
 *     run_model encode model.sdm x.bin moments_out.bin
 * This is the binding a non-Python host of the reference's pipeline would write against include/sd_hip.h. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "coma_hip.h"
#include "sd_hip.h"

#define CHECK(x)                                                                      \
  do {                                                                                \
    if ((x) != 0) { fprintf(stderr, "%s failed: %s\n", #x, coma_last_error()); return 1; } \
  } while (0)

static void* dev_from_file(const char* path, size_t want) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  void* h = malloc(want);
  if (fread(h, 1, want, f) != want) { fprintf(stderr, "%s: expected %zu bytes\n", path, want); exit(2); }
  fclose(f);
  void* d = NULL;
  if (hipMalloc(&d, want) != hipSuccess || hipMemcpy(d, h, want, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "device copy failed\n"); exit(2); }
  free(h);
  return d;
}

static int dev_to_file(const void* d, size_t n, const char* path) {
  void* h = malloc(n);
  if (hipMemcpy(h, d, n, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  FILE* f = fopen(path, "wb");
  if (!f || fwrite(h, 1, n, f) != n) return 1;
  fclose(f);
  free(h);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: run_model unet|decode|encode model inputs... output\n"); return 2; }
  void* m = NULL;
  hipStream_t s;
  if (hipStreamCreate(&s) != hipSuccess) return 1;
  CHECK(sd_model_load(argv[2], &m));
  size_t n_in = 0, n_t = 0, n_ctx = 0, n_out = 0;
  void* p = NULL;
  if (!strcmp(argv[1], "unet") && argc == 7) {
    CHECK(sd_model_binding(m, "ctx", &p, &n_ctx));
    CHECK(sd_model_binding(m, "x_in", &p, &n_in));
    CHECK(sd_model_binding(m, "timesteps", &p, &n_t));
    CHECK(sd_model_binding(m, "eps", &p, &n_out));
    void* ctx = dev_from_file(argv[3], n_ctx);
    void* x = dev_from_file(argv[4], n_in);
    void* t = dev_from_file(argv[5], n_t);
    void* out = NULL;
    if (hipMalloc(&out, n_out) != hipSuccess) return 1;
    CHECK(sd_unet_set_context(m, ctx, s));
    for (int rep = 0; rep < 2; ++rep) CHECK(sd_unet_forward(m, x, (const float*)t, out, s));   /* second call = graph replay */
    if (hipStreamSynchronize(s) != hipSuccess || dev_to_file(out, n_out, argv[6])) return 1;
    printf("unet: %d launches per step, %zu output bytes\n", sd_model_num_launches(m, "step"), n_out);
  } else if ((!strcmp(argv[1], "decode") || !strcmp(argv[1], "encode")) && argc == 5) {
    const int dec = !strcmp(argv[1], "decode");
    CHECK(sd_model_binding(m, dec ? "z" : "x", &p, &n_in));
    CHECK(sd_model_binding(m, dec ? "image" : "moments", &p, &n_out));
    void* x = dev_from_file(argv[3], n_in);
    void* out = NULL;
    if (hipMalloc(&out, n_out) != hipSuccess) return 1;
    CHECK(dec ? sd_vae_decode(m, x, out, s) : sd_vae_encode(m, x, out, s));
    if (hipStreamSynchronize(s) != hipSuccess || dev_to_file(out, n_out, argv[4])) return 1;
    printf("%s: %d launches, %zu output bytes\n", argv[1], sd_model_num_launches(m, argv[1]), n_out);
  } else {
    fprintf(stderr, "bad arguments\n");
    return 2;
  }
  CHECK(sd_model_destroy(m));
  return 0;
}
