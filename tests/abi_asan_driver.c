/* Host-runtime checks under AddressSanitizer (CPU only; built by `make -C pea_diffusion_amd/csrc asan`, run by
 * tests/test_abi_cpu.py).  No GPU: every call below must stay on the host -- argument validation, error strings, the
 * tape builder / memory planner (pea_unet_plan), communicator argument checks -- and must come back with the documented
 * PEA_E_* code instead of touching freed or out-of-bounds memory.  Test infrastructure, not product code. */
#include <stdio.h>
#include <string.h>

#include "pea_hip.h"

static int fails = 0;
#define EXPECT(cond)                                                     \
  do {                                                                   \
    if (!(cond)) { printf("FAIL %s:%d %s (last error: %s)\n", __FILE__, __LINE__, #cond, pea_last_error()); ++fails; } \
  } while (0)

static pea_unet_config sdxl(void) {
  pea_unet_config c;
  memset(&c, 0, sizeof(c));
  c.in_channels = 4; c.out_channels = 4; c.n_levels = 3;
  int bo[3] = {320, 640, 1280}, dc[3] = {0, 1, 1}, uc[3] = {1, 1, 0}, dp[3] = {1, 2, 10}, hd[3] = {5, 10, 20};
  for (int i = 0; i < 3; ++i) { c.block_out[i] = bo[i]; c.down_cross[i] = dc[i]; c.up_cross[i] = uc[i]; c.depth[i] = dp[i]; c.heads[i] = hd[i]; }
  c.layers_per_block = 2; c.cross_dim = 2048; c.linear_proj = 1; c.groups = 32; c.eps = 1e-5f;
  c.text_time = 1; c.add_time_dim = 256; c.proj_in_dim = 2816;
  return c;
}

int main(void) {
  EXPECT(pea_version() >= 100);
  /* ---- operator-level shape errors are rejected before any device work */
  EXPECT(pea_op_gemm(0, 8, 0, 8, 0, 8, 4, 4, 10, 1.0f, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0) == PEA_E_SHAPE);
  EXPECT(strstr(pea_last_error(), "K=10") != NULL);
  EXPECT(pea_op_attention_fwd(0, 64, 0, 64, 0, 64, 0, 64, 0, 1, 1, 0, 0, 1.0f, 1, 0) == PEA_E_SHAPE);
  EXPECT(pea_op_attention_fwd(0, 64, 0, 64, 0, 64, 0, 64, 0, 1, 1, 64, 64, 1.0f, 7, 0) == PEA_E_SHAPE);
  /* ---- handles: NULL everywhere */
  EXPECT(pea_unet_forward(0, 0, 0, 0, 0, 0, 0, 0, 0, 0) != PEA_OK);
  EXPECT(pea_unet_load_weight(0, "x", 0, 0, 0) != PEA_OK);
  EXPECT(pea_train_step(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1.0f, 0, 0, 0, 0) != PEA_OK);
  EXPECT(pea_adapter_forward(0, 0, 0, 0, 0, 0) != PEA_OK);
  EXPECT(pea_unet_num_taps(0) == 0);
  EXPECT(pea_allreduce_grads(0, 0, 4, 0) == PEA_E_INVALID);
  EXPECT(pea_comm_join(0, 0) == PEA_E_INVALID);
  { void* h = 0; char id[128]; memset(id, 0, sizeof(id)); EXPECT(pea_comm_init(3, 2, id, &h) == PEA_E_SHAPE && h == 0); }
  { void* h = 0; pea_unet_config c = sdxl(); EXPECT(pea_unet_create(&c, 1, 128, 128, 77, 1, 1, &h) == PEA_E_HIP && h == 0); }
  /* ---- the host-side graph builder + memory planner: SDXL known answers */
  {
    pea_unet_config c = sdxl();
    int n_ops = 0, n_w = 0; long long np = 0, wb = 0, ab = 0, gb = 0;
    EXPECT(pea_unet_plan(&c, 4, 128, 128, 77, 1, &n_ops, &n_w, &np, &wb, &ab, &gb) == PEA_OK);
    EXPECT(np == 2567463684LL);
    EXPECT(n_ops > 800 && n_w > 1600 && wb > 5000000000LL && ab > 15000000000LL && gb > 10000000000LL);
    printf("sdxl plan: %d ops, %d weight tensors, %lld params, weights %.2f GB, activations %.2f GB, gradients %.2f GB\n", n_ops,
           n_w, np, wb / 1e9, ab / 1e9, gb / 1e9);
    /* SSD-1B layout: per-position depths, reverse list, no mid block */
    c.per_layer_depth = 1; c.depth_mid = -1;
    int dn[3][2] = {{1, 1}, {2, 2}, {4, 4}}, up[3][3] = {{4, 4, 10}, {2, 1, 1}, {1, 1, 1}};
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 2; ++j) c.depth_down[i][j] = dn[i][j]; for (int j = 0; j < 3; ++j) c.depth_up[i][j] = up[i][j]; }
    EXPECT(pea_unet_plan(&c, 4, 128, 128, 77, 1, &n_ops, &n_w, &np, &wb, &ab, &gb) == PEA_OK);
    EXPECT(np == 1300195844LL);
    /* invalid configs come back as PEA_E_SHAPE with a message, never as a crash */
    pea_unet_config bad = sdxl(); bad.block_out[1] = 100;
    EXPECT(pea_unet_plan(&bad, 1, 64, 64, 77, 0, 0, 0, 0, 0, 0, 0) == PEA_E_SHAPE);
    bad = sdxl(); bad.n_levels = 7;
    EXPECT(pea_unet_plan(&bad, 1, 64, 64, 77, 0, 0, 0, 0, 0, 0, 0) == PEA_E_SHAPE);
    bad = sdxl(); bad.heads[2] = 7;
    EXPECT(pea_unet_plan(&bad, 1, 64, 64, 77, 0, 0, 0, 0, 0, 0, 0) == PEA_E_SHAPE);
    bad = sdxl();
    EXPECT(pea_unet_plan(&bad, 1, 63, 64, 77, 0, 0, 0, 0, 0, 0, 0) == PEA_E_SHAPE);
    bad = sdxl(); bad.layers_per_block = 9;
    EXPECT(pea_unet_plan(&bad, 1, 64, 64, 77, 0, 0, 0, 0, 0, 0, 0) == PEA_E_SHAPE);
  }
  printf(fails ? "asan driver: %d FAILED\n" : "asan driver: all host checks passed\n", fails);
  return fails ? 1 : 0;
}
