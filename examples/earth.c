/*
 * earth.c — (image-texture url) through the C ABI, from plain C: reads a picture FILE (JPEG, PNG, GIF or BMP: told
 * apart by content, as the reference's ImageIO/read tells them apart, texture.clj:72), has the library decode it
 * (rt_image_decode), wraps it round the reference's earth world (scenes.clj:39-42: one sphere of radius 2 at the
 * origin, Lambertian over the image) and renders that with the scene table's defaults (core.clj:60-65: lookfrom
 * [13 2 3], lookat the origin, vfov 20, background [0.7 0.8 1.0]).
 *
 *   earth picture-file [width height spp out.ppm [argb.bin]]     defaults: 300 200 32 earth.ppm
 *
 * Build: make -C raytracing_amd/csrc example
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtmi.h"

static void unit(double v[3]) { /* base.clj:39-40 */
  double k = 1.0 / sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  v[0] *= k; v[1] *= k; v[2] *= k;
}

/* camera.clj:7-33 */
static void camera(const double from[3], const double at[3], const double vup[3], double vfov, double aspect,
                   double aperture, double focus, double t0, double t1, rt_camera* c) {
  double theta = vfov * 0.017453292519943295, h = tan(theta * 0.5);
  double vh = h * 2.0, vw = aspect * vh;
  double w[3] = {from[0] - at[0], from[1] - at[1], from[2] - at[2]};
  unit(w);
  double u[3] = {vup[1] * w[2] - vup[2] * w[1], vup[2] * w[0] - vup[0] * w[2], vup[0] * w[1] - vup[1] * w[0]};
  unit(u);
  double v[3] = {w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0]};
  for (int a = 0; a < 3; ++a) {
    c->origin[a] = from[a];
    c->horizontal[a] = u[a] * (focus * vw);
    c->vertical[a] = v[a] * (focus * vh);
    c->lower_left_corner[a] = ((from[a] - c->horizontal[a] * 0.5) - c->vertical[a] * 0.5) - w[a] * focus;
    c->u[a] = u[a]; c->v[a] = v[a]; c->w[a] = w[a];
  }
  c->lens_radius = aperture / 2;
  c->time0 = t0; c->time1 = t1;
}

#define CHECK(call)                                                      \
  do {                                                                   \
    int rc_ = (call);                                                    \
    if (rc_ != RT_OK) {                                                  \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rt_last_error());    \
      return 1;                                                          \
    }                                                                    \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: earth picture-file [width height spp out.ppm [argb.bin]]\n"); return 2; }
  int W = argc > 2 ? atoi(argv[2]) : 300, H = argc > 3 ? atoi(argv[3]) : 200, spp = argc > 4 ? atoi(argv[4]) : 32;
  const char* out = argc > 5 ? argv[5] : "earth.ppm";
  const char* raw = argc > 6 ? argv[6] : NULL;

  /* the file as it is; the library decodes it */
  FILE* pf = fopen(argv[1], "rb");
  if (!pf) { perror(argv[1]); return 1; }
  fseek(pf, 0, SEEK_END);
  long n_file = ftell(pf);
  fseek(pf, 0, SEEK_SET);
  uint8_t* file = (uint8_t*)malloc((size_t)n_file + 1);
  if (fread(file, 1, (size_t)n_file, pf) != (size_t)n_file) { perror(argv[1]); return 1; }
  fclose(pf);
  int32_t iw = 0, ih = 0;
  CHECK(rt_image_decode(file, (uint64_t)n_file, &iw, &ih, NULL, 0));
  uint8_t* texels = (uint8_t*)malloc((size_t)iw * (size_t)ih * 3);
  CHECK(rt_image_decode(file, (uint64_t)n_file, &iw, &ih, texels, (uint64_t)iw * (uint64_t)ih * 3));
  free(file);
  fprintf(stderr, "%s: %d x %d texels\n", argv[1], iw, ih);

  /* scenes.clj:39-42 */
  int32_t prim_type[1] = {RT_PRIM_SPHERE}, prim_mat[1] = {0};
  double prim_d[RT_PRIM_STRIDE] = {0, 0, 0, 2};
  int32_t node_kind[2] = {RT_NODE_PRIM, RT_NODE_LIST}, node_i[2 * RT_NODE_ISTRIDE] = {0}, node_children[1] = {0};
  double node_d[2 * RT_NODE_DSTRIDE] = {0};
  node_i[1 * RT_NODE_ISTRIDE] = 0;      /* the list's first child slot */
  node_i[1 * RT_NODE_ISTRIDE + 1] = 1;  /* ... and how many */
  int32_t mat_type[1] = {RT_MAT_LAMBERTIAN}, mat_tex[1] = {0};
  double mat_d[RT_MAT_DSTRIDE] = {0};
  int32_t tex_type[1] = {RT_TEX_IMAGE}, tex_i[RT_TEX_ISTRIDE] = {0};  /* image 0 */
  double tex_d[RT_TEX_DSTRIDE] = {0};
  const uint8_t* images[1] = {texels};

  rt_scene_desc d;
  memset(&d, 0, sizeof d);
  d.abi_version = RTMI_ABI_VERSION;
  d.n_prims = 1; d.prim_type = prim_type; d.prim_d = prim_d; d.prim_mat = prim_mat;
  d.n_nodes = 2; d.node_kind = node_kind; d.node_i = node_i; d.node_d = node_d;
  d.n_children = 1; d.node_children = node_children; d.root = 1;
  d.n_mats = 1; d.mat_type = mat_type; d.mat_d = mat_d; d.mat_tex = mat_tex;
  d.n_tex = 1; d.tex_type = tex_type; d.tex_d = tex_d; d.tex_i = tex_i;
  d.n_images = 1; d.img_w = &iw; d.img_h = &ih; d.img_rgb8 = images;

  rt_camera cam;
  const double from[3] = {13, 2, 3}, at[3] = {0, 0, 0}, vup[3] = {0, 1, 0};
  camera(from, at, vup, 20.0, (double)W / (double)H, 0.0, 10.0, 0.0, 1.0, &cam);

  rt_params p;
  memset(&p, 0, sizeof p);
  p.width = W; p.height = H; p.spp = spp; p.max_depth = 50; p.shard_count = 1;
  p.background[0] = 0.70; p.background[1] = 0.80; p.background[2] = 1.0;
  p.seed = 0x5EED0001ull;

  rt_ctx* ctx = NULL;
  rt_scene* scene = NULL;
  CHECK(rt_init(0, &ctx));
  CHECK(rt_scene_create(ctx, &d, &scene));
  uint32_t* argb = (uint32_t*)malloc((size_t)W * H * 4);
  rt_stats st;
  CHECK(rt_render(ctx, scene, &cam, &p, NULL, argb, &st));
  fprintf(stderr, "%llu samples, %llu rays, kernel %.3f ms\n", (unsigned long long)st.samples, (unsigned long long)st.rays, st.kernel_ms);

  FILE* f = fopen(out, "wb");
  if (!f) { perror(out); return 1; }
  fprintf(f, "P6\n%d %d\n255\n", W, H);
  for (int y = H - 1; y >= 0; --y)
    for (int x = 0; x < W; ++x) {
      uint32_t c = argb[(size_t)y * W + x];
      unsigned char rgb[3] = {(unsigned char)(c >> 16), (unsigned char)(c >> 8), (unsigned char)c};
      fwrite(rgb, 1, 3, f);
    }
  fclose(f);
  if (raw) {
    f = fopen(raw, "wb");
    if (!f) { perror(raw); return 1; }
    fwrite(argb, 4, (size_t)W * H, f);
    fclose(f);
  }
  free(argb);
  free(texels);
  rt_scene_destroy(scene);
  rt_shutdown(ctx);
  return 0;
}
