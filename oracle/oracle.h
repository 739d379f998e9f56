/*
 * CPU oracle for the PBR frame path and the IBL precompute of maoz233/wgpu-renderer.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar f32 restatement of the reference's WGSL
 * shaders and of the WebGPU fixed-function rules they run under.  Only tests/, the smoke
 * check and the cpu_baseline leg of bench.py may load it; the product library never does.
 *
 * Parity status: the reference ships no tests, golden vectors or fixtures and its shaders
 * need a WebGPU implementation that this build cannot run, so the oracle is pinned only by
 * known-answer values derived by hand from the WGSL (tests/test_oracle_kat.py).  PARITY
 * UNPINNED against an executed reference.
 *
 * All paths cited are relative to the reference checkout.
 */
#ifndef PBR_ORACLE_H
#define PBR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- matrix / camera helpers (wgpu-matrix 3.3.0 conventions, column-major f32) ---- */
void orc_mat4_perspective(float fovy, float aspect, float z_near, float z_far, float* out16);
void orc_mat4_look_at(const float* eye3, const float* target3, const float* up3, float* out16);
void orc_mat4_mul(const float* a16, const float* b16, float* out16);
int  orc_mat4_inverse(const float* m16, float* out16);
void orc_mat4_transpose(const float* m16, float* out16);
void orc_mat4_from_quat(const float* q4, float* out16);
void orc_mat4_scale(const float* m16, const float* v3, float* out16);
void orc_mat4_translate(const float* m16, const float* v3, float* out16);
/* glTF node → model matrix, glTF.ts:112-130 (I·scale·translate·rotation). Any pointer may be NULL. */
void orc_node_matrix(const float* scale3, const float* translation3, const float* rotation4,
                     const float* matrix16, float* out16);

/* Orbit camera state, camera.ts:38-42. */
typedef struct OrcCamera {
  float eye[3];
  float target[3];
  float up[3];
  float right[3];
  float distance;
} OrcCamera;
void orc_camera_init(OrcCamera* cam);
void orc_camera_rotate(OrcCamera* cam, float dx, float dy);   /* camera.ts:163-183 */
void orc_camera_dolly(OrcCamera* cam, float dx, float dy);    /* camera.ts:185-189 */
void orc_camera_track(OrcCamera* cam, float dx, float dy);    /* camera.ts:191-202 */
void orc_camera_view(const OrcCamera* cam, float* out16);     /* camera.ts:72-74 */

/* ---- Radiance .hdr decode, hdr.ts:37-118.  Returns 0 or a negative error code. ---- */
int orc_hdr_dims(const uint8_t* bytes, int64_t n, int* w, int* h);
int orc_hdr_decode(const uint8_t* bytes, int64_t n, float* rgba_out);

/* ---- mip chains.  A chain stores level 0, 1, ... back to back; level l is
 *      max(1,w>>l) x max(1,h>>l) texels.  utils.ts:20-24 gives the level count. ---- */
int     orc_mip_level_count(int w, int h);
int64_t orc_chain_texels(int w, int h, int levels);
int64_t orc_chain_offset(int w, int h, int level);          /* in texels */
/* mipmap.wgsl:23-26 driven by renderer.ts:1209-1317: each level is one bilinear tap of the level above. */
void orc_build_mips_rgba32f(float* chain, int w, int h, int levels);
void orc_build_mips_rgba8(uint8_t* chain, int w, int h, int levels, int is_srgb);

/* ---- IBL precompute.  Cube outputs are [6][size][size][4] f32.  first/count select a
 *      range of the flattened (face,y,x) texel index so large maps can be spot-checked. ---- */
void orc_equirect_to_cube(const float* hdr, int w, int h, int size, float* out,
                          int64_t first, int64_t count);                       /* equirectangular.wgsl */
void orc_irradiance(const float* hdr, int w, int h, int size, float* out,
                    int64_t first, int64_t count);                             /* irradiancemap.wgsl */
void orc_prefilter(const float* hdr_chain, int w, int h, int levels, int size, float roughness,
                   float* out, int64_t first, int64_t count);                  /* prefiltermap.wgsl */
/* Test instrumentation (not part of the restated pipeline): per output texel the interval [lo, hi] of values reachable
 * when every nearest-pixel lookup whose direction lies within `ang` radians of a pixel boundary may take either pixel. */
void orc_equirect_to_cube_bounds(const float* hdr, int w, int h, int size, float ang, float* lo, float* hi, int64_t first, int64_t count);
void orc_irradiance_bounds(const float* hdr, int w, int h, int size, float ang, float* lo, float* hi, int64_t first, int64_t count);
void orc_prefilter_bounds(const float* hdr_chain, int w, int h, int levels, int size, float roughness, float ang, float* lo, float* hi,
                          int64_t first, int64_t count);
/* brdf.wgsl: out is [size][size][4] f32 before the rgba16float store; out_f16 (may be NULL)
 * receives the stored half bits.  Rows [row0,row0+rows). */
void orc_brdf_lut(int size, float* out_f32, uint16_t* out_f16, int row0, int rows);
float    orc_radical_inverse(uint32_t bits);                                   /* prefiltermap.wgsl:32-39 */
void     orc_irradiance_trip_counts(int* n_phi, int* n_theta);                 /* irradiancemap.wgsl:75-76 */
uint16_t orc_f32_to_f16(float f);
float    orc_f16_to_f32(uint16_t h);

/* ---- frame ---- */
typedef struct OrcFrameUniforms {
  float model[16];          /* shader.wgsl:14-19 Transform, filled as renderer.ts:1828-1861 */
  float normal[16];
  float view[16];
  float projection[16];
  float inv_view_proj[16];  /* skybox.wgsl:7, renderer.ts:1864-1875 */
  float view_pos[4];        /* shader.wgsl:22 */
  float light_pos[4][4];    /* shader.wgsl:23, vec3 in 16-byte slots */
  float light_color[4][4];  /* shader.wgsl:24 */
  float emissive_factor[4]; /* shader.wgsl:28 */
  uint32_t sampler_index;   /* renderer.ts:1927-1932: 1 U repeat, 2 V repeat, 4 mag linear, 8 min linear */
  uint32_t use_mipmaps;     /* renderer.ts:1977 */
  uint32_t reserved[2];
} OrcFrameUniforms;

typedef struct OrcScene {
  const float* verts;        /* interleaved pos3 uv2 normal3, glTF.ts:186-214 */
  int32_t n_verts;
  const uint16_t* indices;   /* glTF.ts:222-226 */
  int32_t n_indices;
  int32_t n_instances;       /* 1 = the reference's single draw */
  const float* inst_model;   /* [n_instances][16] or NULL → uniforms.model */
  const float* inst_normal;  /* [n_instances][16] or NULL → uniforms.normal */
  const uint8_t* tex[5];     /* rgba8 chains: albedo, metallicRoughness, normal, emissive, occlusion */
  int32_t tex_w[5], tex_h[5], tex_levels[5];
  const float* irradiance;   /* [6][s][s][4] */
  int32_t irr_size;
  const float* prefilter;    /* levels back to back, each [6][s>>l][s>>l][4] */
  int32_t pre_size, pre_levels;
  const uint16_t* lut_f16;   /* [s][s][4] half */
  int32_t lut_size;
  const float* skybox;       /* [6][s][s][4] */
  int32_t sky_size;
  const uint32_t* indices32; /* when not NULL, read instead of `indices` (meshes beyond the reference's Uint16Array) */
} OrcScene;

/* vs_main (shader.wgsl:37-47) for every instance: clip[n][4], world[n][3], nrm[n][3]. */
void orc_vertex_transform(const OrcScene* sc, const OrcFrameUniforms* u, float* clip, float* world, float* nrm);
/* Fixed-function rasterisation of renderer.ts:469-480 state. vis = triangle id + 1 (0 = background),
 * depth = f32 depth buffer.  Rows [y0,y1) of a W x H target; outputs are (y1-y0) x W. */
void orc_raster(const OrcScene* sc, const OrcFrameUniforms* u, int W, int H, int y0, int y1,
                uint32_t* vis, float* depth);
/* fs_main of shader.wgsl and skybox.wgsl for the pixels of rows [y0,y1) given their visibility. */
void orc_shade(const OrcScene* sc, const OrcFrameUniforms* u, int W, int H, int y0, int y1,
               const uint32_t* vis, uint8_t* rgba8);
/* Convenience: raster + shade. */
void orc_draw_frame(const OrcScene* sc, const OrcFrameUniforms* u, int W, int H, int y0, int y1,
                    uint8_t* rgba8, uint32_t* vis_or_null);
/* Test instrumentation: the same frame plus, per pixel, how close any discrete sampling decision of the pixel (mip level
 * of a "nearest" mipmap filter, min / mag switch, texel of a "nearest" filter) was to its switching point; see frame.cpp. */
void orc_draw_frame_margins(const OrcScene* sc, const OrcFrameUniforms* u, int W, int H, int y0, int y1,
                            uint8_t* rgba8, uint32_t* vis, float* margin);
/* The same plus, per pixel and channel, the lowest and highest 8-bit value reachable when the tangent of shader.wgsl:102
 * (a direction built from differences of neighbouring pixels' positions and texcoords) is turned by its own uncertainty,
 * `tangent_ulps` ulps of those positions and texcoords (test instrumentation; single-sample pipeline). */
void orc_draw_frame_bounds(const OrcScene* sc, const OrcFrameUniforms* u, int W, int H, int y0, int y1, float tangent_ulps,
                           uint8_t* rgba8, uint32_t* vis, float* margin, uint8_t* lo8, uint8_t* hi8);

/* Samplers exposed for unit tests. */
void orc_sample_cube(const float* chain, int size, int levels, const float* dir3, float lod, float* rgba_out);
void orc_sample_2d_rgba8(const uint8_t* chain, int w, int h, int levels, int is_srgb, uint32_t sampler_index,
                         float u, float v, const float* duvdx2, const float* duvdy2, float* rgba_out);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
