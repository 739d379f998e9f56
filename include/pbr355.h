/*
 * pbr355 — C ABI of the MI355X (gfx950) PBR frame path and IBL precompute.
 *
 * The reference (maoz233/wgpu-renderer) has no FFI seam: its Renderer class drives a WebGPU
 * GPUDevice directly.  Each entry point below stands in for the WebGPU calls one reference
 * method makes, so a host that keeps the reference's Renderer / glTFLoader / Camera / HDRLoader
 * surface can bind this library instead of navigator.gpu.  Citations are file:line in the
 * reference checkout.
 *
 * Conventions: every call returns 0 on success or a negative PBR_E_* code; pbr_last_error()
 * gives the message.  The caller owns all host pointers; the library copies during the call.
 * A context is bound to one HIP device and one stream; calls on one context are not re-entrant,
 * different contexts may be driven from different threads.  Matrices are column-major f32,
 * cube maps are [face][y][x][rgba] with faces +X,-X,+Y,-Y,+Z,-Z.
 */
#ifndef PBR355_H
#define PBR355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBR_OK 0
#define PBR_E_ARG (-1)        /* invalid argument */
#define PBR_E_STATE (-2)      /* a prerequisite upload / precompute is missing */
#define PBR_E_DEVICE (-3)     /* HIP runtime error or no usable gfx950 device */
#define PBR_E_UNSUPPORTED (-4)

typedef struct PbrContext PbrContext;

/* Material texture slots in the order renderer.ts:605-632 loads them. */
enum PbrTextureSlot {
  PBR_TEX_ALBEDO = 0,              /* rgba8unorm-srgb, renderer.ts:647 */
  PBR_TEX_METALLIC_ROUGHNESS = 1,  /* rgba8unorm,      renderer.ts:663 */
  PBR_TEX_NORMAL = 2,              /* rgba8unorm,      renderer.ts:681 */
  PBR_TEX_EMISSIVE = 3,            /* rgba8unorm-srgb, renderer.ts:697 */
  PBR_TEX_OCCLUSION = 4            /* rgba8unorm,      renderer.ts:713 */
};

/* IBL products, renderer.ts:757-779 and :978-993. */
enum PbrIblMap {
  PBR_MAP_SKYBOX = 0,      /* CubemapType.SKYBOX:     [6][s][s] rgba32f */
  PBR_MAP_IRRADIANCE = 1,  /* CubemapType.IRRADIANCE: [6][s][s] rgba32f */
  PBR_MAP_PREFILTER = 2,   /* CubemapType.PREFILTER:  mips of [6][s>>m][s>>m] rgba32f */
  PBR_MAP_BRDF_LUT = 3,    /* generateBRDFLUT:        [s][s] rgba16f */
  PBR_MAP_HDR = 4          /* the uploaded equirect HDR and its mip chain, rgba32f */
};

/* What drawFrame writes into its uniform buffers each frame (renderer.ts:1828-1932). */
typedef struct PbrFrameUniforms {
  float model[16];          /* Transform.model       shader.wgsl:15, renderer.ts:1833-1834 */
  float normal[16];         /* Transform.normal      shader.wgsl:16, renderer.ts:1837-1838 */
  float view[16];           /* Transform.view        shader.wgsl:17, renderer.ts:1841-1842 */
  float projection[16];     /* Transform.projection  shader.wgsl:18, renderer.ts:1845-1853 */
  float inv_view_proj[16];  /* skybox matrix         skybox.wgsl:7,  renderer.ts:1864-1867 */
  float view_pos[4];        /* shader.wgsl:22, renderer.ts:1878-1884 */
  float light_pos[4][4];    /* shader.wgsl:23, renderer.ts:1887-1894 (vec3 in 16-byte slots) */
  float light_color[4][4];  /* shader.wgsl:24, renderer.ts:1905-1912 */
  float emissive_factor[4]; /* shader.wgsl:28, renderer.ts:614-626 */
  uint32_t sampler_index;   /* renderer.ts:1927-1932: 1 U repeat | 2 V repeat | 4 mag linear | 8 min linear | 16 four samples per pixel (MSAA) */
  uint32_t use_mipmaps;     /* renderer.ts:1977: material textures sampled with their mip chains */
  uint32_t reserved[2];
} PbrFrameUniforms;

/* Per-frame timings (the reference's timestamp query, renderer.ts:1968-1974, :2040-2044) and
 * the coverage figure the bandwidth model needs.  Filling it synchronises the stream. */
typedef struct PbrFrameStats {
  float ms_vertex;      /* 0: the vertex transform runs inside the setup kernel (kept for layout compatibility) */
  float ms_setup;       /* vertex transform + triangle setup + binning kernel */
  float ms_raster;      /* tile rasteriser */
  float ms_resolve;     /* deferred shading resolve */
  float ms_total;       /* first launch to last completion */
  uint32_t covered_pixels;   /* pixels whose visibility entry (any of the four with MSAA) is a triangle */
  uint32_t binned_refs;      /* triangle-tile references that spilled past a tile's own bin row */
  uint32_t setup_triangles;  /* front-facing (sub)triangles after clipping and culling */
  uint32_t uniform_quads;    /* 2x2 pixel quads whose four pixels show one triangle */
  uint32_t mixed_quads;      /* quads with a covered pixel that are not uniform (edges, silhouettes) */
} PbrFrameStats;

/* ---- lifetime: replaces requestAdapter/requestDevice, renderer.ts:144-198 ---- */
int pbr_create(int device_ordinal, PbrContext** ctx_out);
int pbr_destroy(PbrContext* ctx);
/* A second context on the same device that draws the SAME scene: it shares the parent's mesh, textures and IBL maps
 * (one copy in device memory, so frames in flight meet in the caches) and owns only what a frame in flight needs
 * for itself: a stream, working buffers, counters.  Several frames in flight = one context plus forks, each on its
 * own stream (the reference keeps frames in flight through the browser's queue, renderer.ts:2033-2050).  A fork
 * cannot upload or generate anything (PBR_E_STATE); changes made through the parent are seen by its forks' next
 * frames, and must not overlap them.  Destroy the forks before the parent. */
int pbr_fork(PbrContext* parent, PbrContext** ctx_out);
const char* pbr_last_error(const PbrContext* ctx);   /* ctx may be NULL: error of the last failed pbr_create on this thread */
/* Run this context's work on an existing HIP stream (hipStream_t); NULL restores the context's own. */
int pbr_set_stream(PbrContext* ctx, void* hip_stream);
int pbr_synchronize(PbrContext* ctx);

/* ---- geometry: createVertexBuffer/createIndexBuffer, renderer.ts:504-540 ----
 * verts: n_verts x 8 floats (pos3, uv2, normal3 — the Geometry layout of glTF.ts:186-214);
 * indices: uint16 triangle list (glTF.ts:222-226).  instance_count>1 draws the mesh once per
 * instance with the model / normal matrices given (each instance_count x 16 floats); with
 * instance_count==1 both may be NULL and the frame's Transform is used, as in the reference. */
int pbr_upload_mesh(PbrContext* ctx, const float* verts, int32_t n_verts, const uint16_t* indices, int32_t n_indices,
                    int32_t instance_count, const float* instance_model, const float* instance_normal);
/* The same upload for meshes with more than 65 536 vertices (glTF componentType 5125, UNSIGNED_INT, which
 * the reference's loader does not read: glTF.ts:222-226 builds a Uint16Array whatever the accessor says). */
int pbr_upload_mesh_u32(PbrContext* ctx, const float* verts, int32_t n_verts, const uint32_t* indices, int32_t n_indices,
                        int32_t instance_count, const float* instance_model, const float* instance_normal);

/* ---- material textures: createTexture2DFromSource + copySourceToTexture, renderer.ts:1039-1101.
 * rgba8 is the decoded image, w*h*4 bytes, top row first.  gen_mips!=0 also builds the full chain
 * (calculateMipLevelCount, utils.ts:20-24) the way generateMipmaps does (renderer.ts:1103-1321). ---- */
int pbr_upload_texture(PbrContext* ctx, int slot, const uint8_t* rgba8, int32_t w, int32_t h, int32_t is_srgb, int32_t gen_mips);

/* ---- HDR source: the writeTexture of generateCubemap (renderer.ts:1449, :1520) plus the mip
 * chain generateMipmaps builds for the prefilter source (renderer.ts:1537). ---- */
int pbr_upload_hdr(PbrContext* ctx, const float* rgba32f, int32_t w, int32_t h);

/* ---- IBL precompute: generateCubemap / generateBRDFLUT, renderer.ts:1323-1732 ---- */
int pbr_ibl_equirect_to_cube(PbrContext* ctx, int32_t size);   /* CubemapType.SKYBOX, equirectangular.wgsl */
int pbr_ibl_irradiance(PbrContext* ctx, int32_t size);         /* CubemapType.IRRADIANCE, irradiancemap.wgsl */
/* CubemapType.PREFILTER, prefiltermap.wgsl: mips levels with roughness mip/(mips-1) (renderer.ts:1559),
 * samples must be 1024 (SAMPLE_COUNT, prefiltermap.wgsl:3).  (0,1) computes everything.  With shard_count > 1 the
 * call filters what shard shard_index needs to hold and to contribute (pbr_ibl_prefilter_partition): the leading
 * levels that are cheaper to filter than to send (roughness 0: one sample per texel), whole, and the shard's own
 * range of the remaining texels, which are cut into shard_count ranges of equal cost in the flattened
 * (mip, face, y, x) order.  A texel's value does not depend on the shard that computes it. */
int pbr_ibl_prefilter(PbrContext* ctx, int32_t size, int32_t mips, int32_t samples, int32_t shard_index, int32_t shard_count);
/* The same filter for an arbitrary piece of the map: texels [first_texel, first_texel + texel_count) of the
 * flattened (mip, face, y, x) order (the dispatch loop of renderer.ts:1558-1609 cut at any texel).  For drivers
 * that balance the work themselves, e.g. from measured times; pbr_ibl_prefilter(shard, count) is this call on
 * the range pbr_ibl_prefilter_partition gives. */
int pbr_ibl_prefilter_texels(PbrContext* ctx, int32_t size, int32_t mips, int32_t samples, uint64_t first_texel, uint64_t texel_count);
int pbr_ibl_brdf_lut(PbrContext* ctx, int32_t size);           /* brdf.wgsl */

/* Size in bytes and dimensions of one level of a map (level 0 for single-level maps). */
int pbr_ibl_level_info(PbrContext* ctx, int which, int32_t level, int32_t* w, int32_t* h, int32_t* layers, size_t* bytes);
/* Copy one level to host memory (rgba32f, or rgba16f bits for the LUT). */
int pbr_ibl_download(PbrContext* ctx, int which, int32_t level, void* out, size_t out_bytes);
/* Device address of a whole map (all levels back to back) for device-side exchange, e.g. an
 * all-gather of prefilter shards; total_bytes may be NULL.  The library samples cube maps through
 * a derived copy that it rebuilds at the first frame after this call: call it (again) after the
 * last write through the pointer, and order that write before the frame. */
int pbr_ibl_device_ptr(PbrContext* ctx, int which, void** dev_ptr, size_t* total_bytes);
/* Byte range [begin,end) of the prefilter map that only shard (shard_index, shard_count) computes: what it has to
 * send to the others (the leading levels that every shard filters for itself lie in front of shard 0's range). */
int pbr_ibl_prefilter_shard_range(PbrContext* ctx, int32_t shard_index, int32_t shard_count, size_t* begin, size_t* end);
/* How the per-level loop of generateCubemap (renderer.ts:1558-1609, roughness schedule :1559) is divided among
 * shard_count shards (shard_count + 1 entries).  cuts[0] is the number of texels, at the front of the flattened
 * (mip, face, y, x) order, that every shard filters for itself and nobody sends: the leading levels whose texels
 * keep at most eight of the 1024 GGX samples, i.e. roughness 0, where a texel is one sample and a GPU produces
 * texels faster than its links carry them (level 0: three quarters of the map's bytes, a fiftieth of its work);
 * 0 for a single shard.  Shard k owns texels [cuts[k], cuts[k + 1]), cuts[shard_count] being the texel count of
 * the whole map; the ranges have equal cost.  A texel of level m weighs the number of the 1024 GGX samples that
 * contribute at that level's roughness (those with N.L > 0, prefiltermap.wgsl:138) plus a small constant for the
 * per-texel work; a sample of the +-Y faces - most of all of their middle rows, which look at the poles - and of the
 * rougher levels weighs more by factors measured on MI355X (its source pixels are scattered over more cache lines).  Host arithmetic only: needs neither a context nor a
 * device. */
int pbr_ibl_prefilter_partition(int32_t size, int32_t mips, int32_t shard_count, uint64_t* cuts);
/* Number of the 1024 samples of prefiltermap.wgsl:131-168 that contribute (N.L > 0) at level `level` of `mips`
 * levels: the executed work per texel, for cost models and benchmarks.  Host arithmetic only. */
int pbr_ibl_prefilter_kept_samples(int32_t mips, int32_t level, int32_t* kept);

/* ---- the frame: drawFrame's render pass, renderer.ts:1998-2013 ----
 * Renders rows [y0,y1) of a w x h target (y0,y1 even, or y1==h) with the helmet draw
 * (drawIndexed, renderer.ts:2006) followed by the skybox draw (renderer.ts:2011).
 * rgba8_out: (y1-y0)*w*4 bytes, R,G,B,A per pixel; vis_out (optional): (y1-y0)*w uint32, triangle
 * id + 1 with instance i's triangles offset by i*n_indices/3, 0 where the skybox shows.
 * With bit 16 of u->sampler_index set the frame goes through the 4x multisampled pipeline
 * (renderer.ts:243-259, :492-501, :1951-1954): coverage and depth per sample at the standard sample
 * points, each triangle of a pixel shaded once at the pixel centre, the target resolved to the
 * average of the four samples.  vis_out / d_vis then hold four uint32 per pixel (sample order) and
 * pbr_debug_depth four floats per pixel.
 * Arithmetic: f32 throughout, as the shaders', except the skybox's bilinear taps (skybox.wgsl:28), which are taken
 * from a copy of the cube in halves with weights in halves: at most one 8-bit step from the f32 blend, on a few per
 * cent of the skybox's pixels. */
int pbr_draw_frame(PbrContext* ctx, const PbrFrameUniforms* u, int32_t w, int32_t h, int32_t y0, int32_t y1,
                   uint8_t* rgba8_out, uint32_t* vis_out, PbrFrameStats* stats);
/* Same pass with outputs left in device memory (d_vis may be NULL); returns without synchronising
 * unless stats is given.  The working buffers for clipped and binned triangles are sized for views from outside
 * the scene; a frame that overflows one (a camera inside a dense instanced scene can) is incomplete.  pbr_draw_frame,
 * and this call with stats, notice, enlarge the buffer to its worst case and draw the frame again; this call without
 * stats only queues the frame, and the overflow is reported, and the buffer enlarged for the following frames, by the
 * next pbr_synchronize (PBR_E_STATE, "draw it again"). */
int pbr_draw_frame_device(PbrContext* ctx, const PbrFrameUniforms* u, int32_t w, int32_t h, int32_t y0, int32_t y1,
                          void* d_rgba8, void* d_vis, PbrFrameStats* stats);
/* Several frames of one scene in one pass: view v is drawn with u[v] into d_rgba8[v] (n_views of each, 1 to 4; all
 * views share the target size and row range), and every target ends up with exactly the bytes pbr_draw_frame_device
 * would have given it.  For camera paths and other callers that have several frames' uniforms at hand (the reference
 * draws one frame per requestAnimationFrame, renderer.ts:1820-1826, :2050, and submits it on its own, :2033): each of
 * the pass's three launches then covers all the views, so their fixed costs are paid once per pass instead of once
 * per frame.  Queues the pass and returns; overflow reporting as for pbr_draw_frame_device without stats.  Views with
 * another sampler than the reference's default (index 15), multisampled views, or textures of differing sizes are
 * drawn one after the other instead.  The context keeps working buffers per view (some 70 MB per further view at
 * 3840 x 2160) until it is destroyed or pbr_release_view_buffers is called. */
int pbr_draw_views_device(PbrContext* ctx, int32_t n_views, const PbrFrameUniforms* u, int32_t w, int32_t h, int32_t y0, int32_t y1,
                          void* const* d_rgba8);
/* Gives back the working buffers pbr_draw_views_device keeps for views 1, 2, ... of its passes (view 0 works in
 * the context's own): waits for the context's stream, then frees them.  A later pass allocates them again.  For
 * callers that drew a camera path in passes and go on frame by frame (the reference holds no such memory: it draws
 * one frame per submit, renderer.ts:2033). */
int pbr_release_view_buffers(PbrContext* ctx);
/* pbr_draw_frame_device for callers that cannot allocate device memory: code under HIP graph capture, serving
 * loops that must not touch the allocator, harnesses that forbid hipMalloc.  It has no counterpart in the
 * reference.  Same pass, same result bits, but everything the frame keeps between its launches (append counters,
 * tile bins, set-up triangles, shading records, quad work lists) lives in one fixed array of 31.2 MB that
 * every GPU the library runs on holds until the process ends, so the call makes no hipMalloc / hipFree and the
 * frame has to fit it: single-sample pipeline only (PBR_E_UNSUPPORTED with bit 16 of sampler_index), at most
 * 2048 tiles of 32 x 32 pixels in rows [y0,y1) (e.g. 1920 x 1080) and at most 32768 triangles over all instances
 * (PBR_E_ARG beyond either).  The lookup copies the library derives from the IBL maps and from the material
 * textures must exist already: PBR_E_STATE until one pbr_draw_frame_device / pbr_draw_frame has run after the last
 * IBL or texture call.  Every call on
 * one GPU, from whichever context, uses the same array: calls that overlap on one GPU race, so order them on
 * one stream (or with events).  A capacity overflow is reported by the next pbr_synchronize. */
int pbr_draw_frame_static(PbrContext* ctx, const PbrFrameUniforms* u, int32_t w, int32_t h, int32_t y0, int32_t y1, void* d_rgba8);
/* Stage outputs for verification: clip[n][4], world[n][3], normal[n][3] of vs_main
 * (shader.wgsl:37-47) for n = instance_count*n_verts, and the f32 depth buffer of the last frame. */
int pbr_debug_vertex_stage(PbrContext* ctx, const PbrFrameUniforms* u, float* clip, float* world, float* normal);
/* One level of an uploaded material texture as the device holds it after generateMipmaps (renderer.ts:1103-1321):
 * w x h rgba8, top row first; rgba8_out == NULL only reports the level's size.  For verification of the mip chain
 * (the per-level 8-bit rounding and, for the sRGB formats, the decode - average - encode of every level). */
int pbr_debug_texture_level(PbrContext* ctx, int slot, int32_t level, uint8_t* rgba8_out, size_t out_bytes, int32_t* w, int32_t* h);
/* Test hook: exact capacities of the set-up triangle buffer and of the tile-reference overflow list (0 = the
 * library's own sizing) until an overflow enlarges them, so that the overflow handling can be exercised. */
int pbr_debug_set_capacity(PbrContext* ctx, size_t setup_triangles, size_t overflow_refs);
/* The depth buffer normally lives only in on-chip memory during rasterisation; enable capture to
 * have the following frames also write it to device memory so pbr_debug_depth can return it. */
int pbr_debug_capture_depth(PbrContext* ctx, int32_t enable);
int pbr_debug_depth(PbrContext* ctx, float* depth_out, size_t n_floats);

/* ---- kernel timing for benchmarks: while enabled, every frame records one HIP event before and one
 * after its surface-shading kernel (the frame's dominant kernel) on the context's stream.
 * pbr_profile_end waits for the stream and returns the number of frames, the summed duration of that
 * kernel in milliseconds, and span_ms = the time from the end of the first profiled frame to the end
 * of the last (so span_ms / (frames - 1) is the device time per frame, gaps included).  No counterpart
 * in the reference beyond its timestamp query (renderer.ts:1968-1974). ---- */
int pbr_profile_begin(PbrContext* ctx, int32_t max_frames);
int pbr_profile_end(PbrContext* ctx, int32_t* frames, float* surface_ms_total, float* span_ms);
/* The same for all three launches of a frame (or of a pbr_draw_views_device pass, which counts as one frame): four
 * events per frame, kernel_ms_total[3] = summed durations of the set-up, raster and surface-shading launches.  The
 * events take stream time; meant for frames drawn one at a time, after a timed region, not inside one. */
int pbr_profile_begin_kernels(PbrContext* ctx, int32_t max_frames);
int pbr_profile_end_kernels(PbrContext* ctx, int32_t* frames, float* kernel_ms_total, float* span_ms);

#ifdef __cplusplus
}
#endif
#endif
