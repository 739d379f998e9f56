// Internal declarations shared by the translation units of libpbr355.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/pbr355.h"

#define PBR_MAX_LEVELS 16
#define PBR_TILE 32               // raster tile edge in pixels (one 256-thread workgroup, 2x2 pixels per thread)
#define PBR_SUBPIXEL_BITS 8
#define PBR_GUARD 8.0f            // guard band in units of w
#define PBR_PREFILTER_SAMPLES 1024
#define PBR_MAX_SHARDS 4096         // most shards the prefilter's texels can be cut into
#define PBR_LIVE_TILE_CAP 256   // most tiles that the rasteriser starts ahead of the rest (see LiveTiles)
#ifndef PBR_DENSE_TILE_REFS
#define PBR_DENSE_TILE_REFS 48   // a tile with more triangle references than this is one of them
#endif
#define PBR_BIN_CAP 1024          // triangle references a tile keeps in its own row before spilling to the overflow list

// An rgba8 texture with its mip chain in one allocation.  Every level is stored in blocks of 8 x 4
// texels (128 bytes, one cache line), blocks row by row: a bilinear footprint, and the footprints of
// neighbouring pixels, then fall into few lines whichever way the texture is oriented on screen.
struct DevTex8 {
  const uint32_t* base;           // packed r | g<<8 | b<<16 | a<<24
  uint32_t off[PBR_MAX_LEVELS];   // texel offset of each level (levels are padded to whole blocks)
  int32_t w, h, levels, srgb;
};

// Offset of texel (x, y) inside a level that is w texels wide.
__host__ __device__ inline uint32_t tex8_block_offset(int x, int y, int w) {
  const uint32_t blocks_x = (uint32_t)(w + 7) >> 3;
  return (((uint32_t)y >> 2) * blocks_x + ((uint32_t)x >> 3)) * 32u + (((uint32_t)y & 3u) << 3) + ((uint32_t)x & 7u);
}
__host__ __device__ inline size_t tex8_level_texels(int w, int h) { return (size_t)((w + 7) >> 3) * (size_t)((h + 3) >> 2) * 32u; }

// A cube map (or its mip chain) of rgba32f texels: level l is [6][size>>l][size>>l].
struct DevCube {
  const float4* base;
  uint32_t off[PBR_MAX_LEVELS];
  int32_t size, levels;
};

// The copy of a cube map the frame kernels sample: a one-texel border per face (the adjoining texels of the
// neighbouring faces), and only the three channels the shaders read, as three floats per texel: a tap is one
// 12-byte load, a quarter fewer bytes and registers than the rgba32f texel.
struct DevCubeRgb {
  const float* base;              // three floats per texel
  uint32_t off[PBR_MAX_LEVELS];   // texel offset of each level
  int32_t size, levels;           // size of level 0 without the border
  // Level 0 once more for the skybox pass (null for the other maps), in halves and by pairs: entry i of a row holds
  // texels i and i + 1 of that row as (r r' | g g' | b b' | unused), so a bilinear lookup is two 16-byte loads, one
  // per row, and each channel of a row one v_dot2_f32_f16 against the row's two weights.
  const uint4* pairs;
};


// An equirectangular rgba32f image with its mip chain.
struct DevEquirect {
  const float4* base;
  const float* rgb;               // the same chain without its alpha (always 1), three floats per texel: what the prefilter gathers from
  uint32_t off[PBR_MAX_LEVELS];
  int32_t w, h, levels;
};

// One importance sample of the GGX prefilter, in the tangent frame of the output texel.
struct PrefilterSample {
  float lx, ly, lz;   // light direction; lz is also N.L, the weight
  float mix;          // blend between source levels l0 and l0+1
  int32_t l0, l1;
  float wa, wc;       // weights of the two source texels: (1 - mix) * lz and mix * lz
};

// Snapped, front-facing triangle as the rasteriser consumes it (48 bytes).
struct SetupTri {
  int32_t X[3], Y[3];   // vertices in 1/256 pixel, ordered so edge functions are positive inside
  float z0, dz1, dz2;   // depth at vertex 0 and its differences to vertices 1 and 2
  float inv_area;       // 1 / (twice the signed area), f32
  uint32_t id;          // global triangle id
  uint32_t pad;
};

// Everything the shading resolve needs of one triangle, written by the setup kernel so that a pixel
// fetches it in one round of loads: nine 16-byte loads, three more in the rare non-projective case.
struct TriRecord {
  int32_t X0, Y0, X1, Y1, X2, Y2;   // vertices snapped to 1/256 pixel, in index order (valid when projective)
  float inv_area;                   // 1 / (twice the signed area in snapped units)
  uint32_t projective;              // all w > 0 and inside the guard band: screen-space barycentrics apply
  float iw0, iw1, iw2, pad;         // 1 / w of the vertices
  float pos[9], nrm[9], uv[6];      // world positions, transformed normals, texcoords of the three vertices
  float clip[12];                   // clip positions; written (and read) only when not projective
};

// Per-vertex outputs of the vertex stage, laid out for the resolve's gathers.
struct VertexOut {
  float4 clip;      // projection * view * model * position
  float4 world_u;   // world position xyz, texcoord u
  float4 nrm_v;     // normal-matrix * normal xyz, texcoord v
};

// Work lists the rasteriser leaves for the shading resolve: 2x2 pixel quads holding geometry.  pos is
// (x / 2) | (row-in-range / 2) << 16.
struct QuadLists {
  uint2* uniform;       // (pos, id): all four pixels show triangle id - 1
  uint32_t* mixed_pos;  // quads with several triangles or background pixels ...
  uint4* mixed_ids;     // ... and the ids of their pixels (0,0), (1,0), (0,1), (1,1); 0 = background
};

// The append counters of a frame.  One counter in device memory takes about a hundred atomics per microsecond, however
// many wavefronts issue them: an 8K frame of a hundred meshes has ten thousand tiles with geometry and twenty-four
// thousand wavefronts of triangles, and with one atomic each on one address its launches lasted as long as those
// atomics queued (set-up 383 us with, 125 us without).  So whatever is counted once per tile or per wavefront is
// counted in one of PBR_LIST_SEGMENTS groups of counters, 256 bytes apart; the quad lists are cut into as many
// segments, tile t appends to segment t % PBR_LIST_SEGMENTS, and a segment has room for all the quads of the tiles
// that can fall to it.  A set of counters: [0] pieces of clipped triangles set up, [1] bin references spilled,
// [2] overflow flag, [3] covered pixels (statistics), [4] listed tiles, then from word PBR_SEGMENT_FIRST every
// PBR_SEGMENT_WORDS words a group: [0] uniform quads, [1] mixed quads (one 64-bit atomic reserves room in both),
// [2] whole triangles set up (statistics).  Alternate frames of a context use two sets: the second lies eight words
// behind the first, group by group, so both fit PBR_COUNTER_WORDS words.
#define PBR_LIST_SEGMENTS 32
#define PBR_SEGMENT_WORDS 64
#define PBR_SEGMENT_FIRST 64
#define PBR_COUNTER_WORDS (PBR_SEGMENT_FIRST + PBR_LIST_SEGMENTS * PBR_SEGMENT_WORDS)
// quads a segment of the lists holds in a frame of n_tiles tiles, and the lists' length for it
static inline __host__ __device__ uint32_t quad_segment_cap(uint32_t n_tiles) {
  return ((n_tiles + PBR_LIST_SEGMENTS - 1u) / PBR_LIST_SEGMENTS) * (uint32_t)((PBR_TILE / 2) * (PBR_TILE / 2));
}
static inline __host__ __device__ size_t quad_list_len(size_t n_tiles) { return (size_t)quad_segment_cap((uint32_t)n_tiles) * PBR_LIST_SEGMENTS; }

struct FrameParams {
  PbrFrameUniforms u;
  float pv[16];             // projection * view
  int32_t W, H, y0, y1;
  int32_t tiles_x, tiles_y; // tiles covering rows [y0,y1)
  uint32_t tiles_x_magic;   // ceil(2^32 / tiles_x) for tiles_x > 1: tile / tiles_x == umulhi(tile, magic) for every tile index (tiles_x <= 512)
  int32_t n_verts, n_tris, n_instances;
  int32_t has_instance_mats;
  int32_t msaa;             // 1 = four samples per pixel (bit 16 of the sampler index, renderer.ts:1927-1932)
  float two_over_w, two_over_h;   // 2 / W, 2 / H (one correctly rounded division each, done once on the host)
  // skybox.wgsl:25-27: inv_view_proj * (x_ndc, y_ndc, 1, 1) of pixel (x, y) is sky_a + x * sky_bx + y * sky_by
  float sky_a[4], sky_bx[4], sky_by[4];
};

struct PbrContext {
  int device = 0;
  PbrContext* parent = nullptr;   // set in a forked context: the scene (mesh, textures, IBL maps) is the parent's
  bool view_slot = false;         // a hidden context that only lends its working buffers to one view of another context's multi-view pass
  int children = 0;               // forks alive
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string error;

  // mesh (SoA vertex streams)
  float* d_vsoa = nullptr;          // 8 streams of n_verts floats: px py pz u v nx ny nz
  uint32_t* d_indices = nullptr;
  float* d_inst_model = nullptr;    // [n_instances][16]
  float* d_inst_normal = nullptr;
  int32_t n_verts = 0, n_indices = 0, n_instances = 0;
  bool has_instance_mats = false;
  float mesh_lo[3] = {1.0f, 1.0f, 1.0f}, mesh_hi[3] = {0.0f, 0.0f, 0.0f};   // bounding box of the positions (lo > hi: no mesh)

  // material textures
  uint32_t* d_tex[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  DevTex8 tex[5] = {};
  bool tex_has_mips[5] = {false, false, false, false, false};

  // HDR + IBL
  float4* d_hdr = nullptr;
  float* d_hdr_rgb = nullptr;
  DevEquirect hdr = {};
  float4* d_sky = nullptr;   DevCube sky = {};
  float4* d_irr = nullptr;   DevCube irr = {};
  float4* d_pre = nullptr;   DevCube pre = {};
  size_t pre_total_texels = 0;
  uint16_t* d_lut = nullptr; int32_t lut_size = 0;
  PrefilterSample* d_samples = nullptr;   // PBR_MAX_LEVELS tables of 1024
  float* d_irr_tables = nullptr;

  // frame working set (grown on demand)
  VertexOut* d_vout = nullptr;    size_t vout_cap = 0;
  SetupTri* d_setup = nullptr;    size_t setup_cap = 0;
  TriRecord* d_records = nullptr; size_t record_cap = 0;   // one per triangle of every instance
  uint32_t* d_tile_count = nullptr; uint32_t* d_bins = nullptr; size_t tile_cap = 0; uint32_t* d_tile_live = nullptr;   // [tiles], [tiles][PBR_BIN_CAP]
  unsigned long long* d_live_info = nullptr; unsigned long long* d_live_tiles = nullptr; unsigned long long live_serial = 0;   // see LiveTiles
  QuadLists quad_lists = {nullptr, nullptr, nullptr};                                  // work lists of the resolve, each [quad_list_len(tiles)]
  uint2* d_overflow = nullptr; size_t overflow_cap = 0;                                // (tile, setup index) pairs
  size_t min_setup_cap = 0, min_overflow_cap = 0;         // raised when a frame overflowed one of the two
  size_t forced_setup_cap = 0, forced_overflow_cap = 0;   // pbr_debug_set_capacity: exact capacities, until an overflow enlarges them
  bool enlarged_after_overflow = false;
  uint32_t* d_counters = nullptr; // PBR_COUNTER_WORDS words: two sets, used by alternate frames (see PBR_LIST_SEGMENTS)
  bool capture_depth = false;
  uint32_t* h_counters = nullptr; // pinned mirror
  uint32_t* d_vis = nullptr;      float* d_depth = nullptr; uint8_t* d_color = nullptr; size_t pix_cap = 0, vis_cap = 0;   // vis / depth hold vis_cap entries (one per sample)
  int32_t last_w = 0, last_rows = 0, last_samples = 1;

  // timing
  hipEvent_t ev[6] = {};
  bool profiling = false;
  std::vector<hipEvent_t> prof_events;   // per profiled frame: surface launch begin, end; with prof_all: set-up begin, raster begin, surface begin, end
  int32_t prof_frames = 0, prof_cap = 0;
  bool prof_all = false;                 // pbr_profile_begin_kernels: all three launches of a frame are bracketed
};

// ---- kernel launchers (defined in the .hip files) ----
void launch_build_mips_rgba32f(hipStream_t s, float4* chain, const uint32_t* off, int w, int h, int levels);
// level 0 from a row-major image, then the box-filtered levels; chain is in the blocked layout of DevTex8
// srgb_tables: 256 decoded values, then the 257 store thresholds of the 8-bit sRGB codes
void launch_build_mips_rgba8(hipStream_t s, const uint32_t* linear_level0, uint32_t* chain, const uint32_t* off, int w, int h, int levels, int srgb,
                             const float* srgb_tables);
void launch_drop_alpha(hipStream_t s, const float4* rgba, float* rgb, size_t texels);
void launch_equirect_to_cube(hipStream_t s, DevEquirect hdr, float4* out, int size);
void launch_irradiance(hipStream_t s, DevEquirect hdr, const float* tables, int n_phi, int n_theta, float4* out, int size);
// n_single: the leading samples of the table that read one source level (see build_prefilter_table)
void launch_prefilter(hipStream_t s, DevEquirect hdr, const PrefilterSample* samples, int n_samples, int n_single, float inv_weight,
                      float4* out_level, int level_size, int64_t first, int64_t count);
// one level's part of a prefilter call; launch_prefilter_levels puts the large levels of a call into one launch
struct PrefilterLevelRun { const PrefilterSample* samples; int n_samples, n_single; float total_weight; float4* out_level; int level_size; int64_t first, count; };
void launch_prefilter_levels(hipStream_t s, DevEquirect hdr, const PrefilterLevelRun* runs, int n_runs);
void launch_brdf_lut(hipStream_t s, uint16_t* out, int size);

struct SkyShare;   // sky_shade.h
void launch_vertex_stage(hipStream_t s, const FrameParams& p, const float* vsoa, const float* inst_model, const float* inst_normal,
                         VertexOut* out, int n_verts, int n_instances, const SkyShare& share);
// where the setup kernel takes a triangle's vertices from: transformed vertices in memory (vout), or the
// vertex streams, transformed on the fly (vsoa set, vout unused)
struct VertexSource { const VertexOut* vout; const float* vsoa; const float* inst_model; const float* inst_normal; float mesh_lo[3], mesh_hi[3]; };   // (the box: see BinTargets)
// The fullest tiles (more than PBR_DENSE_TILE_REFS triangle references), listed by the set-up launch as each crosses
// that mark, so that the rasteriser can start them before anything else - the frame waits for the fullest tile:
// tiles[k] = serial << 24 | tile for the k-th tile to cross it (k from frame counter 4, k < cap); info[tile] =
// serial << 16 | min(k + 1, 0xffff).  The serial number is one no earlier frame on these arrays used (never 0), so
// entries of earlier frames read as "not listed" without anyone clearing them, and the rasteriser's workgroup k
// learns from the one load of tiles[k] both whether there is a k-th tile and which.
struct LiveTiles { unsigned long long* tiles; unsigned long long* info; uint32_t cap; unsigned long long serial; };
// Where the set-up launch appends a front-facing (sub)triangle and references it from every tile its bounding box
// touches.  A tile keeps its first PBR_BIN_CAP references in its own row of `bins`; any beyond that go to one shared
// overflow list of (tile, triangle) pairs.
struct BinTargets {
  SetupTri* setup; uint32_t setup_cap;
  uint32_t* tile_count; uint32_t* bins;
  uint2* overflow; uint32_t overflow_cap;
  uint32_t* counters;
  LiveTiles live;
  TriRecord* records;
  // when vsoa is set the triangle's vertices are transformed here from the vertex streams (and vout is not read)
  const float* vsoa; const float* inst_model; const float* inst_normal;
  // bounding box of the mesh's positions in object space (lo > hi: none known): an instance whose box misses the
  // frame's rows or columns is skipped as a whole, before any of its triangles is fetched
  float mesh_lo[3], mesh_hi[3];
};
void launch_setup_and_bin(hipStream_t s, const FrameParams& p, const VertexSource& src, const uint32_t* indices, SetupTri* setup, TriRecord* records,
                          uint32_t setup_cap, uint32_t* tile_count, uint32_t* bins, uint2* overflow, uint32_t overflow_cap, uint32_t* counters,
                          const LiveTiles& live, int n_tris, int n_instances, const SkyShare& share);
void launch_raster(hipStream_t s, const FrameParams& p, const SetupTri* setup, uint32_t* tile_count, const uint32_t* bins, const uint2* overflow,
                   uint32_t overflow_cap, uint32_t* vis, float* depth, uint32_t* counters, uint32_t* next_counters, const QuadLists& lists, bool write_vis,
                   const DevCubeRgb& sky_bordered, uint8_t* rgba8, int sky_from_tile, int tiles_x, int tiles_y, const LiveTiles& live, bool split_listed);
struct ResolveInputs {
  const TriRecord* records;
  const VertexOut* vout;
  const uint32_t* indices;
  DevTex8 tex[5];
  DevCube irr, pre, sky;
  DevCubeRgb irr_b, pre_b, sky_b;   // the same maps with a one-texel border per face (level l: [6][n+2][n+2], n = size >> l), rgb only
  const uint16_t* lut;
  // The table's two used channels (A | B << 16, half bits) of the four texels of the bilinear footprint whose upper
  // left texel is (x, y), with the repeat wrap applied: (x, y), (x + 1, y), (x, y + 1), (x + 1, y + 1).  One load per lookup.
  const uint4* lut_quad;
  // The five material textures of one size interleaved texel by texel, in the blocked order and with the level
  // offsets of tex[0]: x = albedo, y = emissive, z = normal (whole texels), w = occlusion.r | metallicRoughness.g << 8 |
  // metallicRoughness.b << 16.  One 16-byte load fetches what fs_main reads of a texel position from all five
  // (null when the textures differ in size).
  const uint4* atlas;
  int32_t lut_size;
};
// covered pixels; the rasteriser has already written the skybox colour of the background pixels
void launch_surface(hipStream_t s, const FrameParams& p, const ResolveInputs& in, const float* srgb_lut, const uint32_t* vis, uint8_t* rgba8,
                    const QuadLists& lists, const uint32_t* counters, int W, int rows, int tiles_x, int tiles_y, bool fast, int resident_blocks, bool shared_chip);
// The multisampled frame: vis (and depth, when wanted) hold four entries per pixel; the colour target already
// holds the skybox colour of every pixel and the resolve blends the shaded triangles over it by coverage.
void launch_raster_msaa(hipStream_t s, const FrameParams& p, const SetupTri* setup, uint32_t* tile_count, const uint32_t* bins, const uint2* overflow,
                        uint32_t overflow_cap, uint32_t* vis, float* depth, uint32_t* counters, uint32_t* next_counters, uint32_t* tile_live,
                        bool write_empty_tiles, int tiles_x, int tiles_y, const LiveTiles& live);
void launch_surface_msaa(hipStream_t s, const FrameParams& p, const ResolveInputs& in, const float* srgb_lut, const uint32_t* vis, const uint32_t* tile_live,
                         uint8_t* rgba8, int W, int rows, bool fast);
int surface_fast_blocks_per_cu();
void launch_cube_border(hipStream_t s, DevCube src, int level, float* dst_level);   // dst: [6][n+2][n+2][3]
void launch_cube_pairs(hipStream_t s, const float* bordered_level0, uint4* pairs, int size);   // DevCubeRgb::pairs from level 0 of the bordered copy
void launch_lut_quad(hipStream_t s, const uint16_t* lut_rgba16f, uint4* quad, int size);
// interleaves five texture chains of `texels` texels each (slot order of PBR_TEX_*) into the layout of ResolveInputs::atlas
void launch_material_atlas(hipStream_t s, const uint32_t* const chains[5], uint4* atlas, size_t texels);
// pixels with at least one covered sample
void launch_count_covered(hipStream_t s, const uint32_t* vis, size_t n_pixels, int samples, uint32_t* covered);

// ---- several views of one scene in one pass (pbr_draw_views_device) ----
// Each of the pass's three launches covers all the views: what a view has for itself (its parameters, the memory its
// frame works in, its colour target) is one ViewArgs on the host, of which every launch takes what it needs as arrays
// in its kernel arguments, and a workgroup finds its view in its block index.
#define PBR_MAX_VIEWS 4
struct RasterTargets {
  const SetupTri* setup; uint32_t* tile_count; const uint32_t* bins; const uint2* overflow; uint32_t overflow_cap;
  uint32_t* counters; uint32_t* next_counters; QuadLists lists; uint32_t* rgba8; LiveTiles live;
};
struct ViewArgs { FrameParams fp; BinTargets bt; RasterTargets ra; const TriRecord* records; };
void launch_setup_views(hipStream_t s, const ViewArgs* views, int n_views, const uint32_t* indices, int n_tris, int n_instances, const SkyShare& share);
void launch_raster_views(hipStream_t s, const ViewArgs* views, int n_views, const DevCubeRgb& sky_bordered, int sky_from_tile, int tiles_x, int tiles_y,
                         uint32_t live_cap);
// the surface pass of the specialised pipeline; a single frame is a pass of one view
void launch_surface_views(hipStream_t s, const ViewArgs* views, int n_views, const ResolveInputs& in, const float* srgb_lut, int resident_blocks,
                          bool shared_chip);
