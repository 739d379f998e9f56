// Host side of libpbr355: the C ABI declared in include/pbr355.h.  Owns device memory, prepares
// the small per-call tables on the host and enqueues the gfx950 kernels on the context's stream.
// There is no CPU fallback: every entry point needs a HIP device.
#include "pbr_internal.h"
#include "sky_shade.h"
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

// ---- Working memory of pbr_draw_frame_static: one fixed array per GPU instead of buffers grown with hipMalloc.
// What a frame keeps between its three launches (append counters, tile bins, set-up triangles, shading records, the
// two quad work lists) is laid out in it back to back.  Host code never uses the array by name: its address on each
// device is taken once with hipGetSymbolAddress (pbr_create, with that device current) and travels as a pointer.
// The two sets of a context's append counters are eight words apart (see PBR_LIST_SEGMENTS: so are their groups), and
// the first sixteen words hold what the host looks at after every frame (the overflow flags).
constexpr size_t kCounterBytes = PBR_COUNTER_WORDS * sizeof(uint32_t), kCounterHeadBytes = 16 * sizeof(uint32_t);
#define PBR_STATIC_MAX_TILES 2048          // 32 x 32 pixel tiles of the row range, e.g. 1920 x 1080 (60 x 34 = 2040)
#define PBR_STATIC_MAX_TRIANGLES 32768     // triangles x instances of the uploaded mesh
namespace static_layout {
constexpr size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr size_t kSetupCap = PBR_STATIC_MAX_TRIANGLES + PBR_STATIC_MAX_TRIANGLES / 4 + 4096;
constexpr size_t kOverflowCap = kSetupCap * 2 + (1u << 16);
constexpr size_t kQuads = (size_t)(PBR_STATIC_MAX_TILES / PBR_LIST_SEGMENTS) * (PBR_TILE / 2) * (PBR_TILE / 2) * PBR_LIST_SEGMENTS;   // quad_list_len(PBR_STATIC_MAX_TILES)
static_assert(PBR_STATIC_MAX_TILES % PBR_LIST_SEGMENTS == 0, "the quad lists' segments share the tiles evenly");
constexpr size_t kCounters = 0;                                                                   // PBR_COUNTER_WORDS u32 (both sets)
constexpr size_t kTileCount = align_up(kCounters + PBR_COUNTER_WORDS * sizeof(uint32_t));                        // [tiles] u32
constexpr size_t kLiveInfo = align_up(kTileCount + PBR_STATIC_MAX_TILES * sizeof(uint32_t));      // [tiles] u64
constexpr size_t kLiveTiles = align_up(kLiveInfo + PBR_STATIC_MAX_TILES * sizeof(uint64_t));      // [PBR_LIVE_TILE_CAP] u64
constexpr size_t kZeroedBytes = align_up(kLiveTiles + PBR_LIVE_TILE_CAP * sizeof(uint64_t));      // cleared at the start of every call
constexpr size_t kBins = kZeroedBytes;                                                            // [tiles][PBR_BIN_CAP] u32
constexpr size_t kSetup = align_up(kBins + (size_t)PBR_STATIC_MAX_TILES * PBR_BIN_CAP * sizeof(uint32_t));
constexpr size_t kOverflow = align_up(kSetup + kSetupCap * sizeof(SetupTri));
constexpr size_t kRecords = align_up(kOverflow + kOverflowCap * sizeof(uint2));
constexpr size_t kUniform = align_up(kRecords + (size_t)PBR_STATIC_MAX_TRIANGLES * sizeof(TriRecord));
constexpr size_t kMixedPos = align_up(kUniform + kQuads * sizeof(uint2));
constexpr size_t kMixedIds = align_up(kMixedPos + kQuads * sizeof(uint32_t));
constexpr size_t kBytes = align_up(kMixedIds + kQuads * sizeof(uint4));                           // 31.2 MB
}  // namespace static_layout
static __device__ __align__(256) unsigned char scratch[static_layout::kBytes];

namespace {

thread_local std::string g_create_error;

// Address of `scratch` on each device (every GPU holds its own copy of the array at its own address).
constexpr int kMaxDevices = 64;
unsigned char* g_scratch_ptr[kMaxDevices] = {};

int fail(PbrContext* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->error = buf; else g_create_error = buf;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return fail(ctx, PBR_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

template <typename T>
void free_dev(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

int level_count(int w, int h) { return (int)(1 + std::log2((double)std::max(w, h))); }   // utils.ts:20-24

void chain_offsets(int w, int h, int levels, uint32_t* off, size_t* total) {
  size_t o = 0;
  for (int l = 0; l < PBR_MAX_LEVELS; ++l) {
    off[l] = (uint32_t)o;
    if (l < levels) o += (size_t)std::max(1, w >> l) * std::max(1, h >> l);
  }
  *total = o;
}
void tex8_offsets(int w, int h, int levels, uint32_t* off, size_t* total) {
  size_t o = 0;
  for (int l = 0; l < PBR_MAX_LEVELS; ++l) {
    off[l] = (uint32_t)o;
    if (l < levels) o += tex8_level_texels(std::max(1, w >> l), std::max(1, h >> l));
  }
  *total = o;
}
void cube_offsets(int size, int levels, uint32_t* off, size_t* total) {
  size_t o = 0;
  for (int l = 0; l < PBR_MAX_LEVELS; ++l) {
    off[l] = (uint32_t)o;
    if (l < levels) { size_t s = std::max(1, size >> l); o += 6 * s * s; }
  }
  *total = o;
}

// (projection * view) with the element order the vertex kernel and the parity oracle use.
#pragma clang fp contract(off)
void mat4_mul_host(const float* a, const float* b, float* o) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      volatile float p0 = a[r] * b[c * 4], p1 = a[4 + r] * b[c * 4 + 1], p2 = a[8 + r] * b[c * 4 + 2], p3 = a[12 + r] * b[c * 4 + 3];
      volatile float s01 = p0 + p1;
      volatile float s012 = s01 + p2;
      o[c * 4 + r] = s012 + p3;
    }
}

uint32_t radical_inverse(uint32_t bits) {
  bits = (bits << 16) | (bits >> 16);
  bits = ((bits & 0x55555555u) << 1) | ((bits & 0xAAAAAAAAu) >> 1);
  bits = ((bits & 0x33333333u) << 2) | ((bits & 0xCCCCCCCCu) >> 2);
  bits = ((bits & 0x0F0F0F0Fu) << 4) | ((bits & 0xF0F0F0F0u) >> 4);
  bits = ((bits & 0x00FF00FFu) << 8) | ((bits & 0xFF00FF00u) >> 8);
  return bits;
}

// Everything prefiltermap.wgsl:131-168 computes per sample that does not depend on the output
// texel.  With N = R = V the half vector, N.L, the pdf and therefore the source mip are functions
// of the sample index and the roughness only; they are tabulated in the texel's tangent frame
// (tangent, bitangent, N) and samples with N.L <= 0 are dropped.
// The samples whose two source levels coincide (a level of detail below the finest source level or above the coarsest:
// both clamp to the same level, the blend of a pixel with itself) come first in the table, *n_single of them, with
// the whole weight N.L in wa: the kernel fetches one pixel for them instead of two.  Half the samples of the
// smoothest filtered level of a large map, three quarters of the whole work, are of this kind.
int build_prefilter_table(float roughness, int src_levels, PrefilterSample* out, float* total_weight, int* n_single) {
  const float PI = 3.14159265359f;
  const uint32_t N = PBR_PREFILTER_SAMPLES;
  if (roughness == 0.0f) {
    // every sample is H = L = N with weight 1 and source level 0: one entry reproduces the sum exactly
    PrefilterSample s = {0.0f, 0.0f, 1.0f, 0.0f, 0, 0, 1.0f, 0.0f};
    out[0] = s;
    *total_weight = 1.0f;
    *n_single = 1;
    return 1;
  }
  int n = 0;
  float total = 0.0f;
  const float a = roughness * roughness;
  for (uint32_t i = 0; i < N; ++i) {
    const float xi_x = (float)i / (float)N, xi_y = (float)radical_inverse(i) * 2.3283064365386963e-10f;
    const float phi = 2.0f * PI * xi_x;
    const float ct = std::sqrt((1.0f - xi_y) / (1.0f + (a * a - 1.0f) * xi_y));
    const float st = std::sqrt(1.0f - ct * ct);
    float hx = std::cos(phi) * st, hy = std::sin(phi) * st, hz = ct;
    const float hl = std::sqrt(hx * hx + hy * hy + hz * hz);
    hx /= hl; hy /= hl; hz /= hl;
    const float vdoth = hz;   // V = N = (0,0,1) in the tangent frame
    float lx = 2.0f * vdoth * hx, ly = 2.0f * vdoth * hy, lz = 2.0f * vdoth * hz - 1.0f;
    const float ll = std::sqrt(lx * lx + ly * ly + lz * lz);
    lx /= ll; ly /= ll; lz /= ll;
    const float ndotl = std::max(lz, 0.0f);
    if (!(ndotl > 0.0f)) continue;
    const float a2 = a * a, ndoth = std::max(hz, 0.0f), hdotv = std::max(vdoth, 0.0f);
    float denom = ndoth * ndoth * (a2 - 1.0f) + 1.0f;
    denom = PI * denom * denom;
    const float D = a2 / denom;
    const float pdf = D * ndoth / (4.0f * hdotv) + 0.0001f;
    const float sa_texel = 4.0f * PI / (6.0f * 512.0f * 512.0f);   // prefiltermap.wgsl:143-144: 512 is hard-coded
    const float sa_sample = 1.0f / ((float)N * pdf + 0.0001f);
    const float mip = 0.5f * std::log2(sa_sample / sa_texel);
    const int m0 = (int)std::floor(mip);
    PrefilterSample s;
    s.lx = lx; s.ly = ly; s.lz = ndotl;
    s.mix = mip - (float)m0;
    s.l0 = std::min(std::max(m0, 0), src_levels - 1);       // a level outside the chain is clamped into it
    s.l1 = std::min(std::max(m0 + 1, 0), src_levels - 1);
    s.wa = (1.0f - s.mix) * s.lz; s.wc = s.mix * s.lz;   // this file is built without FMA contraction: the kernel's former two roundings each
    if (s.l0 == s.l1) { s.wa = s.lz; s.wc = 0.0f; }
    out[n++] = s;
    total += ndotl;
  }
  *total_weight = total;
  *n_single = (int)(std::stable_partition(out, out + n, [](const PrefilterSample& s) { return s.l0 == s.l1; }) - out);
  return n;
}

// renderer.ts:1559: roughness = mip / (maxMipLevels - 1) stored as f32; a single level gets 0
float prefilter_roughness(int level, int mips) { return mips > 1 ? (float)((double)level / (double)(mips - 1)) : 0.0f; }

// Cost model of the prefilter, for cutting it into shards of equal duration: picoseconds per texel on one MI355X,
// from timings of every face of every level of the 2048^2 x 10 map filtered from a 4096 x 2048 source
// (scripts/gpu_prefilter_shards.py, profiles/r03_prefilter_shards.json).  A texel costs the samples its level's table
// keeps (those with N.L > 0; one at roughness 0) plus a constant for the work a texel does whatever its sample
// count, times what a sample costs there:
//   * 1.22 ns per thousand samples on the side faces of the smoothest filtered level (the kernel is bound by
//     vector-instruction issue there, and half that level's samples read one source level instead of two);
//   * the +-Y faces look at the pole rows of the equirect image, where a block of neighbouring directions is stretched
//     over many more pixels of a row (more cache lines per wavefront load): 1.45 x at the smoothest filtered level,
//     1.74 x at the next; rougher levels scatter their samples wider on every face: 1.47 x, then 1.6 x per sample;
//   * levels of fewer than 2^18 texels spread a texel's samples over 8 or 64 lanes (launch_prefilter) and no longer
//     fill the chip: 1.85 x and 1.6 x per sample;
//   * every level is a launch of its own with a floor of 8.5 us, spread here over the level's texels.
// Only the balance of the shards depends on these numbers, never a result.
constexpr uint64_t PREFILTER_TEXEL_OVERHEAD = 8;
struct PrefilterCost { int kept[PBR_MAX_LEVELS]; };
void prefilter_costs(int mips, PrefilterCost* pc) {
  // (the tables take a fraction of a millisecond of host arithmetic; a caller that filters map after map with one level
  // count would otherwise pay it ahead of every call's first launch)
  static thread_local int cached_mips = -1;
  static thread_local PrefilterCost cached;
  if (cached_mips == mips) { *pc = cached; return; }
  std::vector<PrefilterSample> table(PBR_PREFILTER_SAMPLES);
  for (int m = 0; m < mips; ++m) {
    float weight;
    int single;
    pc->kept[m] = build_prefilter_table(prefilter_roughness(m, mips), PBR_MAX_LEVELS, table.data(), &weight, &single);   // the count does not depend on the source's level count
  }
  cached = *pc; cached_mips = mips;
}
// picoseconds per texel of face f of level m; `part` 0 - 2: the rows of the face in front of its middle quarter, the middle
// quarter, and the rows behind it.  On the +-Y faces the middle rows look at the poles, where the samples of
// neighbouring texels fan out over all longitudes of the equirectangular image: timed in bands of rows, those two
// eighths of a pole face cost 1.7 times the other six, which cost what a side face costs; with a pole face 1.45 (1.73
// at the next level) times a side face in all, that puts its middle quarter at 2.7 (3.8) side faces' rate.
uint64_t prefilter_texel_cost(const PrefilterCost& pc, int size, int mips, int m, int f, int part) {
  const float roughness = prefilter_roughness(m, mips);
  const uint64_t s = (uint64_t)std::max(1, size >> m), level_texels = 6 * s * s;
  if (roughness == 0.0f) return 10 + 8500000 / level_texels;   // a copy of the source pixel
  const bool pole = f == 2 || f == 3, middle = part == 1;
  const uint64_t scatter = roughness <= 0.17f ? (pole ? (middle ? 271 : 103) : 100) : roughness <= 0.28f ? (pole ? (middle ? 470 : 125) : 122) : (pole ? 141 : 133);   // hundredths
  const uint64_t shape = level_texels >= (1u << 18) ? 100 : level_texels >= (1u << 14) ? 185 : 160;
  return ((uint64_t)pc.kept[m] + PREFILTER_TEXEL_OVERHEAD) * 122 * scatter * shape / 1000000 + 8500000 / level_texels;
}
// texels of part `part` of a face of s x s texels (rows [0, 3s/8), [3s/8, 5s/8), [5s/8, s))
uint64_t prefilter_part_texels(uint64_t s, int part) {
  const uint64_t r0 = s * 3 / 8, r1 = s * 5 / 8;
  return (part == 0 ? r0 : part == 1 ? r1 - r0 : s - r1) * s;
}

// A level whose texels take less time to filter than to send is not worth sharding: every shard filters it for itself
// and it never crosses a link.  At roughness 0 a texel is one sample (a copy of its source pixel): one MI355X writes
// 90 G such texels a second (2048^2 x 6 in 0.28 ms), 1.4 TB/s of output, more than its seven xGMI links carry
// together, while a texel of 1000 samples is produced at 9 GB/s.  The break-even lies near ten samples per texel; of
// the reference's roughness schedule (renderer.ts:1559) only roughness 0, i.e. level 0, is below it.
constexpr int PREFILTER_REPLICATE_MAX_SAMPLES = 8;
int prefilter_first_sharded_level(const PrefilterCost& pc, int mips, int n) {
  if (n <= 1) return 0;
  int first = 0;
  while (first < mips && pc.kept[first] <= PREFILTER_REPLICATE_MAX_SAMPLES) ++first;
  return first;
}

// How n shards divide the map.  cuts[0] texels at the front of the flattened (mip, face, y, x) order - the levels
// every shard filters for itself, see above; 0 for a single shard - belong to no shard; shard k owns texels
// [cuts[k], cuts[k + 1]), ranges of equal cost; cuts[n] = all texels.
// The reference filters level after level (renderer.ts:1558-1609); level 0 holds three quarters of the texels
// but, at roughness 0, almost none of the work, so equal texel counts would leave most shards idle.
void prefilter_cuts(int size, int mips, int n, uint64_t* cuts) {
  PrefilterCost pc;
  prefilter_costs(mips, &pc);
  const int first_level = prefilter_first_sharded_level(pc, mips, n);
  uint64_t total_cost = 0, total_texels = 0, replicated_texels = 0;
  for (int m = 0; m < mips; ++m) {
    const uint64_t s = (uint64_t)std::max(1, size >> m);
    if (m < first_level) replicated_texels += 6 * s * s;
    else for (int f = 0; f < 6; ++f) for (int part = 0; part < 3; ++part) total_cost += prefilter_part_texels(s, part) * prefilter_texel_cost(pc, size, mips, m, f, part);
    total_texels += 6 * s * s;
  }
  cuts[0] = replicated_texels; cuts[n] = total_texels;
  for (int k = 1; k < n; ++k) {
    const uint64_t target = (uint64_t)(((unsigned __int128)total_cost * (unsigned)k) / (unsigned)n);
    uint64_t before_cost = 0, before_texels = replicated_texels;
    bool placed = false;
    for (int m = first_level; m < mips && !placed; ++m) {
      const uint64_t s = (uint64_t)std::max(1, size >> m);
      for (int f = 0; f < 6 && !placed; ++f)
        for (int part = 0; part < 3 && !placed; ++part) {
          const uint64_t part_texels = prefilter_part_texels(s, part);
          if (part_texels == 0) continue;
          const uint64_t per_texel = prefilter_texel_cost(pc, size, mips, m, f, part), c = part_texels * per_texel;
          if (before_cost + c > target) { cuts[k] = before_texels + (target - before_cost) / per_texel; placed = true; }
          else { before_cost += c; before_texels += part_texels; }
        }
    }
    if (!placed) cuts[k] = total_texels;
  }
}

// The skybox pass unprojects every pixel centre: x_ndc = (x + 0.5) 2 / W - 1, y_ndc = 1 - (y + 0.5) 2 / H, then
// inv_view_proj * (x_ndc, y_ndc, 1, 1) (skybox.wgsl:13-27).  That is affine in the pixel coordinates; the three vectors
// are worked out here in double precision and rounded once.
// Returns false when a coefficient is not finite: the skybox lookups do not clamp their texel coordinates (see
// cube_foot_pair), which is safe for every finite unprojection and for none that is not.
bool sky_affine(const float* m, int W, int H, FrameParams& fp) {
  bool finite = true;
  for (int r = 0; r < 4; ++r) {
    const double c0 = m[r], c1 = m[4 + r], c2 = m[8 + r], c3 = m[12 + r];
    fp.sky_a[r] = (float)(c0 * (1.0 / W - 1.0) + c1 * (1.0 - 1.0 / H) + c2 + c3);
    fp.sky_bx[r] = (float)(c0 * 2.0 / W);
    fp.sky_by[r] = (float)(-c1 * 2.0 / H);
    finite = finite && std::isfinite(fp.sky_a[r]) && std::isfinite(fp.sky_bx[r]) && std::isfinite(fp.sky_by[r]);
  }
  return finite;
}

// the context that owns the scene (mesh, textures, IBL maps): the context itself, or the one it was forked from
PbrContext* assets_of(PbrContext* c) { return c->parent ? c->parent : c; }

// outputs: also the context's own colour target and visibility / depth buffers (a view slot has none: its view's
// colours go to the caller's target and nothing else is written)
int ensure_frame_buffers(PbrContext* c, int W, int rows, int n_tiles, int samples, bool outputs = true) {
  const size_t ntri = (size_t)(assets_of(c)->n_indices / 3) * assets_of(c)->n_instances;
  bool grew = false;
  if (ntri > c->record_cap) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_dev(c->d_records);
    HIP_TRY(c, hipMalloc(&c->d_records, ntri * sizeof(TriRecord)));
    c->record_cap = ntri; grew = true;
  }
  // Set-up triangles: every triangle that survives culling, plus the pieces of clipped ones.  A quarter more than the
  // mesh is ample for views from outside; a triangle cut by several planes leaves up to seven pieces, so a view from
  // inside a dense scene can exceed it.  The kernels then drop what does not fit and raise a flag; the library, when it
  // sees the flag, enlarges the buffer to that worst case (want_capacity) and the frame is drawn again.
  const size_t want_setup = c->forced_setup_cap ? c->forced_setup_cap : std::max(ntri + ntri / 4 + 4096, c->min_setup_cap);
  if (want_setup > c->setup_cap || (c->forced_setup_cap && want_setup != c->setup_cap)) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_dev(c->d_setup);
    HIP_TRY(c, hipMalloc(&c->d_setup, want_setup * sizeof(SetupTri)));
    c->setup_cap = want_setup;
    grew = true;
  }
  const size_t want_overflow = c->forced_overflow_cap ? c->forced_overflow_cap : std::max(want_setup * 2 + (1u << 16), c->min_overflow_cap);
  if (want_overflow > c->overflow_cap || (c->forced_overflow_cap && want_overflow != c->overflow_cap)) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_dev(c->d_overflow);
    HIP_TRY(c, hipMalloc(&c->d_overflow, want_overflow * sizeof(uint2)));
    c->overflow_cap = want_overflow;
    grew = true;
  }
  if ((size_t)n_tiles > c->tile_cap) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_dev(c->d_tile_count); free_dev(c->d_tile_live); free_dev(c->d_bins); free_dev(c->quad_lists.uniform); free_dev(c->quad_lists.mixed_pos); free_dev(c->quad_lists.mixed_ids);
    free_dev(c->d_live_info); free_dev(c->d_live_tiles);
    const size_t n_quads = quad_list_len((size_t)n_tiles);
    HIP_TRY(c, hipMalloc(&c->quad_lists.uniform, sizeof(uint2) * n_quads));
    HIP_TRY(c, hipMalloc(&c->quad_lists.mixed_pos, sizeof(uint32_t) * n_quads));
    HIP_TRY(c, hipMalloc(&c->quad_lists.mixed_ids, sizeof(uint4) * n_quads));
    HIP_TRY(c, hipMalloc(&c->d_tile_count, sizeof(uint32_t) * n_tiles));
    HIP_TRY(c, hipMalloc(&c->d_tile_live, sizeof(uint32_t) * n_tiles));
    HIP_TRY(c, hipMalloc(&c->d_bins, sizeof(uint32_t) * (size_t)n_tiles * PBR_BIN_CAP));
    HIP_TRY(c, hipMemset(c->d_tile_count, 0, sizeof(uint32_t) * n_tiles));   // kept clear from frame to frame by the raster kernel
    HIP_TRY(c, hipMalloc(&c->d_live_info, sizeof(unsigned long long) * n_tiles));
    HIP_TRY(c, hipMalloc(&c->d_live_tiles, sizeof(unsigned long long) * PBR_LIVE_TILE_CAP));
    HIP_TRY(c, hipMemset(c->d_live_tiles, 0, sizeof(unsigned long long) * PBR_LIVE_TILE_CAP));
    HIP_TRY(c, hipMemset(c->d_live_info, 0, sizeof(unsigned long long) * n_tiles));   // serial 0: no frame
    c->tile_cap = n_tiles; grew = true;
  }
  const size_t npix = outputs ? (size_t)W * rows : 0;
  if (npix > c->pix_cap) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_dev(c->d_color);
    HIP_TRY(c, hipMalloc(&c->d_color, npix * 4));
    c->pix_cap = npix; grew = true;
  }
  if (npix * samples > c->vis_cap) {   // one entry per sample
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_dev(c->d_vis); free_dev(c->d_depth);
    HIP_TRY(c, hipMalloc(&c->d_vis, npix * samples * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc(&c->d_depth, npix * samples * sizeof(float)));
    c->vis_cap = npix * samples; grew = true;
  }
  (void)grew;
  return PBR_OK;
}

}  // namespace

// State that lives behind the context; keeping it out of the header keeps PbrContext plain.
struct PbrContextPrivate {
  float* d_srgb_lut = nullptr;
  int prefilter_tables_for = -1;          // mips the cached sample tables were built for
  int prefilter_src_levels = -1;
  int table_count[PBR_MAX_LEVELS] = {};
  int table_single[PBR_MAX_LEVELS] = {};   // leading samples of each table that read one source level
  float table_weight[PBR_MAX_LEVELS] = {};
  uint32_t frames_drawn = 0;
  // Bordered copies of the cube maps for the resolve: every face carries the adjoining texels of its
  // neighbours (and the three-texel mean at corners), so a bilinear lookup never leaves its face.
  struct Bordered { float* d = nullptr; uint4* d_pairs = nullptr; DevCubeRgb cube = {}; bool dirty = true; };
  Bordered sky_b, irr_b, pre_b;
  uint4* d_lut_quad = nullptr; int lut_quad_size = 0; bool lut_quad_dirty = true;   // footprint copy of the BRDF table for the resolve (ResolveInputs::lut_quad)
  uint4* d_atlas = nullptr; size_t atlas_texels = 0; bool atlas_dirty = true;       // interleaved copy of five equally sized material textures (ResolveInputs::atlas)
  // Rebuilds of the lookup copies above are queued on the owner's stream without waiting; this event marks the end of the
  // last one, and every frame drawn on another stream (a fork's, or a stream set later) waits for it first.
  hipEvent_t copies_ready = nullptr; hipStream_t copies_stream = nullptr; bool copies_pending = false;
  int surface_blocks = 0;                 // resident workgroups of the surface kernel on this device
  int counter_set = 0;                    // which of the two sets of append counters the next frame uses
  bool used_static = false;               // pbr_draw_frame_static has run: its overflow flag lives in the device's fixed array
  // pbr_draw_views_device: hidden contexts that lend their working buffers and counters to views 1, 2, ... of a pass
  // (view 0 works in the context's own)
  std::vector<PbrContext*> view_slots;
};
static PbrContextPrivate* priv_of(PbrContext* c);

namespace {
struct CtxBox { PbrContext pub; PbrContextPrivate priv; };
}
static PbrContextPrivate* priv_of(PbrContext* c) { return &reinterpret_cast<CtxBox*>(c)->priv; }

static int check_sticky(PbrContext* c) {
  // counter [2] of either set is raised by the binning kernels when a capacity was exceeded
  for (int set = 0; set < 2; ++set) {
    if (c->h_counters && c->h_counters[8 * set + 2]) {
      const uint32_t code = c->h_counters[8 * set + 2];
      c->h_counters[8 * set + 2] = 0;
      (void)hipMemset(&c->d_counters[8 * set + 2], 0, sizeof(uint32_t));
      // enlarge for the frames to come: the worst case for the set-up buffer, four times the list for the tile references
      const size_t ntri = (size_t)(assets_of(c)->n_indices / 3) * assets_of(c)->n_instances;
      if (code == 1) { c->min_setup_cap = 7 * ntri + 4096; c->forced_setup_cap = 0; }
      else { c->min_overflow_cap = std::max(c->overflow_cap, (size_t)1 << 16) * 4; c->forced_overflow_cap = 0; }
      c->enlarged_after_overflow = true;
      return fail(c, PBR_E_STATE, code == 1 ? "triangle setup buffer overflow: a frame is incomplete; the buffer has been enlarged, draw it again"
                                            : "tile bin overflow list full: a frame is incomplete; the list has been enlarged, draw it again");
    }
  }
  return PBR_OK;
}

static int refresh_bordered(PbrContext* c, PbrContextPrivate::Bordered& b, const DevCube& src, bool with_pairs = false) {
  if (!b.dirty && b.d) return PBR_OK;
  if (!b.d || (with_pairs && !b.d_pairs) || b.cube.size != src.size || b.cube.levels != src.levels) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (b.d) { (void)hipFree(b.d); b.d = nullptr; }
    if (b.d_pairs) { (void)hipFree(b.d_pairs); b.d_pairs = nullptr; b.cube.pairs = nullptr; }
    size_t total = 0;
    for (int l = 0; l < src.levels; ++l) {
      const size_t n = (size_t)std::max(1, src.size >> l);
      b.cube.off[l] = (uint32_t)total;
      total += 6 * (n + 2) * (n + 2);
    }
    if (total > 0xffffffffull) return fail(c, PBR_E_ARG, "cube map too large");
    HIP_TRY(c, hipMalloc(&b.d, total * 3 * sizeof(float)));
    b.cube.base = b.d; b.cube.size = src.size; b.cube.levels = src.levels;
    if (with_pairs) {
      const size_t m = (size_t)std::max(1, src.size) + 2;
      HIP_TRY(c, hipMalloc(&b.d_pairs, 6 * m * m * sizeof(uint4)));
      b.cube.pairs = b.d_pairs;
    }
  }
  for (int l = 0; l < src.levels; ++l) launch_cube_border(c->stream, src, l, b.d + 3 * (size_t)b.cube.off[l]);
  if (with_pairs) launch_cube_pairs(c->stream, b.d + 3 * (size_t)b.cube.off[0], b.d_pairs, src.size);
  b.dirty = false;
  return PBR_OK;
}

// The lookup copies derived from the IBL maps and material textures (bordered cubes, BRDF footprint table, interleaved
// textures) belong to the scene's owner and are rebuilt on its stream, whoever finds them stale; the rebuild is only
// queued.  Frames on the owner's stream follow it in stream order; frames on any other stream (the forks') wait for the
// event recorded behind it (wait_for_lookup_copies), whether they triggered the rebuild or the owner's own first frame did.
static int refresh_lookup_copies(PbrContext* c, PbrContext* a, PbrContextPrivate* apr, bool one_size) {
  int rc = PBR_OK;
  if ((rc = refresh_bordered(a, apr->sky_b, a->sky, true)) || (rc = refresh_bordered(a, apr->irr_b, a->irr)) || (rc = refresh_bordered(a, apr->pre_b, a->pre))) return rc;
  if (apr->lut_quad_dirty || apr->lut_quad_size != a->lut_size) {
    if (apr->lut_quad_size != a->lut_size) {
      HIP_TRY(c, hipStreamSynchronize(a->stream));
      if (apr->d_lut_quad) { (void)hipFree(apr->d_lut_quad); apr->d_lut_quad = nullptr; }
      HIP_TRY(c, hipMalloc(&apr->d_lut_quad, (size_t)a->lut_size * a->lut_size * sizeof(uint4)));
      apr->lut_quad_size = a->lut_size;
    }
    launch_lut_quad(a->stream, a->d_lut, apr->d_lut_quad, a->lut_size);
    apr->lut_quad_dirty = false;
  }
  if (one_size && (apr->atlas_dirty || !apr->d_atlas)) {
    size_t texels;
    uint32_t off[PBR_MAX_LEVELS];
    tex8_offsets(a->tex[0].w, a->tex[0].h, a->tex[0].levels, off, &texels);
    if (apr->atlas_texels != texels || !apr->d_atlas) {
      HIP_TRY(c, hipStreamSynchronize(a->stream));
      if (apr->d_atlas) { (void)hipFree(apr->d_atlas); apr->d_atlas = nullptr; }
      HIP_TRY(c, hipMalloc(&apr->d_atlas, texels * sizeof(uint4)));
      apr->atlas_texels = texels;
    }
    const uint32_t* chains[5];
    for (int i = 0; i < 5; ++i) chains[i] = a->tex[i].base;
    launch_material_atlas(a->stream, chains, apr->d_atlas, texels);
    apr->atlas_dirty = false;
  }
  if (!apr->copies_ready) HIP_TRY(c, hipEventCreateWithFlags(&apr->copies_ready, hipEventDisableTiming));
  HIP_TRY(c, hipEventRecord(apr->copies_ready, a->stream));
  apr->copies_stream = a->stream; apr->copies_pending = true;
  return PBR_OK;
}

// Orders this context's next launches behind the last rebuild of the lookup copies when that ran on another stream.
static int wait_for_lookup_copies(PbrContext* c, PbrContextPrivate* apr) {
  if (!apr->copies_pending || c->stream == apr->copies_stream) return PBR_OK;
  const hipError_t q = hipEventQuery(apr->copies_ready);
  if (q == hipSuccess) { apr->copies_pending = false; return PBR_OK; }
  if (q != hipErrorNotReady) HIP_TRY(c, q);
  HIP_TRY(c, hipStreamWaitEvent(c->stream, apr->copies_ready, 0));
  return PBR_OK;
}

// The memory a frame works in between its launches: the context's own buffers (grown on demand) or, for
// pbr_draw_frame_static, slices of the device's fixed scratch array.
struct FrameWork {
  SetupTri* setup; size_t setup_cap;
  TriRecord* records;
  uint32_t* tile_count; uint32_t* bins;
  LiveTiles live;
  uint2* overflow; size_t overflow_cap;
  QuadLists lists;
  uint32_t* counters; uint32_t* next_counters;
};

#ifndef PBR_SKY_SHARE_SETUP
#define PBR_SKY_SHARE_SETUP 0.35f
#endif
#ifndef PBR_SKY_SHARE_SETUP_ALONE
#define PBR_SKY_SHARE_SETUP_ALONE 0.65f
#endif
// The share for the passes of several views (pbr_draw_views_device): their launches each fill the chip, and a short
// set-up launch lets the next pass start under the tail of this one (49.4 / 49.5 / 50.3 / 50.4 / 51.5 / 52.6 / 53.8 us
// per 4K frame at 0 / 0.10 / 0.20 / 0.35 / 0.50 / 0.65 / 0.80, three passes of four views in flight).
#ifndef PBR_SKY_SHARE_SETUP_PASS
#define PBR_SKY_SHARE_SETUP_PASS 0.10f
#endif
static int draw_common(PbrContext* c, const PbrFrameUniforms* u, int W, int H, int y0, int y1, void* d_rgba8, void* d_vis, bool want_host_vis,
                       PbrFrameStats* stats, bool static_memory = false) {
  PbrContext* a = assets_of(c);              // scene data may belong to the context this one was forked from
  PbrContextPrivate* apr = priv_of(a);
  if (!u) return fail(c, PBR_E_ARG, "uniforms pointer is null");
  if (W <= 0 || H <= 0 || W > 16384 || H > 16384) return fail(c, PBR_E_ARG, "bad target size %dx%d", W, H);
  if (y0 < 0 || y1 > H || y0 >= y1 || (y0 & 1) || ((y1 & 1) && y1 != H)) return fail(c, PBR_E_ARG, "bad row range [%d,%d) for height %d (rows must start and end on even lines)", y0, y1, H);
  if (u->sampler_index & ~31u) return fail(c, PBR_E_ARG, "sampler index %u: the reference builds 32 samplers (renderer.ts:822-859)", u->sampler_index);
  if (!a->d_vsoa) return fail(c, PBR_E_STATE, "no mesh uploaded");
  for (int i = 0; i < 5; ++i) if (!a->d_tex[i]) return fail(c, PBR_E_STATE, "material texture slot %d not uploaded", i);
  if (!a->d_irr) return fail(c, PBR_E_STATE, "irradiance map not generated");
  if (!a->d_pre) return fail(c, PBR_E_STATE, "prefilter map not generated");
  if (!a->d_lut) return fail(c, PBR_E_STATE, "BRDF LUT not generated");
  if (!a->d_sky) return fail(c, PBR_E_STATE, "skybox cube not generated");
  if (a->has_instance_mats == false && a->n_instances != 1) return fail(c, PBR_E_STATE, "instanced mesh without instance matrices");
  DeviceGuard guard(c->device);
  PbrContextPrivate* pr = priv_of(c);

  const int rows = y1 - y0;
  const int tiles_x = (W + PBR_TILE - 1) / PBR_TILE, tiles_y = (rows + PBR_TILE - 1) / PBR_TILE;
  // bit 16 of the sampler index selects the multisampled pipeline (renderer.ts:1927-1932, :1951-1954, :1979-1980)
  const bool msaa = (u->sampler_index & 16u) != 0u;
  const int samples = msaa ? 4 : 1;
  // the specialised resolve covers the reference's default sampler on five power-of-two textures of one size, and reads
  // them from one interleaved copy
  static const int kFormatSrgb[5] = {1, 0, 0, 1, 0};   // renderer.ts:647, :663, :681, :697, :713
  bool one_size = true;
  for (int i = 0; i < 5; ++i)
    one_size = one_size && a->tex[i].base && a->tex[i].srgb == kFormatSrgb[i] && (a->tex[i].w & (a->tex[i].w - 1)) == 0 && (a->tex[i].h & (a->tex[i].h - 1)) == 0 &&
               a->tex[i].w == a->tex[0].w && a->tex[i].h == a->tex[0].h && a->tex[i].levels == a->tex[0].levels;
  const bool fast = one_size && (u->sampler_index & 15u) == 15u;
  const bool stale = apr->sky_b.dirty || !apr->sky_b.d || apr->irr_b.dirty || !apr->irr_b.d || apr->pre_b.dirty || !apr->pre_b.d || apr->lut_quad_dirty ||
                     apr->lut_quad_size != a->lut_size || (one_size && (apr->atlas_dirty || !apr->d_atlas));
  int rc = PBR_OK;
  FrameWork fw;
  if (static_memory) {
    // nothing may be allocated on this path: the frame has to fit the fixed array, and everything derived from the
    // IBL maps has to exist already
    using namespace static_layout;
    const size_t ntri = (size_t)(a->n_indices / 3) * a->n_instances;
    if (msaa) return fail(c, PBR_E_UNSUPPORTED, "pbr_draw_frame_static renders the single-sample pipeline only");
    if (tiles_x * tiles_y > PBR_STATIC_MAX_TILES)
      return fail(c, PBR_E_ARG, "pbr_draw_frame_static: %d x %d pixels are %d tiles of 32 x 32, the fixed working memory holds %d (e.g. 1920 x 1080)", W, rows,
                  tiles_x * tiles_y, PBR_STATIC_MAX_TILES);
    if (ntri > PBR_STATIC_MAX_TRIANGLES)
      return fail(c, PBR_E_ARG, "pbr_draw_frame_static: %zu triangles (all instances), the fixed working memory holds %d", ntri, PBR_STATIC_MAX_TRIANGLES);
    if (c->device < 0 || c->device >= kMaxDevices || !g_scratch_ptr[c->device]) return fail(c, PBR_E_STATE, "no fixed working memory on device %d", c->device);
    if (stale)
      return fail(c, PBR_E_STATE, "pbr_draw_frame_static: the lookup copies of the IBL maps and material textures are not built yet; draw one frame with "
                                  "pbr_draw_frame_device after the last IBL or texture call");
    unsigned char* base = g_scratch_ptr[c->device];
    fw.counters = reinterpret_cast<uint32_t*>(base + kCounters); fw.next_counters = fw.counters + 8;
    fw.tile_count = reinterpret_cast<uint32_t*>(base + kTileCount);
    fw.live = LiveTiles{reinterpret_cast<unsigned long long*>(base + kLiveTiles), reinterpret_cast<unsigned long long*>(base + kLiveInfo), PBR_LIVE_TILE_CAP, 1ull};   // (both arrays were just cleared: any serial but 0)
    fw.bins = reinterpret_cast<uint32_t*>(base + kBins);
    fw.setup = reinterpret_cast<SetupTri*>(base + kSetup); fw.setup_cap = kSetupCap;
    fw.overflow = reinterpret_cast<uint2*>(base + kOverflow); fw.overflow_cap = kOverflowCap;
    fw.records = reinterpret_cast<TriRecord*>(base + kRecords);
    fw.lists.uniform = reinterpret_cast<uint2*>(base + kUniform);
    fw.lists.mixed_pos = reinterpret_cast<uint32_t*>(base + kMixedPos);
    fw.lists.mixed_ids = reinterpret_cast<uint4*>(base + kMixedIds);
    // every call on this GPU shares the array and finds the previous call's contents in it: the append counters and the
    // per-tile reference counts must start at zero (the lists and records are written before they are read)
    HIP_TRY(c, hipMemsetAsync(base, 0, kZeroedBytes, c->stream));
    pr->used_static = true;
  } else {
    rc = ensure_frame_buffers(c, W, rows, tiles_x * tiles_y, samples);
    if (rc) return rc;
    fw.setup = c->d_setup; fw.setup_cap = c->setup_cap; fw.records = c->d_records; fw.tile_count = c->d_tile_count; fw.bins = c->d_bins;
    fw.live = LiveTiles{c->d_live_tiles, c->d_live_info, PBR_LIVE_TILE_CAP, ++c->live_serial};
    fw.overflow = c->d_overflow; fw.overflow_cap = c->overflow_cap; fw.lists = c->quad_lists;
    // two sets of append counters, used by alternate frames (the rasteriser re-arms the next frame's set)
    fw.counters = c->d_counters + 8 * pr->counter_set;
    fw.next_counters = c->d_counters + 8 * (1 - pr->counter_set);
  }

  // per-frame parameters travel as kernel arguments
  FrameParams fp;
  fp.u = *u;
  mat4_mul_host(u->projection, u->view, fp.pv);
  fp.W = W; fp.H = H; fp.y0 = y0; fp.y1 = y1; fp.tiles_x = tiles_x; fp.tiles_y = tiles_y;
  fp.tiles_x_magic = (uint32_t)((0x100000000ull + (uint64_t)tiles_x - 1) / (uint64_t)tiles_x);
  fp.n_verts = a->n_verts; fp.n_tris = a->n_indices / 3; fp.n_instances = a->n_instances;
  fp.has_instance_mats = a->has_instance_mats ? 1 : 0;
  fp.msaa = msaa ? 1 : 0;
  fp.two_over_w = 2.0f / (float)W; fp.two_over_h = 2.0f / (float)H;
  if (!sky_affine(u->inv_view_proj, W, H, fp)) return fail(c, PBR_E_ARG, "the inverse view-projection matrix (skybox uniform, renderer.ts:1864-1867) is not finite");
  const FrameParams& dp = fp;
  hipStream_t s = c->stream;

  if (!static_memory && stale && (rc = refresh_lookup_copies(c, a, apr, one_size))) return rc;
  if ((rc = wait_for_lookup_copies(c, apr))) return rc;

  const bool timed = stats != nullptr;
  const bool prof = c->profiling && c->prof_frames < c->prof_cap;
  const int prof_stride = c->prof_all ? 4 : 2;
  hipEvent_t* pe = prof ? &c->prof_events[prof_stride * c->prof_frames] + (prof_stride - 2) : nullptr;   // surface pass begin, end
  if (prof && c->prof_all) HIP_TRY(c, hipEventRecord(pe[-2], s));
  if (timed) HIP_TRY(c, hipEventRecord(c->ev[0], s));
  uint8_t* color = static_cast<uint8_t*>(d_rgba8);

  uint32_t* counters = fw.counters;
  uint32_t* next_counters = fw.next_counters;
  const int this_set = pr->counter_set;
  if (!static_memory) pr->counter_set = 1 - pr->counter_set;

  // The skybox tiles are shared out over the frame's geometry launches: those kernels are
  // latency-bound and leave the chip almost idle, so each carries a slice of the tiles as extra
  // workgroups; the rasteriser shades the rest (each workgroup its own tile).
  const int n_tiles = tiles_x * tiles_y;
  // Skybox tiles that ride in the setup launch; the rasteriser's launch shades the rest.  With frames in flight a
  // third is best (52.6 / 52.8 / 53.9 / 54.2 us per 4K frame at 0.35 / 0.45 / 0.55 / 0.65).  A context on its own
  // wants two thirds there: its rasteriser's launch lasts as long as its skybox tiles do, while the set-up launch
  // waits on its triangles (single frame 80.2 / 79.1 / 78.2 / 78.4 us at 0.45 / 0.55 / 0.65 / 0.75, with the rows of
  // tiles handed out from the middle so that the tiles with triangles no longer end the launch).  Measurement builds set
  // other shares with -DPBR_SKY_SHARE_SETUP=... / -DPBR_SKY_SHARE_SETUP_ALONE=...; the library reads no environment
  // variable.
  const bool alone = c->parent == nullptr && c->children == 0;   // no forks: nothing else of this scene is in flight
  const float share_setup = alone ? PBR_SKY_SHARE_SETUP_ALONE : PBR_SKY_SHARE_SETUP;
  // A frame has no vertex launch: the setup kernel transforms each triangle's vertices from the streams itself
  // and leaves a shading record per triangle, which is all the resolve kernels read.
  // The multisampled resolve blends over the skybox colour, which therefore has to be complete before the
  // rasteriser: all its tiles ride in the setup launch.
  SkyShare sky_b;
  sky_b.sky = apr->sky_b.cube; sky_b.rgba8 = reinterpret_cast<uint32_t*>(color); sky_b.first = 0; sky_b.own_blocks = 0;
  sky_b.count = msaa ? n_tiles : std::min(n_tiles, std::max(0, (int)(n_tiles * share_setup)));
  const int sky_from_tile = sky_b.first + sky_b.count;
  const VertexSource source = {nullptr, a->d_vsoa, a->d_inst_model, a->d_inst_normal, {a->mesh_lo[0], a->mesh_lo[1], a->mesh_lo[2]}, {a->mesh_hi[0], a->mesh_hi[1], a->mesh_hi[2]}};
  if (timed) HIP_TRY(c, hipEventRecord(c->ev[1], s));
  launch_setup_and_bin(s, dp, source, a->d_indices, fw.setup, fw.records, (uint32_t)fw.setup_cap, fw.tile_count, fw.bins, fw.overflow,
                       (uint32_t)fw.overflow_cap, counters, fw.live, a->n_indices / 3, a->n_instances, sky_b);
  if (timed) HIP_TRY(c, hipEventRecord(c->ev[2], s));
  if (prof && c->prof_all) HIP_TRY(c, hipEventRecord(pe[-1], s));
  uint32_t* vis = d_vis ? static_cast<uint32_t*>(d_vis) : c->d_vis;
  // the visibility buffer is only written when someone will read it: the caller (vis output) or the
  // coverage count of the statistics
  const bool write_vis = d_vis != nullptr || want_host_vis || timed;
  if (msaa)
    launch_raster_msaa(s, dp, fw.setup, fw.tile_count, fw.bins, fw.overflow, (uint32_t)fw.overflow_cap, vis, c->capture_depth ? c->d_depth : nullptr,
                       counters, next_counters, c->d_tile_live, write_vis || c->capture_depth, tiles_x, tiles_y, fw.live);
  else
    launch_raster(s, dp, fw.setup, fw.tile_count, fw.bins, fw.overflow, (uint32_t)fw.overflow_cap, vis, (c->capture_depth && !static_memory) ? c->d_depth : nullptr,
                  counters, next_counters, fw.lists, write_vis, apr->sky_b.cube, color, sky_from_tile, tiles_x, tiles_y, fw.live,
                  /*split_listed=*/alone);   // see the kernel
  if (timed) HIP_TRY(c, hipEventRecord(c->ev[3], s));

  ResolveInputs in;
  in.records = fw.records; in.vout = a->d_vout; in.indices = a->d_indices;
  for (int i = 0; i < 5; ++i) in.tex[i] = a->tex[i];
  in.irr = a->irr; in.pre = a->pre; in.sky = a->sky;
  in.irr_b = apr->irr_b.cube; in.pre_b = apr->pre_b.cube; in.sky_b = apr->sky_b.cube;
  in.lut = a->d_lut; in.lut_quad = apr->d_lut_quad; in.atlas = fast ? apr->d_atlas : nullptr; in.lut_size = a->lut_size;
  if (prof) HIP_TRY(c, hipEventRecord(pe[0], s));
  if (msaa) launch_surface_msaa(s, dp, in, pr->d_srgb_lut, vis, c->d_tile_live, color, W, rows, fast);
  // a context that shares its scene with forks is one of several frames in flight on this chip (see the two shapes of the surface kernel)
  else launch_surface(s, dp, in, pr->d_srgb_lut, vis, color, fw.lists, counters, W, rows, tiles_x, tiles_y, fast, pr->surface_blocks, /*shared_chip=*/!alone);
  if (timed) HIP_TRY(c, hipEventRecord(c->ev[4], s));
  if (prof) { HIP_TRY(c, hipEventRecord(pe[1], s)); ++c->prof_frames; }
  HIP_TRY(c, hipGetLastError());
  c->last_w = W; c->last_rows = rows; c->last_samples = samples;
  ++pr->frames_drawn;

  if (timed) {
    HIP_TRY(c, hipMemsetAsync(&counters[3], 0, sizeof(uint32_t), s));
    launch_count_covered(s, vis, (size_t)W * rows, samples, &counters[3]);
    HIP_TRY(c, hipMemcpyAsync(c->h_counters, c->d_counters, kCounterBytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    (void)hipEventElapsedTime(&stats->ms_vertex, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&stats->ms_setup, c->ev[1], c->ev[2]);
    (void)hipEventElapsedTime(&stats->ms_raster, c->ev[2], c->ev[3]);
    (void)hipEventElapsedTime(&stats->ms_resolve, c->ev[3], c->ev[4]);
    (void)hipEventElapsedTime(&stats->ms_total, c->ev[0], c->ev[4]);
    const uint32_t* hc = c->h_counters + 8 * this_set;
    stats->setup_triangles = hc[0];   // pieces of clipped triangles; the whole ones are counted per segment
    stats->binned_refs = hc[1];       // references that spilled past a tile's own row
    stats->covered_pixels = hc[3];
    stats->uniform_quads = 0; stats->mixed_quads = 0;
    for (int seg = 0; seg < PBR_LIST_SEGMENTS; ++seg) {
      const uint32_t* group = hc + PBR_SEGMENT_FIRST + PBR_SEGMENT_WORDS * seg;
      stats->uniform_quads += group[0]; stats->mixed_quads += group[1]; stats->setup_triangles += group[2];
    }
    return check_sticky(c);
  }
  return PBR_OK;
}

extern "C" {

int pbr_create(int device_ordinal, PbrContext** ctx_out) {
  if (!ctx_out) return fail(nullptr, PBR_E_ARG, "ctx_out is null");
  *ctx_out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) return fail(nullptr, PBR_E_DEVICE, "no HIP device available (%s)", e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device_ordinal < 0 || device_ordinal >= count) return fail(nullptr, PBR_E_ARG, "device ordinal %d out of range (0..%d)", device_ordinal, count - 1);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return fail(nullptr, PBR_E_DEVICE, "cannot query device %d", device_ordinal);
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, PBR_E_DEVICE, "device %d is %s; this library carries gfx950 code only", device_ordinal, prop.gcnArchName);
  CtxBox* box = new CtxBox();
  PbrContext* c = &box->pub;
  c->device = device_ordinal;
  DeviceGuard guard(device_ordinal);
  auto bail = [&](const char* what, hipError_t err) {
    int rc = fail(nullptr, PBR_E_DEVICE, "%s failed: %s", what, hipGetErrorString(err));
    pbr_destroy(c);
    return rc;
  };
  if (device_ordinal < kMaxDevices && !g_scratch_ptr[device_ordinal]) {
    // once per device, with that device current: where this GPU keeps its copy of the fixed working memory
    void* p = nullptr;
    if ((e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch))) != hipSuccess) return bail("hipGetSymbolAddress(scratch)", e);
    g_scratch_ptr[device_ordinal] = static_cast<unsigned char*>(p);
  }
  if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  c->stream = c->own_stream;
  {
    PbrContextPrivate* pr = &box->priv;
    hipDeviceProp_t dp;
    if ((e = hipGetDeviceProperties(&dp, device_ordinal)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    pr->surface_blocks = dp.multiProcessorCount * surface_fast_blocks_per_cu();
  }
  for (auto& ev : c->ev) if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipMalloc(&c->d_counters, kCounterBytes)) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMemset(c->d_counters, 0, kCounterBytes)) != hipSuccess) return bail("hipMemset", e);
  if ((e = hipHostMalloc(&c->h_counters, kCounterBytes)) != hipSuccess) return bail("hipHostMalloc", e);
  std::memset(c->h_counters, 0, kCounterBytes);
  if ((e = hipMalloc(&c->d_samples, sizeof(PrefilterSample) * PBR_PREFILTER_SAMPLES * PBR_MAX_LEVELS)) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMalloc(&c->d_irr_tables, sizeof(float) * 2 * (512 + 128))) != hipSuccess) return bail("hipMalloc", e);
  {
    // sRGB transfer function of the rgba8unorm-srgb formats (renderer.ts:647, :697): decode table [0, 256), and from 256
    // on the store thresholds of the mip generator: code k is stored for T[k] <= v < T[k + 1], T[k] = the linear value
    // whose encoding is k - 1/2, from double precision, rounded up to the next float (the exactly rounded store; the
    // device's powf only proposes a candidate)
    float lut[256 + 257];
    for (int i = 0; i < 256; ++i) { float v = (float)i / 255.0f; lut[i] = v <= 0.04045f ? v / 12.92f : std::pow((v + 0.055f) / 1.055f, 2.4f); }
    lut[256] = 0.0f;
    for (int k = 1; k <= 255; ++k) {
      const double cc = ((double)k - 0.5) / 255.0;
      const double lin = cc <= 0.04045 ? cc / 12.92 : std::pow((cc + 0.055) / 1.055, 2.4);
      float f = (float)lin;
      if ((double)f < lin) f = std::nextafterf(f, 2.0f);
      lut[256 + k] = f;
    }
    lut[256 + 256] = 3.0e38f;
    if ((e = hipMalloc(&box->priv.d_srgb_lut, sizeof lut)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemcpy(box->priv.d_srgb_lut, lut, sizeof lut, hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy", e);
  }
  *ctx_out = c;
  return PBR_OK;
}

int pbr_fork(PbrContext* parent, PbrContext** ctx_out) {
  if (!ctx_out) return fail(parent, PBR_E_ARG, "ctx_out is null");
  *ctx_out = nullptr;
  if (!parent) return fail(nullptr, PBR_E_ARG, "parent context is null");
  if (parent->parent) return fail(parent, PBR_E_ARG, "fork the context that owns the scene, not one of its forks");
  PbrContext* child = nullptr;
  const int rc = pbr_create(parent->device, &child);
  if (rc) return fail(parent, rc, "%s", pbr_last_error(nullptr));
  child->parent = parent;
  ++parent->children;
  *ctx_out = child;
  return PBR_OK;
}

int pbr_destroy(PbrContext* c) {
  if (!c) return PBR_OK;
  if (c->children > 0) return fail(c, PBR_E_STATE, "%d forked context(s) still draw from this one: destroy them first", c->children);
  if (c->parent) { if (!c->view_slot) --c->parent->children; c->parent = nullptr; }
  DeviceGuard guard(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  PbrContextPrivate* pr = priv_of(c);
  for (PbrContext* slot : pr->view_slots) { slot->stream = slot->own_stream; (void)pbr_destroy(slot); }
  pr->view_slots.clear();
  free_dev(c->d_vsoa); free_dev(c->d_indices); free_dev(c->d_inst_model); free_dev(c->d_inst_normal);
  for (auto& t : c->d_tex) free_dev(t);
  free_dev(c->d_hdr); free_dev(c->d_hdr_rgb); free_dev(c->d_sky); free_dev(c->d_irr); free_dev(c->d_pre); free_dev(c->d_lut);
  { PbrContextPrivate* pr = priv_of(c); free_dev(pr->sky_b.d); free_dev(pr->sky_b.d_pairs); free_dev(pr->irr_b.d); free_dev(pr->pre_b.d); free_dev(pr->d_lut_quad); free_dev(pr->d_atlas); }
  free_dev(c->d_samples); free_dev(c->d_irr_tables);
  free_dev(c->d_vout); free_dev(c->d_setup); free_dev(c->d_records); free_dev(c->d_tile_count); free_dev(c->d_live_info); free_dev(c->d_live_tiles); free_dev(c->d_tile_live); free_dev(c->d_bins); free_dev(c->quad_lists.uniform); free_dev(c->quad_lists.mixed_pos); free_dev(c->quad_lists.mixed_ids);
  free_dev(c->d_overflow); free_dev(c->d_counters); free_dev(c->d_vis); free_dev(c->d_depth); free_dev(c->d_color);
  free_dev(pr->d_srgb_lut);
  if (pr->copies_ready) (void)hipEventDestroy(pr->copies_ready);
  if (c->h_counters) (void)hipHostFree(c->h_counters);
  for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : c->prof_events) if (ev) (void)hipEventDestroy(ev);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete reinterpret_cast<CtxBox*>(c);
  return PBR_OK;
}

const char* pbr_last_error(const PbrContext* c) { return c ? c->error.c_str() : g_create_error.c_str(); }

int pbr_set_stream(PbrContext* c, void* hip_stream) {
  if (!c) return PBR_E_ARG;
  DeviceGuard guard(c->device);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (priv_of(c)->copies_stream == c->stream) priv_of(c)->copies_pending = false;   // (a rebuild queued on the old stream has just completed)
  c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
  return PBR_OK;
}

int pbr_synchronize(PbrContext* c) {
  if (!c) return PBR_E_ARG;
  DeviceGuard guard(c->device);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(c->h_counters, c->d_counters, kCounterHeadBytes, hipMemcpyDeviceToHost));
  if (priv_of(c)->used_static && c->device < kMaxDevices && g_scratch_ptr[c->device]) {
    uint32_t flag = 0;   // the capacity flag of the last pbr_draw_frame_static on this device
    HIP_TRY(c, hipMemcpy(&flag, g_scratch_ptr[c->device] + static_layout::kCounters + 2 * sizeof(uint32_t), sizeof flag, hipMemcpyDeviceToHost));
    if (flag) return fail(c, PBR_E_STATE, flag == 1 ? "triangle setup buffer overflow" : "tile bin overflow list full");
  }
  for (PbrContext* slot : priv_of(c)->view_slots) {   // views 1, 2, ... of the multi-view passes count in their slots' counters
    HIP_TRY(c, hipMemcpy(slot->h_counters, slot->d_counters, kCounterHeadBytes, hipMemcpyDeviceToHost));
    if (check_sticky(slot)) return fail(c, PBR_E_STATE, "%s", slot->error.c_str());
  }
  return check_sticky(c);
}

// Both index widths end up as 32-bit indices on the device: the index stream is a few hundred kilobytes a
// frame against hundreds of megabytes of pixel traffic, and one width keeps one set of kernels.
static int upload_mesh_common(PbrContext* c, const float* verts, int32_t n_verts, const uint16_t* idx16, const uint32_t* idx32, int32_t n_indices,
                              int32_t instance_count, const float* instance_model, const float* instance_normal) {
  if (c && c->parent) return fail(c, PBR_E_STATE, "a forked context draws the scene of the context it was forked from and cannot change or read it back");
  if (!c) return PBR_E_ARG;
  if (!verts || (!idx16 && !idx32) || n_verts <= 0 || n_indices <= 0) return fail(c, PBR_E_ARG, "empty mesh");
  if (n_indices % 3) return fail(c, PBR_E_ARG, "index count %d is not a multiple of 3 (triangle-list)", n_indices);
  if (instance_count < 1) return fail(c, PBR_E_ARG, "instance count must be at least 1");
  if (instance_count > 1 && (!instance_model || !instance_normal)) return fail(c, PBR_E_ARG, "instanced upload needs model and normal matrices");
  if ((uint64_t)instance_count * (uint64_t)(n_indices / 3) >= 0xffffffffull) return fail(c, PBR_E_ARG, "too many triangles for 32-bit visibility ids");
  if ((uint64_t)instance_count * (uint64_t)n_verts > 0x7fffffffull) return fail(c, PBR_E_ARG, "too many vertices");
  std::vector<uint32_t> wide((size_t)n_indices);
  for (int32_t i = 0; i < n_indices; ++i) {
    const uint32_t v = idx16 ? (uint32_t)idx16[i] : idx32[i];
    if (v >= (uint32_t)n_verts) return fail(c, PBR_E_ARG, "index %d refers to vertex %u of %d", i, v, n_verts);
    wide[i] = v;
  }
  DeviceGuard guard(c->device);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  free_dev(c->d_vsoa); free_dev(c->d_indices); free_dev(c->d_inst_model); free_dev(c->d_inst_normal);
  std::vector<float> soa((size_t)n_verts * 8);
  for (int32_t i = 0; i < n_verts; ++i) {
    const float* v = verts + 8 * (size_t)i;   // pos3, uv2, normal3 (glTF.ts:193-213)
    soa[i] = v[0]; soa[(size_t)n_verts + i] = v[1]; soa[2 * (size_t)n_verts + i] = v[2];
    soa[3 * (size_t)n_verts + i] = v[3]; soa[4 * (size_t)n_verts + i] = v[4];
    soa[5 * (size_t)n_verts + i] = v[5]; soa[6 * (size_t)n_verts + i] = v[6]; soa[7 * (size_t)n_verts + i] = v[7];
  }
  HIP_TRY(c, hipMalloc(&c->d_vsoa, soa.size() * sizeof(float)));
  HIP_TRY(c, hipMemcpy(c->d_vsoa, soa.data(), soa.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMalloc(&c->d_indices, wide.size() * sizeof(uint32_t)));
  HIP_TRY(c, hipMemcpy(c->d_indices, wide.data(), wide.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  c->has_instance_mats = instance_model && instance_normal;
  if (c->has_instance_mats) {
    const size_t bytes = (size_t)instance_count * 16 * sizeof(float);
    HIP_TRY(c, hipMalloc(&c->d_inst_model, bytes));
    HIP_TRY(c, hipMalloc(&c->d_inst_normal, bytes));
    HIP_TRY(c, hipMemcpy(c->d_inst_model, instance_model, bytes, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_inst_normal, instance_normal, bytes, hipMemcpyHostToDevice));
  }
  c->n_verts = n_verts; c->n_indices = n_indices; c->n_instances = instance_count;
  // Bounding box of the positions, for skipping whole instances that miss the frame.  A position that is not finite
  // bounds nothing (its triangles are still drawn or dropped one by one, as the reference's would be): the mesh then
  // gets the "no box known" state, lo > hi, and no instance is ever skipped.
  bool all_finite = true;
  for (int k = 0; k < 3; ++k) { c->mesh_lo[k] = verts[k]; c->mesh_hi[k] = verts[k]; }
  for (int32_t i = 0; i < n_verts; ++i)
    for (int k = 0; k < 3; ++k) {
      const float v = verts[8 * (size_t)i + k];
      all_finite = all_finite && std::isfinite(v);
      if (v < c->mesh_lo[k]) c->mesh_lo[k] = v;
      if (v > c->mesh_hi[k]) c->mesh_hi[k] = v;
    }
  if (!all_finite) for (int k = 0; k < 3; ++k) { c->mesh_lo[k] = 1.0f; c->mesh_hi[k] = -1.0f; }
  return PBR_OK;
}

int pbr_upload_mesh(PbrContext* c, const float* verts, int32_t n_verts, const uint16_t* indices, int32_t n_indices,
                    int32_t instance_count, const float* instance_model, const float* instance_normal) {
  return upload_mesh_common(c, verts, n_verts, indices, nullptr, n_indices, instance_count, instance_model, instance_normal);
}

int pbr_upload_mesh_u32(PbrContext* c, const float* verts, int32_t n_verts, const uint32_t* indices, int32_t n_indices,
                        int32_t instance_count, const float* instance_model, const float* instance_normal) {
  return upload_mesh_common(c, verts, n_verts, nullptr, indices, n_indices, instance_count, instance_model, instance_normal);
}

int pbr_upload_texture(PbrContext* c, int slot, const uint8_t* rgba8, int32_t w, int32_t h, int32_t is_srgb, int32_t gen_mips) {
  if (c && c->parent) return fail(c, PBR_E_STATE, "a forked context draws the scene of the context it was forked from and cannot change or read it back");
  if (!c) return PBR_E_ARG;
  if (slot < 0 || slot > 4) return fail(c, PBR_E_ARG, "texture slot %d out of range", slot);
  if (!rgba8 || w <= 0 || h <= 0 || w > 16384 || h > 16384) return fail(c, PBR_E_ARG, "bad texture %dx%d", w, h);
  DeviceGuard guard(c->device);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  free_dev(c->d_tex[slot]);
  DevTex8& t = c->tex[slot];
  t.w = w; t.h = h; t.srgb = is_srgb ? 1 : 0;
  t.levels = gen_mips ? level_count(w, h) : 1;
  if (t.levels > PBR_MAX_LEVELS) return fail(c, PBR_E_ARG, "too many mip levels");
  size_t total;
  tex8_offsets(w, h, t.levels, t.off, &total);
  HIP_TRY(c, hipMalloc(&c->d_tex[slot], total * sizeof(uint32_t)));
  t.base = c->d_tex[slot];
  uint32_t* staging = nullptr;   // the row-major image, re-laid into blocks on the device
  HIP_TRY(c, hipMalloc(&staging, (size_t)w * h * 4));
  hipError_t up = hipMemcpyAsync(staging, rgba8, (size_t)w * h * 4, hipMemcpyHostToDevice, c->stream);
  if (up == hipSuccess) launch_build_mips_rgba8(c->stream, staging, c->d_tex[slot], t.off, w, h, t.levels, t.srgb, priv_of(c)->d_srgb_lut);
  if (up == hipSuccess) up = hipStreamSynchronize(c->stream);
  (void)hipFree(staging);
  HIP_TRY(c, up);
  HIP_TRY(c, hipGetLastError());
  c->tex_has_mips[slot] = t.levels > 1;
  priv_of(c)->atlas_dirty = true;
  return PBR_OK;
}

int pbr_upload_hdr(PbrContext* c, const float* rgba32f, int32_t w, int32_t h) {
  if (c && c->parent) return fail(c, PBR_E_STATE, "a forked context draws the scene of the context it was forked from and cannot change or read it back");
  if (!c) return PBR_E_ARG;
  if (!rgba32f || w <= 0 || h <= 0 || w > 32768 || h > 32768) return fail(c, PBR_E_ARG, "bad HDR %dx%d", w, h);
  DeviceGuard guard(c->device);
  if (level_count(w, h) > PBR_MAX_LEVELS) return fail(c, PBR_E_ARG, "too many mip levels");
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  // the old image goes first, whole: if an allocation below fails the context holds no HDR (and says so), not half of one
  free_dev(c->d_hdr); free_dev(c->d_hdr_rgb);
  DevEquirect& e = c->hdr;
  e = DevEquirect{};
  priv_of(c)->prefilter_tables_for = -1;
  const int levels = level_count(w, h);   // renderer.ts:1513
  uint32_t off[PBR_MAX_LEVELS];
  size_t total;
  chain_offsets(w, h, levels, off, &total);
  HIP_TRY(c, hipMalloc(&c->d_hdr, total * sizeof(float4)));
  // the prefilter's gathers fetch three floats per source pixel instead of four from this copy
  if (hipError_t err = hipMalloc(&c->d_hdr_rgb, total * 3 * sizeof(float)); err != hipSuccess) {
    free_dev(c->d_hdr);
    return fail(c, PBR_E_DEVICE, "hipMalloc of the HDR's alpha-less copy failed: %s", hipGetErrorString(err));
  }
  e.w = w; e.h = h; e.levels = levels;
  std::memcpy(e.off, off, sizeof off);
  e.base = c->d_hdr; e.rgb = c->d_hdr_rgb;
  HIP_TRY(c, hipMemcpyAsync(c->d_hdr, rgba32f, (size_t)w * h * sizeof(float4), hipMemcpyHostToDevice, c->stream));
  launch_build_mips_rgba32f(c->stream, c->d_hdr, e.off, w, h, e.levels);
  launch_drop_alpha(c->stream, c->d_hdr, c->d_hdr_rgb, total);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  priv_of(c)->prefilter_tables_for = -1;
  return PBR_OK;
}

int pbr_ibl_equirect_to_cube(PbrContext* c, int32_t size) {
  if (c && c->parent) return fail(c, PBR_E_STATE, "a forked context draws the scene of the context it was forked from and cannot change or read it back");
  if (!c) return PBR_E_ARG;
  if (size <= 0 || size > 8192) return fail(c, PBR_E_ARG, "bad cube size %d", size);
  if (!c->d_hdr) return fail(c, PBR_E_STATE, "no HDR uploaded");
  DeviceGuard guard(c->device);
  if (c->sky.size != size || !c->d_sky) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_dev(c->d_sky);
    size_t total;
    cube_offsets(size, 1, c->sky.off, &total);
    HIP_TRY(c, hipMalloc(&c->d_sky, total * sizeof(float4)));
    c->sky.base = c->d_sky; c->sky.size = size; c->sky.levels = 1;
  }
  launch_equirect_to_cube(c->stream, c->hdr, c->d_sky, size);
  priv_of(c)->sky_b.dirty = true;
  HIP_TRY(c, hipGetLastError());
  return PBR_OK;
}

int pbr_ibl_irradiance(PbrContext* c, int32_t size) {
  if (c && c->parent) return fail(c, PBR_E_STATE, "a forked context draws the scene of the context it was forked from and cannot change or read it back");
  if (!c) return PBR_E_ARG;
  if (size <= 0 || size > 2048) return fail(c, PBR_E_ARG, "bad cube size %d", size);
  if (!c->d_hdr) return fail(c, PBR_E_STATE, "no HDR uploaded");
  DeviceGuard guard(c->device);
  if (c->irr.size != size || !c->d_irr) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_dev(c->d_irr);
    size_t total;
    cube_offsets(size, 1, c->irr.off, &total);
    HIP_TRY(c, hipMalloc(&c->d_irr, total * sizeof(float4)));
    c->irr.base = c->d_irr; c->irr.size = size; c->irr.levels = 1;
  }
  // The angles the reference's f32 loop counters take (irradiancemap.wgsl:73-76): 252 x 63 steps.
  const float PI = 3.14159265359f, delta = 0.025f;
  std::vector<float> cp, sp, st, ct;
  for (float phi = 0.0f; phi < 2.0f * PI; phi += delta) { cp.push_back(std::cos(phi)); sp.push_back(std::sin(phi)); }
  for (float th = 0.0f; th < 0.5f * PI; th += delta) { st.push_back(std::sin(th)); ct.push_back(std::cos(th)); }
  const int n_phi = (int)cp.size(), n_theta = (int)st.size();
  if (n_phi > 512 || n_theta > 128) return fail(c, PBR_E_STATE, "irradiance tables too large");
  std::vector<float> tables;
  tables.insert(tables.end(), cp.begin(), cp.end()); tables.insert(tables.end(), sp.begin(), sp.end());
  tables.insert(tables.end(), st.begin(), st.end()); tables.insert(tables.end(), ct.begin(), ct.end());
  HIP_TRY(c, hipMemcpy(c->d_irr_tables, tables.data(), tables.size() * sizeof(float), hipMemcpyHostToDevice));
  launch_irradiance(c->stream, c->hdr, c->d_irr_tables, n_phi, n_theta, c->d_irr, size);
  priv_of(c)->irr_b.dirty = true;
  HIP_TRY(c, hipGetLastError());
  return PBR_OK;
}

int pbr_ibl_prefilter_shard_range(PbrContext* c, int32_t shard_index, int32_t shard_count, size_t* begin, size_t* end) {
  if (c && c->parent) return fail(c, PBR_E_STATE, "a forked context draws the scene of the context it was forked from and cannot change or read it back");
  if (!c || !begin || !end) return PBR_E_ARG;
  if (shard_count < 1 || shard_index < 0 || shard_index >= shard_count) return fail(c, PBR_E_ARG, "bad shard %d of %d", shard_index, shard_count);
  if (!c->d_pre) return fail(c, PBR_E_STATE, "prefilter map not allocated; call pbr_ibl_prefilter first");
  if (shard_count > PBR_MAX_SHARDS) return fail(c, PBR_E_ARG, "at most %d shards", PBR_MAX_SHARDS);
  std::vector<uint64_t> cuts((size_t)shard_count + 1);
  prefilter_cuts(c->pre.size, c->pre.levels, shard_count, cuts.data());
  *begin = (size_t)cuts[shard_index] * sizeof(float4);
  *end = (size_t)cuts[shard_index + 1] * sizeof(float4);
  return PBR_OK;
}

int pbr_ibl_prefilter_partition(int32_t size, int32_t mips, int32_t shard_count, uint64_t* cuts) {
  if (!cuts) return fail(nullptr, PBR_E_ARG, "cuts is null");
  if (size <= 0 || size > 8192) return fail(nullptr, PBR_E_ARG, "bad cube size %d", size);
  if (mips < 1 || mips > PBR_MAX_LEVELS || (size >> (mips - 1)) < 1) return fail(nullptr, PBR_E_ARG, "bad mip count %d for size %d", mips, size);
  if (shard_count < 1 || shard_count > PBR_MAX_SHARDS) return fail(nullptr, PBR_E_ARG, "bad shard count %d (1..%d)", shard_count, PBR_MAX_SHARDS);
  prefilter_cuts(size, mips, shard_count, cuts);
  return PBR_OK;
}

int pbr_ibl_prefilter_kept_samples(int32_t mips, int32_t level, int32_t* kept) {
  if (!kept) return fail(nullptr, PBR_E_ARG, "kept is null");
  if (mips < 1 || mips > PBR_MAX_LEVELS || level < 0 || level >= mips) return fail(nullptr, PBR_E_ARG, "bad level %d of %d", level, mips);
  PrefilterCost pc;
  prefilter_costs(mips, &pc);
  *kept = pc.kept[level];
  return PBR_OK;
}

// Filters texels [begin, end) of the flattened (mip, face, y, x) order; end == 0 stands for the whole map.
static int prefilter_texel_range(PbrContext* c, int32_t size, int32_t mips, int32_t samples, uint64_t begin, uint64_t end, int shard_index, int shard_count) {
  if (c && c->parent) return fail(c, PBR_E_STATE, "a forked context draws the scene of the context it was forked from and cannot change or read it back");
  if (!c) return PBR_E_ARG;
  if (size <= 0 || size > 8192) return fail(c, PBR_E_ARG, "bad cube size %d", size);
  if (mips < 1 || mips > PBR_MAX_LEVELS || (size >> (mips - 1)) < 1) return fail(c, PBR_E_ARG, "bad mip count %d for size %d", mips, size);
  if (samples != PBR_PREFILTER_SAMPLES) return fail(c, PBR_E_UNSUPPORTED, "sample count %d: the reference fixes SAMPLE_COUNT at 1024", samples);
  if (shard_count >= 0 && (shard_count < 1 || shard_count > PBR_MAX_SHARDS || shard_index < 0 || shard_index >= shard_count))
    return fail(c, PBR_E_ARG, "bad shard %d of %d", shard_index, shard_count);
  if (!c->d_hdr) return fail(c, PBR_E_STATE, "no HDR uploaded");
  DeviceGuard guard(c->device);
  PbrContextPrivate* pr = priv_of(c);
  if (c->pre.size != size || c->pre.levels != mips || !c->d_pre) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_dev(c->d_pre);
    size_t total;
    cube_offsets(size, mips, c->pre.off, &total);
    HIP_TRY(c, hipMalloc(&c->d_pre, total * sizeof(float4)));
    HIP_TRY(c, hipMemset(c->d_pre, 0, total * sizeof(float4)));
    c->pre.base = c->d_pre; c->pre.size = size; c->pre.levels = mips;
    c->pre_total_texels = total;
    pr->prefilter_tables_for = -1;
  }
  if (pr->prefilter_tables_for != mips || pr->prefilter_src_levels != c->hdr.levels) {
    std::vector<PrefilterSample> host((size_t)PBR_PREFILTER_SAMPLES * mips);
    for (int m = 0; m < mips; ++m) {
      const float roughness = prefilter_roughness(m, mips);
      pr->table_count[m] = build_prefilter_table(roughness, c->hdr.levels, host.data() + (size_t)m * PBR_PREFILTER_SAMPLES, &pr->table_weight[m], &pr->table_single[m]);
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->d_samples, host.data(), host.size() * sizeof(PrefilterSample), hipMemcpyHostToDevice));
    pr->prefilter_tables_for = mips; pr->prefilter_src_levels = c->hdr.levels;
  }
  uint64_t replicated = 0;
  if (shard_count >= 0) {
    std::vector<uint64_t> cuts((size_t)shard_count + 1);
    prefilter_cuts(size, mips, shard_count, cuts.data());
    begin = cuts[shard_index]; end = cuts[shard_index + 1];
    replicated = cuts[0];
  } else if (begin > end || end > c->pre_total_texels) {
    return fail(c, PBR_E_ARG, "texel range [%llu, %llu) outside the map's %zu texels", (unsigned long long)begin, (unsigned long long)end, c->pre_total_texels);
  }
  PrefilterLevelRun runs[PBR_MAX_LEVELS];
  int n_runs = 0;
  for (int m = 0; m < mips; ++m) {
    const size_t s = (size_t)std::max(1, size >> m);
    const size_t lo = c->pre.off[m], hi = lo + 6 * s * s;
    // the levels in front of the shards' ranges are filtered whole by every shard (prefilter_cuts)
    const size_t a = hi <= (size_t)replicated ? lo : std::max(lo, (size_t)begin), b = hi <= (size_t)replicated ? hi : std::min(hi, (size_t)end);
    if (a >= b) continue;
    runs[n_runs++] = PrefilterLevelRun{c->d_samples + (size_t)m * PBR_PREFILTER_SAMPLES, pr->table_count[m], pr->table_single[m], pr->table_weight[m],
                                       c->d_pre + lo, (int)s, (int64_t)(a - lo), (int64_t)(b - a)};
  }
  launch_prefilter_levels(c->stream, c->hdr, runs, n_runs);
  pr->pre_b.dirty = true;
  HIP_TRY(c, hipGetLastError());
  return PBR_OK;
}

int pbr_ibl_prefilter(PbrContext* c, int32_t size, int32_t mips, int32_t samples, int32_t shard_index, int32_t shard_count) {
  if (shard_count < 0) return c ? fail(c, PBR_E_ARG, "bad shard %d of %d", shard_index, shard_count) : PBR_E_ARG;
  return prefilter_texel_range(c, size, mips, samples, 0, 0, shard_index, shard_count);
}

int pbr_ibl_prefilter_texels(PbrContext* c, int32_t size, int32_t mips, int32_t samples, uint64_t first_texel, uint64_t texel_count) {
  return prefilter_texel_range(c, size, mips, samples, first_texel, first_texel + texel_count, 0, -1);
}

int pbr_ibl_brdf_lut(PbrContext* c, int32_t size) {
  if (c && c->parent) return fail(c, PBR_E_STATE, "a forked context draws the scene of the context it was forked from and cannot change or read it back");
  if (!c) return PBR_E_ARG;
  if (size < 2 || size > 8192) return fail(c, PBR_E_ARG, "bad LUT size %d", size);
  DeviceGuard guard(c->device);
  if (c->lut_size != size || !c->d_lut) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_dev(c->d_lut);
    HIP_TRY(c, hipMalloc(&c->d_lut, (size_t)size * size * 4 * sizeof(uint16_t)));
    c->lut_size = size;
  }
  launch_brdf_lut(c->stream, c->d_lut, size);
  priv_of(c)->lut_quad_dirty = true;
  HIP_TRY(c, hipGetLastError());
  return PBR_OK;
}

static int map_level(PbrContext* c, int which, int level, const void** ptr, int* w, int* h, int* layers, size_t* bytes) {
  if (c && c->parent) return fail(c, PBR_E_STATE, "a forked context draws the scene of the context it was forked from and cannot change or read it back");
  switch (which) {
    case PBR_MAP_SKYBOX:
    case PBR_MAP_IRRADIANCE:
    case PBR_MAP_PREFILTER: {
      const DevCube& m = which == PBR_MAP_SKYBOX ? c->sky : which == PBR_MAP_IRRADIANCE ? c->irr : c->pre;
      if (!m.base) return fail(c, PBR_E_STATE, "map %d not generated", which);
      if (level < 0 || level >= m.levels) return fail(c, PBR_E_ARG, "level %d out of range", level);
      const int s = std::max(1, m.size >> level);
      *ptr = m.base + m.off[level]; *w = s; *h = s; *layers = 6; *bytes = (size_t)6 * s * s * sizeof(float4);
      return PBR_OK;
    }
    case PBR_MAP_BRDF_LUT:
      if (!c->d_lut) return fail(c, PBR_E_STATE, "BRDF LUT not generated");
      if (level != 0) return fail(c, PBR_E_ARG, "level %d out of range", level);
      *ptr = c->d_lut; *w = c->lut_size; *h = c->lut_size; *layers = 1; *bytes = (size_t)c->lut_size * c->lut_size * 4 * sizeof(uint16_t);
      return PBR_OK;
    case PBR_MAP_HDR:
      if (!c->d_hdr) return fail(c, PBR_E_STATE, "no HDR uploaded");
      if (level < 0 || level >= c->hdr.levels) return fail(c, PBR_E_ARG, "level %d out of range", level);
      *w = std::max(1, c->hdr.w >> level); *h = std::max(1, c->hdr.h >> level); *layers = 1;
      *ptr = c->hdr.base + c->hdr.off[level]; *bytes = (size_t)*w * *h * sizeof(float4);
      return PBR_OK;
    default:
      return fail(c, PBR_E_ARG, "unknown map %d", which);
  }
}

int pbr_ibl_level_info(PbrContext* c, int which, int32_t level, int32_t* w, int32_t* h, int32_t* layers, size_t* bytes) {
  if (!c) return PBR_E_ARG;
  const void* p; int ww, hh, ll; size_t bb;
  int rc = map_level(c, which, level, &p, &ww, &hh, &ll, &bb);
  if (rc) return rc;
  if (w) *w = ww;
  if (h) *h = hh;
  if (layers) *layers = ll;
  if (bytes) *bytes = bb;
  return PBR_OK;
}

int pbr_ibl_download(PbrContext* c, int which, int32_t level, void* out, size_t out_bytes) {
  if (!c || !out) return PBR_E_ARG;
  const void* p; int w, h, l; size_t bytes;
  int rc = map_level(c, which, level, &p, &w, &h, &l, &bytes);
  if (rc) return rc;
  if (out_bytes < bytes) return fail(c, PBR_E_ARG, "output buffer holds %zu bytes, level needs %zu", out_bytes, bytes);
  DeviceGuard guard(c->device);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(out, p, bytes, hipMemcpyDeviceToHost));
  return PBR_OK;
}

int pbr_ibl_device_ptr(PbrContext* c, int which, void** dev_ptr, size_t* total_bytes) {
  if (c && c->parent) return fail(c, PBR_E_STATE, "a forked context draws the scene of the context it was forked from and cannot change or read it back");
  if (!c || !dev_ptr) return PBR_E_ARG;
  size_t bytes = 0;
  void* p = nullptr;
  switch (which) {
    case PBR_MAP_SKYBOX: p = c->d_sky; bytes = (size_t)6 * c->sky.size * c->sky.size * sizeof(float4); break;
    case PBR_MAP_IRRADIANCE: p = c->d_irr; bytes = (size_t)6 * c->irr.size * c->irr.size * sizeof(float4); break;
    case PBR_MAP_PREFILTER: p = c->d_pre; bytes = c->pre_total_texels * sizeof(float4); break;
    case PBR_MAP_BRDF_LUT: p = c->d_lut; bytes = (size_t)c->lut_size * c->lut_size * 4 * sizeof(uint16_t); break;
    case PBR_MAP_HDR: { p = c->d_hdr; size_t t; uint32_t off[PBR_MAX_LEVELS]; chain_offsets(c->hdr.w, c->hdr.h, c->hdr.levels, off, &t); bytes = t * sizeof(float4); break; }
    default: return fail(c, PBR_E_ARG, "unknown map %d", which);
  }
  if (!p) return fail(c, PBR_E_STATE, "map %d not generated", which);
  // the caller may write through this pointer (e.g. an all-gather of shards): derived copies are rebuilt at the next frame
  PbrContextPrivate* pr = priv_of(c);
  if (which == PBR_MAP_SKYBOX) pr->sky_b.dirty = true;
  if (which == PBR_MAP_IRRADIANCE) pr->irr_b.dirty = true;
  if (which == PBR_MAP_BRDF_LUT) pr->lut_quad_dirty = true;
  if (which == PBR_MAP_PREFILTER) pr->pre_b.dirty = true;
  *dev_ptr = p;
  if (total_bytes) *total_bytes = bytes;
  return PBR_OK;
}

int pbr_draw_frame_device(PbrContext* c, const PbrFrameUniforms* u, int32_t w, int32_t h, int32_t y0, int32_t y1, void* d_rgba8, void* d_vis,
                          PbrFrameStats* stats) {
  if (!c) return PBR_E_ARG;
  if (!d_rgba8) return fail(c, PBR_E_ARG, "colour output pointer is null");
  // with statistics the call synchronises and sees a capacity overflow itself: the buffers have then been enlarged and
  // the frame is drawn again; without, the frame is only queued and an overflow is reported by pbr_synchronize
  int rc = PBR_OK;
  for (int attempt = 0; attempt < 4; ++attempt) {
    c->enlarged_after_overflow = false;
    rc = draw_common(c, u, w, h, y0, y1, d_rgba8, d_vis, false, stats);
    if (!(rc == PBR_E_STATE && c->enlarged_after_overflow)) break;
  }
  return rc;
}

// A hidden context for one more view of a multi-view pass: working buffers and counters of its own, the scene of the
// caller (not counted among the owner's forks: it is destroyed with the context that made it).
static int make_view_slot(PbrContext* c, PbrContext** out) {
  PbrContext* slot = nullptr;
  const int rc = pbr_create(c->device, &slot);
  if (rc) return fail(c, rc, "%s", pbr_last_error(nullptr));
  slot->parent = assets_of(c);
  slot->view_slot = true;
  slot->forced_setup_cap = c->forced_setup_cap; slot->forced_overflow_cap = c->forced_overflow_cap;   // (test hook: pbr_debug_set_capacity)
  *out = slot;
  return PBR_OK;
}

int pbr_draw_views_device(PbrContext* c, int32_t n_views, const PbrFrameUniforms* u, int32_t W, int32_t H, int32_t y0, int32_t y1, void* const* d_rgba8) {
  if (!c) return PBR_E_ARG;
  if (n_views < 1 || n_views > PBR_MAX_VIEWS) return fail(c, PBR_E_ARG, "%d views: a pass draws 1 to %d", n_views, PBR_MAX_VIEWS);
  if (!u || !d_rgba8) return fail(c, PBR_E_ARG, "uniforms or colour output array is null");
  for (int v = 0; v < n_views; ++v) {
    if (!d_rgba8[v]) return fail(c, PBR_E_ARG, "colour output pointer of view %d is null", v);
    for (int w = 0; w < v; ++w) if (d_rgba8[w] == d_rgba8[v]) return fail(c, PBR_E_ARG, "views %d and %d share one colour target", w, v);
  }
  PbrContext* a = assets_of(c);
  PbrContextPrivate* apr = priv_of(a);
  PbrContextPrivate* pr = priv_of(c);
  // the one-launch pass exists for the specialised pipeline (the reference's default sampler on five power-of-two
  // textures of one size, single-sampled); anything else is drawn view after view, with the same result
  static const int kFormatSrgb[5] = {1, 0, 0, 1, 0};   // renderer.ts:647, :663, :681, :697, :713
  bool one_size = a->d_tex[0] != nullptr;
  for (int i = 0; i < 5 && one_size; ++i)
    one_size = a->d_tex[i] && a->tex[i].base && a->tex[i].srgb == kFormatSrgb[i] && (a->tex[i].w & (a->tex[i].w - 1)) == 0 && (a->tex[i].h & (a->tex[i].h - 1)) == 0 &&
               a->tex[i].w == a->tex[0].w && a->tex[i].h == a->tex[0].h && a->tex[i].levels == a->tex[0].levels;
  bool fast = one_size && n_views > 1;
  for (int v = 0; v < n_views; ++v) fast = fast && u[v].sampler_index == 15u;
  if (!fast) {
    for (int v = 0; v < n_views; ++v) {
      const int rc = pbr_draw_frame_device(c, &u[v], W, H, y0, y1, d_rgba8[v], nullptr, nullptr);
      if (rc) return rc;
    }
    return PBR_OK;
  }
  if (W <= 0 || H <= 0 || W > 16384 || H > 16384) return fail(c, PBR_E_ARG, "bad target size %dx%d", W, H);
  if (y0 < 0 || y1 > H || y0 >= y1 || (y0 & 1) || ((y1 & 1) && y1 != H)) return fail(c, PBR_E_ARG, "bad row range [%d,%d) for height %d (rows must start and end on even lines)", y0, y1, H);
  if (!a->d_vsoa) return fail(c, PBR_E_STATE, "no mesh uploaded");
  if (!a->d_irr) return fail(c, PBR_E_STATE, "irradiance map not generated");
  if (!a->d_pre) return fail(c, PBR_E_STATE, "prefilter map not generated");
  if (!a->d_lut) return fail(c, PBR_E_STATE, "BRDF LUT not generated");
  if (!a->d_sky) return fail(c, PBR_E_STATE, "skybox cube not generated");
  if (a->has_instance_mats == false && a->n_instances != 1) return fail(c, PBR_E_STATE, "instanced mesh without instance matrices");
  DeviceGuard guard(c->device);
  int rc = PBR_OK;
  const bool stale = apr->sky_b.dirty || !apr->sky_b.d || apr->irr_b.dirty || !apr->irr_b.d || apr->pre_b.dirty || !apr->pre_b.d || apr->lut_quad_dirty ||
                     apr->lut_quad_size != a->lut_size || apr->atlas_dirty || !apr->d_atlas;
  if (stale && (rc = refresh_lookup_copies(c, a, apr, true))) return rc;
  if ((rc = wait_for_lookup_copies(c, apr))) return rc;

  const int rows = y1 - y0;
  const int tiles_x = (W + PBR_TILE - 1) / PBR_TILE, tiles_y = (rows + PBR_TILE - 1) / PBR_TILE, n_tiles = tiles_x * tiles_y;
  hipStream_t s = c->stream;
  while ((int)pr->view_slots.size() < n_views - 1) {
    PbrContext* slot = nullptr;
    if ((rc = make_view_slot(c, &slot))) return rc;
    pr->view_slots.push_back(slot);
  }

  SkyShare share;
  share.sky = apr->sky_b.cube; share.rgba8 = nullptr; share.first = 0; share.own_blocks = 0;
  share.count = std::min(n_tiles, std::max(0, (int)(n_tiles * PBR_SKY_SHARE_SETUP_PASS)));
  ViewArgs views[PBR_MAX_VIEWS];
  std::memset(views, 0, sizeof views);
  // everything that can fail comes before anything a failure would leave half done: the counter sets below are flipped
  // on the promise that this pass's rasteriser re-arms the other set
  for (int v = 0; v < n_views; ++v) {
    PbrContext* w = v == 0 ? c : pr->view_slots[v - 1];   // whose working buffers and counters this view uses
    w->stream = s;                                        // (a slot's buffers are used, and may only be regrown, in this stream's order)
    if ((rc = ensure_frame_buffers(w, W, rows, n_tiles, 1, /*outputs=*/false))) return w == c ? rc : fail(c, rc, "%s", w->error.c_str());
  }
  for (int v = 0; v < n_views; ++v) {
    PbrContext* w = v == 0 ? c : pr->view_slots[v - 1];
    PbrContextPrivate* wpr = priv_of(w);
    ViewArgs& va = views[v];
    FrameParams& fp = va.fp;
    fp.u = u[v];
    mat4_mul_host(u[v].projection, u[v].view, fp.pv);
    fp.W = W; fp.H = H; fp.y0 = y0; fp.y1 = y1; fp.tiles_x = tiles_x; fp.tiles_y = tiles_y;
    fp.tiles_x_magic = (uint32_t)((0x100000000ull + (uint64_t)tiles_x - 1) / (uint64_t)tiles_x);
    fp.n_verts = a->n_verts; fp.n_tris = a->n_indices / 3; fp.n_instances = a->n_instances;
    fp.has_instance_mats = a->has_instance_mats ? 1 : 0;
    fp.msaa = 0;
    fp.two_over_w = 2.0f / (float)W; fp.two_over_h = 2.0f / (float)H;
    if (!sky_affine(u[v].inv_view_proj, W, H, fp)) return fail(c, PBR_E_ARG, "the inverse view-projection matrix of view %d (skybox uniform, renderer.ts:1864-1867) is not finite", v);
    uint32_t* counters = w->d_counters + 8 * wpr->counter_set;
    uint32_t* next_counters = w->d_counters + 8 * (1 - wpr->counter_set);
    wpr->counter_set = 1 - wpr->counter_set;
    const LiveTiles live = {w->d_live_tiles, w->d_live_info, PBR_LIVE_TILE_CAP, ++w->live_serial};
    BinTargets& bt = va.bt;
    bt.setup = w->d_setup; bt.setup_cap = (uint32_t)w->setup_cap; bt.tile_count = w->d_tile_count; bt.bins = w->d_bins; bt.overflow = w->d_overflow;
    bt.overflow_cap = (uint32_t)w->overflow_cap; bt.counters = counters; bt.live = live; bt.records = w->d_records;
    bt.vsoa = a->d_vsoa; bt.inst_model = a->d_inst_model; bt.inst_normal = a->d_inst_normal;
    for (int k = 0; k < 3; ++k) { bt.mesh_lo[k] = a->mesh_lo[k]; bt.mesh_hi[k] = a->mesh_hi[k]; }
    RasterTargets& ra = va.ra;
    ra.setup = w->d_setup; ra.tile_count = w->d_tile_count; ra.bins = w->d_bins; ra.overflow = w->d_overflow; ra.overflow_cap = (uint32_t)w->overflow_cap;
    ra.counters = counters; ra.next_counters = next_counters; ra.lists = w->quad_lists; ra.rgba8 = static_cast<uint32_t*>(d_rgba8[v]); ra.live = live;
    va.records = w->d_records;
    w->last_w = W; w->last_rows = rows; w->last_samples = 1;
    ++wpr->frames_drawn;
  }
  const bool prof = c->profiling && c->prof_frames < c->prof_cap;
  const int prof_stride = c->prof_all ? 4 : 2;
  hipEvent_t* pe = prof ? &c->prof_events[prof_stride * c->prof_frames] + (prof_stride - 2) : nullptr;   // the pass's launches count as one profiled frame
  if (prof && c->prof_all) HIP_TRY(c, hipEventRecord(pe[-2], s));
  launch_setup_views(s, views, n_views, a->d_indices, a->n_indices / 3, a->n_instances, share);
  if (prof && c->prof_all) HIP_TRY(c, hipEventRecord(pe[-1], s));
  launch_raster_views(s, views, n_views, apr->sky_b.cube, share.first + share.count, tiles_x, tiles_y, PBR_LIVE_TILE_CAP);
  ResolveInputs in;
  in.records = nullptr; in.vout = a->d_vout; in.indices = a->d_indices;
  for (int i = 0; i < 5; ++i) in.tex[i] = a->tex[i];
  in.irr = a->irr; in.pre = a->pre; in.sky = a->sky;
  in.irr_b = apr->irr_b.cube; in.pre_b = apr->pre_b.cube; in.sky_b = apr->sky_b.cube;
  in.lut = a->d_lut; in.lut_quad = apr->d_lut_quad; in.atlas = apr->d_atlas; in.lut_size = a->lut_size;
  if (prof) HIP_TRY(c, hipEventRecord(pe[0], s));
  const bool alone = c->parent == nullptr && c->children == 0;
  launch_surface_views(s, views, n_views, in, pr->d_srgb_lut, pr->surface_blocks, /*shared_chip=*/!alone);
  if (prof) { HIP_TRY(c, hipEventRecord(pe[1], s)); ++c->prof_frames; }
  HIP_TRY(c, hipGetLastError());
  return PBR_OK;
}

int pbr_release_view_buffers(PbrContext* c) {
  if (!c) return PBR_E_ARG;
  DeviceGuard guard(c->device);
  HIP_TRY(c, hipStreamSynchronize(c->stream));   // the slots' buffers are used in this stream's order
  PbrContextPrivate* pr = priv_of(c);
  int rc = PBR_OK;
  for (PbrContext* slot : pr->view_slots) {
    HIP_TRY(c, hipMemcpy(slot->h_counters, slot->d_counters, kCounterHeadBytes, hipMemcpyDeviceToHost));
    if (check_sticky(slot) && rc == PBR_OK) rc = fail(c, PBR_E_STATE, "%s", slot->error.c_str());   // an overflow of a queued pass is still reported
    slot->stream = slot->own_stream;
    (void)pbr_destroy(slot);
  }
  pr->view_slots.clear();
  return rc;
}

int pbr_draw_frame_static(PbrContext* c, const PbrFrameUniforms* u, int32_t w, int32_t h, int32_t y0, int32_t y1, void* d_rgba8) {
  if (!c) return PBR_E_ARG;
  if (!d_rgba8) return fail(c, PBR_E_ARG, "colour output pointer is null");
  return draw_common(c, u, w, h, y0, y1, d_rgba8, nullptr, false, nullptr, true);
}

int pbr_draw_frame(PbrContext* c, const PbrFrameUniforms* u, int32_t w, int32_t h, int32_t y0, int32_t y1, uint8_t* rgba8_out, uint32_t* vis_out,
                   PbrFrameStats* stats) {
  if (!c) return PBR_E_ARG;
  if (!rgba8_out) return fail(c, PBR_E_ARG, "colour output pointer is null");
  if (w <= 0 || h <= 0 || y0 < 0 || y1 > h || y0 >= y1) return fail(c, PBR_E_ARG, "bad target %dx%d rows [%d,%d)", w, h, y0, y1);
  DeviceGuard guard(c->device);
  // make sure the internal colour / visibility buffers exist at this size before drawing into them
  const int rows = y1 - y0;
  const int samples = (u && (u->sampler_index & 16u)) ? 4 : 1;
  const size_t npix = (size_t)w * rows;
  int rc = PBR_OK;
  for (int attempt = 0; attempt < 4; ++attempt) {   // a frame that overflowed a working buffer is drawn again into the enlarged one
    c->enlarged_after_overflow = false;
    rc = ensure_frame_buffers(c, w, rows, ((w + PBR_TILE - 1) / PBR_TILE) * ((rows + PBR_TILE - 1) / PBR_TILE), samples);
    if (rc) return rc;
    rc = draw_common(c, u, w, h, y0, y1, c->d_color, nullptr, vis_out != nullptr, stats);
    if (rc == PBR_OK) {
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      HIP_TRY(c, hipMemcpy(c->h_counters, c->d_counters, kCounterHeadBytes, hipMemcpyDeviceToHost));
      rc = check_sticky(c);
    }
    if (!(rc == PBR_E_STATE && c->enlarged_after_overflow)) break;
  }
  if (rc) return rc;
  HIP_TRY(c, hipMemcpy(rgba8_out, c->d_color, npix * 4, hipMemcpyDeviceToHost));
  if (vis_out) HIP_TRY(c, hipMemcpy(vis_out, c->d_vis, npix * samples * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return PBR_OK;
}

int pbr_debug_set_capacity(PbrContext* c, size_t setup_triangles, size_t overflow_refs) {
  if (!c) return PBR_E_ARG;
  c->forced_setup_cap = setup_triangles; c->forced_overflow_cap = overflow_refs;
  c->min_setup_cap = 0; c->min_overflow_cap = 0;
  for (PbrContext* slot : priv_of(c)->view_slots) {   // the views of a pass each work in buffers of their own
    slot->forced_setup_cap = setup_triangles; slot->forced_overflow_cap = overflow_refs;
    slot->min_setup_cap = 0; slot->min_overflow_cap = 0;
  }
  return PBR_OK;
}

int pbr_debug_vertex_stage(PbrContext* c, const PbrFrameUniforms* u, float* clip, float* world, float* normal) {
  if (c && c->parent) return fail(c, PBR_E_STATE, "a forked context draws the scene of the context it was forked from and cannot change or read it back");
  if (!c || !u) return PBR_E_ARG;
  if (!c->d_vsoa) return fail(c, PBR_E_STATE, "no mesh uploaded");
  DeviceGuard guard(c->device);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  {   // transformed vertices exist in memory only for this call: frames transform inside the setup kernel
    const size_t want = (size_t)c->n_verts * c->n_instances;
    if (want > c->vout_cap) {
      free_dev(c->d_vout);
      HIP_TRY(c, hipMalloc(&c->d_vout, want * sizeof(VertexOut)));
      c->vout_cap = want;
    }
  }
  FrameParams fp;
  fp.u = *u;
  mat4_mul_host(u->projection, u->view, fp.pv);
  fp.W = 32; fp.H = 32; fp.y0 = 0; fp.y1 = 32; fp.tiles_x = 1; fp.tiles_y = 1; fp.tiles_x_magic = 0u;   // (no skybox tiles ride in this launch)
  fp.n_verts = c->n_verts; fp.n_tris = c->n_indices / 3; fp.n_instances = c->n_instances; fp.has_instance_mats = c->has_instance_mats ? 1 : 0;
  fp.msaa = 0;
  fp.two_over_w = 2.0f / 32.0f; fp.two_over_h = 2.0f / 32.0f;
  sky_affine(u->inv_view_proj, 32, 32, fp);
  SkyShare none = {};
  launch_vertex_stage(c->stream, fp, c->d_vsoa, c->d_inst_model, c->d_inst_normal, c->d_vout, c->n_verts, c->n_instances, none);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const size_t nv = (size_t)c->n_verts * c->n_instances;
  std::vector<VertexOut> host(nv);
  HIP_TRY(c, hipMemcpy(host.data(), c->d_vout, nv * sizeof(VertexOut), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < nv; ++i) {
    if (clip) { clip[4 * i] = host[i].clip.x; clip[4 * i + 1] = host[i].clip.y; clip[4 * i + 2] = host[i].clip.z; clip[4 * i + 3] = host[i].clip.w; }
    if (world) { world[3 * i] = host[i].world_u.x; world[3 * i + 1] = host[i].world_u.y; world[3 * i + 2] = host[i].world_u.z; }
    if (normal) { normal[3 * i] = host[i].nrm_v.x; normal[3 * i + 1] = host[i].nrm_v.y; normal[3 * i + 2] = host[i].nrm_v.z; }
  }
  return PBR_OK;
}

int pbr_debug_texture_level(PbrContext* c, int slot, int32_t level, uint8_t* rgba8_out, size_t out_bytes, int32_t* w_out, int32_t* h_out) {
  if (!c) return PBR_E_ARG;
  if (c->parent) return fail(c, PBR_E_STATE, "a forked context draws the scene of the context it was forked from and cannot change or read it back");
  if (slot < 0 || slot > 4) return fail(c, PBR_E_ARG, "texture slot %d out of range", slot);
  if (!c->d_tex[slot]) return fail(c, PBR_E_STATE, "material texture slot %d not uploaded", slot);
  const DevTex8& t = c->tex[slot];
  if (level < 0 || level >= t.levels) return fail(c, PBR_E_ARG, "level %d out of range (0..%d)", level, t.levels - 1);
  const int w = std::max(1, t.w >> level), h = std::max(1, t.h >> level);
  if (w_out) *w_out = w;
  if (h_out) *h_out = h;
  if (!rgba8_out) return PBR_OK;   // size query
  if (out_bytes < (size_t)w * h * 4) return fail(c, PBR_E_ARG, "output buffer holds %zu bytes, level needs %zu", out_bytes, (size_t)w * h * 4);
  DeviceGuard guard(c->device);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  // the level is stored in blocks of 8 x 4 texels on the device; undo that on the host
  std::vector<uint32_t> blocked(tex8_level_texels(w, h));
  HIP_TRY(c, hipMemcpy(blocked.data(), t.base + t.off[level], blocked.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  uint32_t* out = reinterpret_cast<uint32_t*>(rgba8_out);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) { const uint32_t v = blocked[tex8_block_offset(x, y, w)]; std::memcpy(out + (size_t)y * w + x, &v, 4); }
  return PBR_OK;
}

int pbr_debug_capture_depth(PbrContext* c, int32_t enable) {
  if (!c) return PBR_E_ARG;
  c->capture_depth = enable != 0;
  return PBR_OK;
}

int pbr_debug_depth(PbrContext* c, float* depth_out, size_t n_floats) {
  if (!c || !depth_out) return PBR_E_ARG;
  if (!c->capture_depth) return fail(c, PBR_E_STATE, "depth capture is off; call pbr_debug_capture_depth(ctx, 1) before drawing");
  const size_t n = (size_t)c->last_w * c->last_rows * c->last_samples;
  if (!c->d_depth || n == 0) return fail(c, PBR_E_STATE, "no frame drawn yet");
  if (n_floats < n) return fail(c, PBR_E_ARG, "depth buffer needs %zu floats", n);
  DeviceGuard guard(c->device);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(depth_out, c->d_depth, n * sizeof(float), hipMemcpyDeviceToHost));
  return PBR_OK;
}

static int profile_begin(PbrContext* c, int32_t max_frames, bool all) {
  if (!c || max_frames <= 0 || max_frames > (1 << 20)) return c ? fail(c, PBR_E_ARG, "bad frame count") : PBR_E_ARG;
  DeviceGuard guard(c->device);
  while ((int)c->prof_events.size() < (all ? 4 : 2) * max_frames) {
    hipEvent_t ev;
    HIP_TRY(c, hipEventCreate(&ev));
    c->prof_events.push_back(ev);
  }
  c->prof_cap = max_frames; c->prof_frames = 0; c->profiling = true; c->prof_all = all;
  return PBR_OK;
}

int pbr_profile_begin(PbrContext* c, int32_t max_frames) { return profile_begin(c, max_frames, false); }
int pbr_profile_begin_kernels(PbrContext* c, int32_t max_frames) { return profile_begin(c, max_frames, true); }

int pbr_profile_end_kernels(PbrContext* c, int32_t* frames, float* kernel_ms_total, float* span_ms) {
  if (!c) return PBR_E_ARG;
  DeviceGuard guard(c->device);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const int stride = c->prof_all ? 4 : 2;
  float total[3] = {0, 0, 0}, span = 0;
  for (int i = 0; i < c->prof_frames; ++i) {
    hipEvent_t* e = &c->prof_events[stride * i];
    // four events: set-up begin, raster begin, surface begin, end; two events: surface begin, end
    for (int k = c->prof_all ? 0 : 2; k < 3; ++k) {
      const int first = c->prof_all ? k : 0;
      float ms = 0;
      HIP_TRY(c, hipEventElapsedTime(&ms, e[first], e[first + 1]));
      total[k] += ms;
    }
  }
  if (c->prof_frames > 1) HIP_TRY(c, hipEventElapsedTime(&span, c->prof_events[stride - 1], c->prof_events[stride * c->prof_frames - 1]));
  if (frames) *frames = c->prof_frames;
  if (kernel_ms_total) for (int k = 0; k < 3; ++k) kernel_ms_total[k] = total[k];
  if (span_ms) *span_ms = span;
  c->profiling = false; c->prof_all = false;
  return PBR_OK;
}

int pbr_profile_end(PbrContext* c, int32_t* frames, float* surface_ms_total, float* span_ms) {
  float total[3] = {0, 0, 0};
  const int rc = pbr_profile_end_kernels(c, frames, total, span_ms);
  if (rc == PBR_OK && surface_ms_total) *surface_ms_total = total[2];
  return rc;
}

}  // extern "C"
