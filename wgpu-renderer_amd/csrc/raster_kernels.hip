// gfx950 kernels for the geometry half of the frame: vertex transform, triangle setup with
// clipping and culling, tile binning, and the tile rasteriser that writes the triangle-id
// visibility buffer.  Together they stand in for vs_main (src/shaders/shader.wgsl:37-47) and the
// fixed-function rasteriser configured at src/renderer.ts:469-480 and :1943-1966.
//
// Rasterisation rules (the same ones the parity oracle encodes, so the visibility buffer can be
// compared bit for bit): clip volume 0<=z<=w with an 8w guard band in x and y; vertices snapped
// to 1/256 pixel; front face counter-clockwise, back faces culled; top-left fill rule at pixel
// centres; depth = z/w interpolated linearly in screen space in f32, test "less" against a clear
// of 1.0, equal depths resolved in draw order.  All arithmetic that feeds an integer decision uses
// explicitly rounded operations (no FMA contraction).
#include "pbr_internal.h"
#include "devmath.h"
#include "sky_shade.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Vertex stage.  Attribute streams are SoA (px[], py[], pz[], u[], v[], nx[], ny[], nz[]) so a
// wavefront's 64 loads of one attribute are one contiguous 256-byte request.
// ---------------------------------------------------------------------------------------------
// vs_main (shader.wgsl:37-47) for vertex i of one instance; every product and sum rounded on its own.
struct VertexMatrices { float model[16], nmat[16], pvm[16]; };

PBR_DEV void load_matrices(const FrameParams* __restrict__ p, const float* __restrict__ inst_model, const float* __restrict__ inst_normal, int inst,
                           VertexMatrices& m) {
  const float* msrc = p->has_instance_mats ? inst_model + 16 * inst : p->u.model;
  const float* nsrc = p->has_instance_mats ? inst_normal + 16 * inst : p->u.normal;
#pragma unroll
  for (int k = 0; k < 16; ++k) { m.model[k] = msrc[k]; m.nmat[k] = nsrc[k]; }
  mat4_mul_exact(p->pv, m.model, m.pvm);   // (projection * view) * model
}

PBR_DEV VertexOut transform_vertex(const VertexMatrices& m, const float* __restrict__ vsoa, size_t n, int i) {
  const float x = vsoa[i], y = vsoa[n + i], z = vsoa[2 * n + i];
  const float tu = vsoa[3 * n + i], tv = vsoa[4 * n + i];
  const float nx = vsoa[5 * n + i], ny = vsoa[6 * n + i], nz = vsoa[7 * n + i];
  VertexOut o;
  o.clip = mat4_vec_exact(m.pvm, x, y, z, 1.0f);
  const float4 w = mat4_vec_exact(m.model, x, y, z, 1.0f);
  const float4 nn = mat4_vec_exact(m.nmat, nx, ny, nz, 0.0f);
  o.world_u = make_float4(w.x, w.y, w.z, tu);
  o.nrm_v = make_float4(nn.x, nn.y, nn.z, tv);
  return o;
}

// The vertex stage as a launch of its own: used when the vertex outputs are wanted in memory (the
// general-sampler resolve, verification).  The specialised frame transforms vertices inside the setup kernel.
__global__ void __launch_bounds__(256) vertex_kernel(const FrameParams frame_params, const float* __restrict__ vsoa,
                                                     const float* __restrict__ inst_model, const float* __restrict__ inst_normal,
                                                     VertexOut* __restrict__ out, int n_verts, int blocks_per_instance, SkyShare share) {
  const FrameParams* __restrict__ p = &frame_params;   // parameters travel as kernel arguments: no per-frame copy to device memory
  if ((int)blockIdx.x >= share.own_blocks) { sky_tile(p, share.sky, share.rgba8, share.first + (int)blockIdx.x - share.own_blocks); return; }
  const int inst = (int)blockIdx.x / blocks_per_instance;
  const int i = ((int)blockIdx.x - inst * blocks_per_instance) * 256 + threadIdx.x;
  if (i >= n_verts) return;
  VertexMatrices m;
  load_matrices(p, inst_model, inst_normal, inst, m);
  out[(size_t)inst * n_verts + i] = transform_vertex(m, vsoa, (size_t)n_verts, i);
}

// ---------------------------------------------------------------------------------------------
// Triangle setup
// ---------------------------------------------------------------------------------------------
struct Snapped { int X, Y; float z; };

PBR_DEV Snapped snap_vertex(float4 c, int W, int H) {
  const float nx = div_rn(c.x, c.w), ny = div_rn(c.y, c.w), nz = div_rn(c.z, c.w);
  const float xf = mul_rn(mul_rn(add_rn(nx, 1.0f), 0.5f), (float)W);
  const float yf = mul_rn(mul_rn(sub_rn(1.0f, ny), 0.5f), (float)H);
  Snapped s;
  s.X = __float2int_rn(mul_rn(xf, 256.0f));
  s.Y = __float2int_rn(mul_rn(yf, 256.0f));
  s.z = nz;
  return s;
}

PBR_DEV float plane_dist(int plane, float4 c) {
  switch (plane) {
    case 0: return c.z;
    case 1: return sub_rn(c.w, c.z);
    case 2: return sub_rn(mul_rn(PBR_GUARD, c.w), c.x);
    case 3: return add_rn(mul_rn(PBR_GUARD, c.w), c.x);
    case 4: return sub_rn(mul_rn(PBR_GUARD, c.w), c.y);
    default: return add_rn(mul_rn(PBR_GUARD, c.w), c.y);
  }
}

struct TileBox { int tx0, tx1, ty0, ty1; };

PBR_DEV bool setup_bounds(const SetupTri& t, const FrameParams* p, TileBox& box) {
  const int minX = min(t.X[0], min(t.X[1], t.X[2])), maxX = max(t.X[0], max(t.X[1], t.X[2]));
  const int minY = min(t.Y[0], min(t.Y[1], t.Y[2])), maxY = max(t.Y[0], max(t.Y[1], t.Y[2]));
  // pixel centres px*256+128 within [min, max]; with four samples per pixel, every pixel whose square meets [min, max]
  const int lo = p->msaa ? 0 : 127, hi = p->msaa ? 0 : 128;
  const int minx = max((minX + lo) >> 8, 0), maxx = min((maxX - hi) >> 8, p->W - 1);
  const int miny = max((minY + lo) >> 8, p->y0), maxy = min((maxY - hi) >> 8, p->y1 - 1);
  if (minx > maxx || miny > maxy) return false;
  box.tx0 = minx / PBR_TILE; box.tx1 = maxx / PBR_TILE;
  box.ty0 = (miny - p->y0) / PBR_TILE; box.ty1 = (maxy - p->y0) / PBR_TILE;
  return true;
}

// (BinTargets, where set-up triangles and tile references go, is declared in pbr_internal.h)

#ifdef PBR_SURFACE_TRACE
// Diagnostic build only: per workgroup of the set-up launch the 100 MHz timestamps of its start, of its triangles'
// vertices being transformed, of its record and set-up stores being issued, of its binning and of its end;
// pbr_debug_setup_trace copies them out.
__device__ unsigned long long g_setup_trace[16384][8];
#define SETUP_TRACE(slot) do { if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 1 && blockIdx.x < 16384) g_setup_trace[blockIdx.x][slot] = wall_clock64(); } while (0)
#define SETUP_TRACE_ANY(slot) do { const unsigned long long m_ = __ballot(1); if ((threadIdx.x >> 6) == 0 && (int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1 && blockIdx.x < 16384) g_setup_trace[blockIdx.x][slot] = wall_clock64(); } while (0)
#else
#define SETUP_TRACE_ANY(slot) do { } while (0)
#define SETUP_TRACE(slot) do { } while (0)
#endif

// Zeroes the append counters of the set the next frame counts in (by the first threads of one workgroup of the raster
// launch; the overflow flag [2] stays until the host has seen it, the covered pixels [3] are cleared where they are counted).
PBR_DEV void rearm_counters(uint32_t* __restrict__ next_counters, int tid) {
  if (tid == 0) { next_counters[0] = 0u; next_counters[1] = 0u; next_counters[4] = 0u; }
  if (tid < PBR_LIST_SEGMENTS) {
    uint32_t* group = next_counters + PBR_SEGMENT_FIRST + PBR_SEGMENT_WORDS * tid;
    group[0] = 0u; group[1] = 0u; group[2] = 0u;
  }
}

PBR_DEV void bin_store(const BinTargets& bt, uint32_t tile, uint32_t pos, uint32_t slot) {
  if (pos == PBR_DENSE_TILE_REFS) {   // the tile has just become one of the full ones: list it (see LiveTiles)
    const uint32_t k = atomicAdd(&bt.counters[4], 1u);
    if (k < bt.live.cap) bt.live.tiles[k] = (bt.live.serial << 24) | (unsigned long long)tile;
    bt.live.info[tile] = (bt.live.serial << 16) | (unsigned long long)min(k + 1u, 0xffffu);
  }
  if (pos < PBR_BIN_CAP) bt.bins[(size_t)tile * PBR_BIN_CAP + pos] = slot;
  else {
    const uint32_t o = atomicAdd(&bt.counters[1], 1u);
    if (o < bt.overflow_cap) bt.overflow[o] = make_uint2(tile, slot); else bt.counters[2] = 2u;
  }
}
PBR_DEV void bin_reference(const FrameParams* p, const BinTargets& bt, uint32_t tile, uint32_t slot) {
  bin_store(bt, tile, atomicAdd(&bt.tile_count[tile], 1u), slot);
}
// References tiles [first, first + 4) of a bounding box, in row-major order: the four position
// atomics are issued back to back so that their latencies overlap, then the four stores follow.
PBR_DEV void bin_four(const FrameParams* p, const BinTargets& bt, int tx0, int ty0, int bw, int n, int first, uint32_t slot) {
  uint32_t tile[4], pos[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = first + k;
    const int row = bw == 1 ? i : bw == 2 ? i >> 1 : i / bw;
    tile[k] = (uint32_t)((ty0 + row) * p->tiles_x + tx0 + (i - row * bw));
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) if (first + k < n) pos[k] = atomicAdd(&bt.tile_count[tile[k]], 1u);
#pragma unroll
  for (int k = 0; k < 4; ++k) if (first + k < n) bin_store(bt, tile[k], pos[k], slot);
}

// Snaps, culls and bounds one (sub)triangle; false when nothing of it can be covered.
PBR_DEV bool make_setup(uint32_t id, float4 a, float4 b, float4 c, const FrameParams* p, SetupTri& t, TileBox& box) {
  Snapped s0 = snap_vertex(a, p->W, p->H), s1 = snap_vertex(b, p->W, p->H), s2 = snap_vertex(c, p->W, p->H);
  const long long area2 = (long long)(s1.X - s0.X) * (s2.Y - s0.Y) - (long long)(s2.X - s0.X) * (s1.Y - s0.Y);
  if (area2 >= 0) return false;   // clockwise on screen or degenerate: culled (cullMode "back", frontFace "ccw")
  t.X[0] = s0.X; t.Y[0] = s0.Y; t.X[1] = s2.X; t.Y[1] = s2.Y; t.X[2] = s1.X; t.Y[2] = s1.Y;   // reorder: edge functions positive inside
  t.z0 = s0.z; t.dz1 = sub_rn(s2.z, s0.z); t.dz2 = sub_rn(s1.z, s0.z);
  t.inv_area = div_rn(1.0f, (float)(-area2));
  t.id = id; t.pad = 0;
  return setup_bounds(t, p, box);
}

// A piece of a clipped triangle (rare): appended and binned by its own thread, one returning atomic after the other.
PBR_DEV void emit_piece(uint32_t id, float4 a, float4 b, float4 c, const FrameParams* p, const BinTargets& bt) {
  SetupTri t;
  TileBox box;
  if (!make_setup(id, a, b, c, p, t, box)) return;
  const uint32_t slot = (uint32_t)p->n_tris * (uint32_t)p->n_instances + atomicAdd(&bt.counters[0], 1u);   // behind the whole triangles' entries (bin_wave)
  if (slot >= bt.setup_cap) { bt.counters[2] = 1u; return; }
  bt.setup[slot] = t;
  const int bw = box.tx1 - box.tx0 + 1, n_tiles = bw * (box.ty1 - box.ty0 + 1);
  for (int first = 0; first < n_tiles; first += 4) bin_four(p, bt, box.tx0, box.ty0, bw, n_tiles, first, slot);
}

// A whole triangle that survived culling, handed back by its thread so that its wavefront appends and bins its 64
// triangles together (bin_wave).
struct Pending { SetupTri t; TileBox box; int n_tiles; };   // n_tiles 0 = none

// Scratch of one wavefront for bin_wave: reference counts and then positions of a 16 x 16 window of tiles, the list of
// its tiles that have references, and the window's bounds.
struct BinWindow { uint32_t count[256]; uint32_t next[256]; uint32_t list[256]; int bounds[4]; };

PBR_DEV void wave_sync() {   // orders this wavefront's LDS traffic for the compiler; the hardware serves a wavefront's LDS instructions in order
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Enters the wavefront's pending triangles in the setup list and references them from the tiles of their bounding
// boxes.  A whole triangle's entry in the list is its own number: the list has a place for every triangle of every
// instance (and room behind them for the pieces of clipped ones), so nothing is reserved and no wavefront waits for an
// atomic on a counter all the others use too (see PBR_LIST_SEGMENTS; the count of the statistics goes to one of the
// segments' counters, and nobody waits for it).  An atomic on device memory that returns its value takes some 2 us
// here, and one per reference, from 64 lanes whose triangles are neighbours and so count on the same few tiles, took
// 6 us a round (measured per workgroup with the launch's timeline): the set-up launch spent more than half its time in
// them.  So the wavefront first counts its references per tile in LDS (the tiles of its triangles nearly always fit a
// 16 x 16 window), then one lane per tile adds the tile's count to the tile's counter in memory - a single round of
// atomics, every one on a different address - and the lanes take their positions from LDS again.  The order of the
// references inside a tile's row differs from run to run; the rasteriser's result does not depend on it (nearest
// fragment, lowest triangle number on ties).
PBR_DEV void bin_wave(const FrameParams* p, const BinTargets& bt, const Pending& pd, int lane, BinWindow* w) {
  const bool have = pd.n_tiles > 0;
  const unsigned long long have_mask = __ballot(have);
  if (!have_mask) return;
  if (lane == __ffsll((long long)have_mask) - 1)
    atomicAdd(bt.counters + PBR_SEGMENT_FIRST + PBR_SEGMENT_WORDS * (blockIdx.x % PBR_LIST_SEGMENTS) + 2, (uint32_t)__popcll(have_mask));
  const bool small = have && pd.n_tiles <= 16;
  bool window = false;
  int minx = 0, miny = 0;
  if (__ballot(small)) {
    reinterpret_cast<uint4*>(w->count)[lane] = make_uint4(0u, 0u, 0u, 0u);
    if (lane < 4) w->bounds[lane] = lane < 2 ? 0x7fffffff : -1;
    wave_sync();
    if (small) { atomicMin(&w->bounds[0], pd.box.tx0); atomicMin(&w->bounds[1], pd.box.ty0); atomicMax(&w->bounds[2], pd.box.tx1); atomicMax(&w->bounds[3], pd.box.ty1); }
    wave_sync();
    minx = w->bounds[0]; miny = w->bounds[1];
    window = w->bounds[2] - minx < 16 && w->bounds[3] - miny < 16;   // then (ty & 15, tx & 15) is a different entry for every tile
  }
  if (window) {
    if (small)
      for (int ty = pd.box.ty0; ty <= pd.box.ty1; ++ty)
        for (int tx = pd.box.tx0; tx <= pd.box.tx1; ++tx) atomicAdd(&w->count[((ty & 15) << 4) | (tx & 15)], 1u);
    wave_sync();
    // the window's tiles that have references, gathered into a list so that one lane takes one tile and the atomics
    // on the tiles' counters are a single instruction (four conditional ones, an entry each, wait for one another)
    uint32_t count[4], at[4], total = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      count[k] = w->count[lane + 64 * k];
      const unsigned long long m = __ballot(count[k] != 0u);
      at[k] = total + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      total += (uint32_t)__popcll(m);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) if (count[k]) w->list[at[k]] = (uint32_t)(lane + 64 * k) | (count[k] << 8);
    wave_sync();
    for (uint32_t j = (uint32_t)lane; j < total; j += 64u) {
      const uint32_t item = w->list[j], entry = item & 255u;
      const int tx = minx + ((((int)entry & 15) - minx) & 15), ty = miny + ((((int)entry >> 4) - miny) & 15);
      w->next[entry] = atomicAdd(&bt.tile_count[ty * p->tiles_x + tx], item >> 8);
    }
    wave_sync();
  }
  SETUP_TRACE(5);
  const uint32_t slot = pd.t.id;
  const bool fits = slot < bt.setup_cap;
  if (have && !fits) bt.counters[2] = 1u;   // the frame is drawn again with a larger list
  if (have && fits) bt.setup[slot] = pd.t;
  if (small) {
    if (window) {
      // the positions were counted above whether or not the triangle fits the list: one that does not leaves a
      // reference to the list's first entry, which is memory the rasteriser may read, in a frame that is not kept
      const uint32_t ref = fits ? slot : 0u;
      for (int ty = pd.box.ty0; ty <= pd.box.ty1; ++ty)
        for (int tx = pd.box.tx0; tx <= pd.box.tx1; ++tx)
          bin_store(bt, (uint32_t)(ty * p->tiles_x + tx), atomicAdd(&w->next[((ty & 15) << 4) | (tx & 15)], 1u), ref);
    } else if (fits) {
      const int bw = pd.box.tx1 - pd.box.tx0 + 1;
      for (int first = 0; first < pd.n_tiles; first += 4) bin_four(p, bt, pd.box.tx0, pd.box.ty0, bw, pd.n_tiles, first, slot);
    }
  }
  SETUP_TRACE(6);
  // triangles that span more than sixteen tiles: the whole wavefront bins each, one tile per lane
  unsigned long long todo = __ballot(have && !small && fits);
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)slot, src);
    const int tx0 = __builtin_amdgcn_readlane(pd.box.tx0, src), tx1 = __builtin_amdgcn_readlane(pd.box.tx1, src);
    const int ty0 = __builtin_amdgcn_readlane(pd.box.ty0, src), ty1 = __builtin_amdgcn_readlane(pd.box.ty1, src);
    const int bw = tx1 - tx0 + 1, n = bw * (ty1 - ty0 + 1);
    for (int i = lane; i < n; i += 64) bin_reference(p, bt, (uint32_t)((ty0 + i / bw) * p->tiles_x + tx0 + i % bw), s);
  }
}

// The triangle as the shading resolve wants it.  The snapped coordinates are the rasteriser's own (same
// arithmetic), so the barycentrics of a pixel agree with the coverage test that produced it.
PBR_DEV void write_record(TriRecord* __restrict__ dst, const VertexOut& a, const VertexOut& b, const VertexOut& c, int W, int H) {
  const float4 ca = a.clip, cb = b.clip, cc = c.clip;
  bool projective = ca.w > 0.0f && cb.w > 0.0f && cc.w > 0.0f && fabsf(ca.x) <= mul_rn(PBR_GUARD, ca.w) && fabsf(ca.y) <= mul_rn(PBR_GUARD, ca.w) &&
                    fabsf(cb.x) <= mul_rn(PBR_GUARD, cb.w) && fabsf(cb.y) <= mul_rn(PBR_GUARD, cb.w) && fabsf(cc.x) <= mul_rn(PBR_GUARD, cc.w) &&
                    fabsf(cc.y) <= mul_rn(PBR_GUARD, cc.w);
  int X0 = 0, Y0 = 0, X1 = 0, Y1 = 0, X2 = 0, Y2 = 0;
  float inv_area = 0.0f;
  if (projective) {
    const Snapped s0 = snap_vertex(ca, W, H), s1 = snap_vertex(cb, W, H), s2 = snap_vertex(cc, W, H);
    X0 = s0.X; Y0 = s0.Y; X1 = s1.X; Y1 = s1.Y; X2 = s2.X; Y2 = s2.Y;
    const long long area = (long long)(X1 - X0) * (Y2 - Y0) - (long long)(X2 - X0) * (Y1 - Y0);
    if (area == 0) projective = false; else inv_area = rcp_fast((float)area);
  }
  float4* d = reinterpret_cast<float4*>(dst);
  d[0] = make_float4(__int_as_float(X0), __int_as_float(Y0), __int_as_float(X1), __int_as_float(Y1));
  d[1] = make_float4(__int_as_float(X2), __int_as_float(Y2), inv_area, __int_as_float(projective ? 1 : 0));
  d[2] = make_float4(rcp_fast(ca.w), rcp_fast(cb.w), rcp_fast(cc.w), 0.0f);
  d[3] = make_float4(a.world_u.x, a.world_u.y, a.world_u.z, b.world_u.x);
  d[4] = make_float4(b.world_u.y, b.world_u.z, c.world_u.x, c.world_u.y);
  d[5] = make_float4(c.world_u.z, a.nrm_v.x, a.nrm_v.y, a.nrm_v.z);
  d[6] = make_float4(b.nrm_v.x, b.nrm_v.y, b.nrm_v.z, c.nrm_v.x);
  d[7] = make_float4(c.nrm_v.y, c.nrm_v.z, a.world_u.w, a.nrm_v.w);
  d[8] = make_float4(b.world_u.w, b.nrm_v.w, c.world_u.w, c.nrm_v.w);
  if (!projective) { d[9] = ca; d[10] = cb; d[11] = cc; }   // only the homogeneous fallback of the resolve reads these
}

// The three vertices of triangle t of instance inst after the vertex stage: transformed here from the vertex streams,
// or read from the vertex launch's output.
PBR_DEV void fetch_triangle(const FrameParams* p, const VertexOut* vout, const uint32_t* indices, const BinTargets& bt, int inst, int t, VertexOut& va,
                            VertexOut& vb, VertexOut& vc) {
  const size_t base = (size_t)inst * p->n_verts;
  const int ia = (int)indices[3 * (size_t)t], ib = (int)indices[3 * (size_t)t + 1], ic = (int)indices[3 * (size_t)t + 2];
  if (bt.vsoa) {
    VertexMatrices m;
    load_matrices(p, bt.inst_model, bt.inst_normal, inst, m);
    va = transform_vertex(m, bt.vsoa, (size_t)p->n_verts, ia); vb = transform_vertex(m, bt.vsoa, (size_t)p->n_verts, ib);
    vc = transform_vertex(m, bt.vsoa, (size_t)p->n_verts, ic);
  } else {
    va = vout[base + ia]; vb = vout[base + ib]; vc = vout[base + ic];
  }
}

// Whether instance `inst` lies wholly above, below, left or right of the frame's rows and columns: the eight corners
// of the mesh's bounding box go through the instance's transform (lanes 0 - 7 of every group of eight take one each,
// three butterfly steps give every lane the extremes); with every corner in front of the eye their screen positions
// bound those of all the instance's vertices.  Two pixels of margin stand for the rounding of this arithmetic against
// the vertex stage's: the test only ever skips work whose triangles the row and column test of their own bounding
// boxes would reject.  (For a rank that draws one strip of a large instanced scene this is most of the scene.)
PBR_DEV bool instance_misses_frame(const FrameParams* __restrict__ p, const BinTargets& bt, int inst) {
  if (!(bt.mesh_lo[0] <= bt.mesh_hi[0])) return false;   // no box known
  const int c = threadIdx.x & 7;
  const float x = (c & 1) ? bt.mesh_hi[0] : bt.mesh_lo[0], y = (c & 2) ? bt.mesh_hi[1] : bt.mesh_lo[1], z = (c & 4) ? bt.mesh_hi[2] : bt.mesh_lo[2];
  const float* m = p->has_instance_mats ? bt.inst_model + 16 * inst : p->u.model;
  const float wx = m[0] * x + m[4] * y + m[8] * z + m[12], wy = m[1] * x + m[5] * y + m[9] * z + m[13], wz = m[2] * x + m[6] * y + m[10] * z + m[14],
              ww = m[3] * x + m[7] * y + m[11] * z + m[15];
  const float* pv = p->pv;
  const float cx = pv[0] * wx + pv[4] * wy + pv[8] * wz + pv[12] * ww, cy = pv[1] * wx + pv[5] * wy + pv[9] * wz + pv[13] * ww,
              cw = pv[3] * wx + pv[7] * wy + pv[11] * wz + pv[15] * ww;
  const bool in_front = cw > 1e-6f;
  if (__ballot(!in_front) != 0ull) return false;   // a corner at or behind the eye (or a NaN): its screen position bounds nothing
  float sx = (cx / cw + 1.0f) * 0.5f * (float)p->W, sy = (1.0f - cy / cw) * 0.5f * (float)p->H;
  float xmin = sx, xmax = sx, ymin = sy, ymax = sy;
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    xmin = fminf(xmin, __shfl_xor(xmin, o)); xmax = fmaxf(xmax, __shfl_xor(xmax, o));
    ymin = fminf(ymin, __shfl_xor(ymin, o)); ymax = fmaxf(ymax, __shfl_xor(ymax, o));
  }
  const float margin = 2.0f;
  return ymax < (float)p->y0 - margin || ymin > (float)p->y1 + margin || xmax < -margin || xmin > (float)p->W + margin;
}

// One triangle per thread: trivial rejection, the shading record, and the triangle set up for the rasteriser in
// `pending` (left empty when it is culled).  Returns true for a triangle that crosses a clip plane: its thread clips
// it afterwards (clip_one), away from the common path, whose wavefronts then never wait for what only the clipping
// leaves in flight.
PBR_DEV bool setup_one(const FrameParams* p, const VertexOut* vout, const uint32_t* indices, const BinTargets& bt, int inst, int t, int n_tris,
                       Pending& pending) {
  VertexOut va, vb, vc;
  fetch_triangle(p, vout, indices, bt, inst, t, va, vb, vc);
  const float4 a = va.clip, b = vb.clip, c = vc.clip;
#ifdef PBR_SURFACE_TRACE
  asm volatile("" :: "v"(a.x), "v"(b.x), "v"(c.x));
#endif
  SETUP_TRACE(1);
  const uint32_t id = (uint32_t)inst * (uint32_t)n_tris + (uint32_t)t;
  const float4 v[3] = {a, b, c};
  bool inside_all = true;
#pragma unroll
  for (int pl = 0; pl < 6; ++pl) {
    int outside = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) outside += plane_dist(pl, v[k]) < 0.0f;
    if (outside == 3) return false;
    if (outside) inside_all = false;
  }
  {
    int l = 0, r = 0, bt_ = 0, tp = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { l += v[k].x < -v[k].w; r += v[k].x > v[k].w; bt_ += v[k].y < -v[k].w; tp += v[k].y > v[k].w; }
    if (l == 3 || r == 3 || bt_ == 3 || tp == 3) return false;   // beside the viewport: no pixel centre can be covered
  }
  // The shading record is only ever read through a pixel that shows the triangle: a triangle that is culled (turned
  // away, without area, or between pixel centres) needs none, and in a scene of small triangles that is every other
  // one (144 bytes each).  A triangle that crosses a clip plane gets its record before it is known which of its pieces
  // survive.
  if (!inside_all) { write_record(bt.records + id, va, vb, vc, p->W, p->H); SETUP_TRACE(2); return true; }
  if (make_setup(id, a, b, c, p, pending.t, pending.box)) {
    pending.n_tiles = (pending.box.tx1 - pending.box.tx0 + 1) * (pending.box.ty1 - pending.box.ty0 + 1);
    write_record(bt.records + id, va, vb, vc, p->W, p->H);
  }
  SETUP_TRACE(2);
  return false;
}

// Rare path: Sutherland-Hodgman against the six planes, then a fan from the first vertex.
PBR_DEV void clip_one(const FrameParams* p, const VertexOut* vout, const uint32_t* indices, const BinTargets& bt, int inst, int t, int n_tris) {
  VertexOut va, vb, vc;
  fetch_triangle(p, vout, indices, bt, inst, t, va, vb, vc);
  const uint32_t id = (uint32_t)inst * (uint32_t)n_tris + (uint32_t)t;
  float4 poly[12], tmp[12];
  int n = 3;
  poly[0] = va.clip; poly[1] = vb.clip; poly[2] = vc.clip;
  for (int pl = 0; pl < 6 && n >= 3; ++pl) {
    int m = 0;
    for (int k = 0; k < n; ++k) {
      const float4 cur = poly[k], nxt = poly[(k + 1) % n];
      const float dc = plane_dist(pl, cur), dn = plane_dist(pl, nxt);
      if (dc >= 0.0f) tmp[m++] = cur;
      if ((dc >= 0.0f) != (dn >= 0.0f)) {
        const float tt = div_rn(dc, sub_rn(dc, dn));
        tmp[m++] = make_float4(add_rn(cur.x, mul_rn(sub_rn(nxt.x, cur.x), tt)), add_rn(cur.y, mul_rn(sub_rn(nxt.y, cur.y), tt)),
                               add_rn(cur.z, mul_rn(sub_rn(nxt.z, cur.z), tt)), add_rn(cur.w, mul_rn(sub_rn(nxt.w, cur.w), tt)));
      }
    }
    n = m;
    for (int k = 0; k < n; ++k) poly[k] = tmp[k];
  }
  for (int k = 1; k + 1 < n; ++k) emit_piece(id, poly[0], poly[k], poly[k + 1], p, bt);
}

__global__ void __launch_bounds__(256) setup_kernel(const FrameParams frame_params, const VertexOut* __restrict__ vout,
                                                    const uint32_t* __restrict__ indices, BinTargets bt, int n_tris, int blocks_per_instance,
                                                    SkyShare share) {
  const FrameParams* __restrict__ p = &frame_params;   // parameters travel as kernel arguments: no per-frame copy to device memory
  SETUP_TRACE(0);
  if ((int)blockIdx.x >= share.own_blocks) { sky_tile(p, share.sky, share.rgba8, share.first + (int)blockIdx.x - share.own_blocks); SETUP_TRACE(4); return; }
  const int inst = (int)blockIdx.x / blocks_per_instance;
  const int t = ((int)blockIdx.x - inst * blocks_per_instance) * 256 + threadIdx.x;
  if (instance_misses_frame(p, bt, inst)) return;   // (the same answer in every wavefront of the workgroup)
  __shared__ BinWindow s_window[4];
  Pending pending;
  pending.n_tiles = 0;
  const bool crosses = t < n_tris && setup_one(p, vout, indices, bt, inst, t, n_tris, pending);
  SETUP_TRACE(3);
  bin_wave(p, bt, pending, threadIdx.x & 63, &s_window[threadIdx.x >> 6]);
  if (crosses) clip_one(p, vout, indices, bt, inst, t, n_tris);
  SETUP_TRACE(4);
}

// The same launch for several views of one scene: blockIdx.y is the view, whose parameters and targets are its entries
// of the argument arrays; the triangles always come from the vertex streams (vout is null; it is a kernel argument,
// not a literal, because AMD clang 22 of ROCm 7.2 crashes in its control-flow simplification on the literal).
struct SetupViews { FrameParams fp[PBR_MAX_VIEWS]; BinTargets bt[PBR_MAX_VIEWS]; uint32_t* rgba8[PBR_MAX_VIEWS]; };
static_assert(sizeof(SetupViews) + sizeof(SkyShare) + 64 <= 4096, "kernel arguments are limited to 4 KB");
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8)))
setup_views_kernel(const SetupViews sv, const VertexOut* __restrict__ vout, const uint32_t* __restrict__ indices, int n_tris, int blocks_per_instance,
                   SkyShare share) {
  const FrameParams* __restrict__ p = &sv.fp[blockIdx.y];
  if ((int)blockIdx.x >= share.own_blocks) { sky_tile(p, share.sky, sv.rgba8[blockIdx.y], share.first + (int)blockIdx.x - share.own_blocks); return; }
  const BinTargets& bt = sv.bt[blockIdx.y];
  const int inst = (int)blockIdx.x / blocks_per_instance;
  const int t = ((int)blockIdx.x - inst * blocks_per_instance) * 256 + threadIdx.x;
  if (instance_misses_frame(p, bt, inst)) return;
  __shared__ BinWindow s_window[4];
  Pending pending;
  pending.n_tiles = 0;
  const bool crosses = t < n_tris && setup_one(p, vout, indices, bt, inst, t, n_tris, pending);
  bin_wave(p, bt, pending, threadIdx.x & 63, &s_window[threadIdx.x >> 6]);
  if (crosses) clip_one(p, vout, indices, bt, inst, t, n_tris);
}

// ---------------------------------------------------------------------------------------------
// Tile rasteriser.  One 256-thread workgroup owns a 32x32 pixel tile whose depth / id buffer lives
// in LDS as one 64-bit key per pixel (order-preserving depth bits in the high word, triangle id + 1
// in the low word), so "nearest fragment, earliest triangle on ties" is a single ds_min_u64 and
// the depth buffer never touches HBM.  Work is triangle-parallel: each wavefront takes 64 bin
// entries at a time, one per lane, where the lane sets the triangle up against the tile (edge
// functions at the tile origin, exact tile-overlap test, bounding box inside the tile); the
// surviving lanes are then served one after the other — the lane's record is broadcast with
// readlane and the 64 lanes sweep the triangle's bounding box as 8x8 pixel blocks.
// ---------------------------------------------------------------------------------------------
PBR_DEV uint32_t depth_key(float z) {
  const uint32_t b = __float_as_uint(z);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // unsigned order == float order
}
PBR_DEV float key_depth(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

PBR_DEV int bcast(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
PBR_DEV long long bcast64(long long v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), src_lane), hi = __builtin_amdgcn_readlane((int)(v >> 32), src_lane);
  return ((long long)hi << 32) | (unsigned int)lo;
}

#ifdef PBR_SURFACE_TRACE
// Diagnostic build only: per tile of the rasteriser's last launch the 100 MHz timestamps of its workgroup's start, of the
// end of its triangle loop and of its end, and its number of bin entries; pbr_debug_raster_trace copies them out.
__device__ unsigned long long g_raster_trace[32768][8];
#define RASTER_TRACE(slot, value) do { if (tid == 0 && tile < 32768) g_raster_trace[tile][slot] = (value); } while (0)
#else
#define RASTER_TRACE(slot, value) do { } while (0)
#endif

constexpr int kRasterThreads = 256, kRasterWaves = kRasterThreads / 64;
// Small triangles (bounding box within the tile at most 16 x 16 pixels, 32-bit edge functions) are swept four at a time,
// each by sixteen lanes as 4 x 4 pixel blocks; a wavefront stages this many of them in LDS per round.
constexpr int kSmallBatch = 32, kSmallBoxPixels = 16;
// ... and the smallest one per lane: boxes of at most PBR_TINY_AREA pixels, when at least PBR_TINY_MIN_LANES lanes of
// the wavefront hold one.
#ifndef PBR_TINY_AREA
#define PBR_TINY_AREA 32
#endif
#ifndef PBR_TINY_MIN_LANES
#define PBR_TINY_MIN_LANES 16
#endif

// grid_row: the workgroup's row in its view's grid of (tiles_x, listed_rows + tiles_y) workgroups.
__device__ __forceinline__ void raster_tile(const FrameParams* __restrict__ p, const int grid_row, const SetupTri* __restrict__ setup,
                                            uint32_t* __restrict__ tile_count, const uint32_t* __restrict__ bins,
                                            const uint2* __restrict__ overflow, uint32_t overflow_cap, uint32_t* __restrict__ vis,
                                            float* __restrict__ depth, uint32_t* __restrict__ counters, uint32_t* __restrict__ next_counters,
                                            const QuadLists& lists, int write_vis, const DevCubeRgb& sky, uint32_t* __restrict__ rgba8, int sky_from_tile,
                                            const LiveTiles& live, int listed_rows, int listed_parts) {
  __shared__ unsigned long long s_key[PBR_TILE * PBR_TILE];
  __shared__ uint32_t s_pairs[kRasterWaves], s_base[2];
  // staging of the small triangles a wavefront has set up, for the four-at-a-time sweep below: 16 dwords per triangle
  __shared__ uint4 s_tri[kRasterWaves][kSmallBatch][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // The frame's append counters come in two sets used by alternate frames; this launch re-arms the set of
  // the next frame, which nothing is using (the previous frame, its last user, finished before this launch began).
  if (blockIdx.x == 0 && grid_row == 0) rearm_counters(next_counters, tid);
  // The first rows of workgroups take the tiles the set-up launch listed as the fullest: the frame waits for
  // those, so they must not queue behind thousands of skybox tiles (the fullest tile of the 4K benchmark frame used to
  // start 7 us into the launch and end it at 38 us).  Every tile also has a workgroup of its own after those, which
  // leaves at once if the tile was taken from the list.  Only the full tiles go first: all tiles with geometry at the
  // head of every launch cost 4 us per frame with four frames in flight, whose rasterisers then hold the chip's
  // wave slots with latency-bound loops at the same time.
  // With the chip to itself (listed_parts 2) a listed tile is rasterised by two workgroups, its upper and its lower
  // sixteen rows: the time of a full tile is that of its per-triangle loop, one wavefront issuing an instruction every
  // 7 - 8 cycles, and two workgroups halve the blocks each sweeps (not the triangles each sets up): rasteriser 36 -> 31 us,
  // single 4K frame 82.4 -> 78.2 us.  With other frames in flight the doubled set-up work costs more than the shorter
  // tail gains (53.3 against 52.7 us per frame), and one workgroup takes the whole tile (listed_parts 1).
  // A listed tile's reference count is therefore read by listed_parts + 1 workgroups, the parts' and the tile's own, in
  // any order, and has to be zero again for the next frame: each counts its visit in the count's top two bits, and the
  // last does the clearing.
  constexpr uint32_t kVisit = 1u << 30, kCountMask = kVisit - 1u;
  // (the listed tiles' workgroups are whole rows of the grid ahead of the tiles' own rows, so that a tile's own
  // workgroup finds its tile coordinates in its block index without a division)
  int tile, tile_x, tile_y, first_row = 0, rows = PBR_TILE;
  uint32_t n;
  bool clear_count = true;
  uint32_t first_ref = 0u;
  bool have_first_ref = false;
  if (grid_row < listed_rows) {
    // Every memory round trip here is 1.5 - 2 us on the frame's critical path (the launch ends with the fullest of
    // these tiles): one load says whether there is a k-th listed tile and which, and the tile's first references
    // are fetched next to the atomic that returns how many of them count.
    const uint32_t block = (uint32_t)grid_row * gridDim.x + blockIdx.x, parts = (uint32_t)listed_parts, k = parts == 2u ? block >> 1 : block;
    if (k >= live.cap || parts * k + parts - 1u >= (uint32_t)listed_rows * gridDim.x) return;   // (a lone last workgroup has no partner: the tile's own takes it)
    const unsigned long long listed = live.tiles[k];
    if ((listed >> 24) != live.serial) return;
    tile = (int)(listed & 0xffffffull);
    tile_y = p->tiles_x == 1 ? tile : (int)__umulhi((uint32_t)tile, p->tiles_x_magic); tile_x = tile - tile_y * p->tiles_x;
    rows = PBR_TILE / (int)parts; first_row = (int)(block & (parts - 1u)) * rows;
    if (tid == 0) s_base[0] = atomicAdd(&tile_count[tile], kVisit);
    first_ref = bins[(size_t)tile * PBR_BIN_CAP + (uint32_t)kRasterWaves * lane + wave];   // the entry of this lane in the first round below
    have_first_ref = true;
    __syncthreads();
    n = s_base[0] & kCountMask;
    clear_count = (s_base[0] >> 30) == parts;
    __syncthreads();   // s_base is used again below
  } else {
    // The rows of tiles are handed out from the middle of the target outwards (a scene's geometry is mostly near it):
    // workgroups start in grid order, and a tile with triangles that starts behind thousands of background tiles
    // ends the launch that much later.
    const int k = grid_row - listed_rows, mid = p->tiles_y >> 1;
    tile_x = (int)blockIdx.x; tile_y = (k & 1) ? mid - 1 - (k >> 1) : mid + (k >> 1);
    tile = tile_y * p->tiles_x + tile_x;
    n = tile_count[tile] & kCountMask;
    if (n > PBR_DENSE_TILE_REFS) {
      const unsigned long long v = live.info[tile];
      const uint32_t k = (uint32_t)(v & 0xffffu) - 1u;   // position in the list
      const uint32_t parts = (uint32_t)listed_parts;
      if ((v >> 16) == live.serial && parts * k + parts - 1u < (uint32_t)listed_rows * gridDim.x && k < live.cap) {
        if (tid == 0 && (atomicAdd(&tile_count[tile], kVisit) >> 30) == parts) tile_count[tile] = 0u;
        return;
      }
    }
  }
  const int px0 = tile_x * PBR_TILE, py0 = p->y0 + tile_y * PBR_TILE + first_row;   // origin in pixels of the tile, or of the half
  RASTER_TRACE(0, wall_clock64()); RASTER_TRACE(3, (unsigned long long)n);
  // the first PBR_BIN_CAP references sit in the tile's own row; a fuller tile also filters the shared overflow list
  const uint32_t n_row = min(n, (uint32_t)PBR_BIN_CAP);
  const uint32_t n_entries = n_row + (n > PBR_BIN_CAP ? min(counters[1], overflow_cap) : 0u);
  const int xmax = min(px0 + PBR_TILE, p->W) - 1, ymax = min(py0 + rows, p->y1) - 1;
  if (py0 > ymax) {   // the lower half of a tile that the target's edge cuts above it
    if (tid == 0 && clear_count) tile_count[tile] = 0u;
    return;
  }
  const unsigned long long clear_key = (unsigned long long)depth_key(1.0f) << 32;   // depth clear 1.0, no triangle

  if (n != 0u) {
    for (int i = tid; i < PBR_TILE * PBR_TILE; i += kRasterThreads) s_key[i] = clear_key;
    __syncthreads();
    const long long ox = (long long)px0 * 256 + 128, oy = (long long)py0 * 256 + 128;
    for (uint32_t chunk = 0; chunk < n_entries; chunk += kRasterThreads) {
      // lane-parallel: fetch one triangle per lane and set it up against this tile.  Entry
      // chunk + 4*lane + wave goes to this lane, so a short bin is still spread over all four waves.
      bool keep = false;
      long long e0 = 0, e1 = 0, e2 = 0;
      int a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0, tl = 0, box = 0;
      float z0 = 0, dz1 = 0, dz2 = 0, inv_area = 0;
      uint32_t id = 0;
      const uint32_t entry = chunk + (uint32_t)kRasterWaves * lane + wave;
      uint32_t slot = 0xffffffffu;
      if (entry < n_row) slot = have_first_ref && chunk == 0u ? first_ref : bins[(size_t)tile * PBR_BIN_CAP + entry];
      else if (entry < n_entries) { const uint2 pr = overflow[entry - n_row]; if (pr.x == (uint32_t)tile) slot = pr.y; }
      if (slot != 0xffffffffu) {
        const SetupTri t = setup[slot];
        keep = true;
        long long e[3]; int a[3], b[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int ia = (k + 1) % 3, ib = (k + 2) % 3;
          const int dx = t.X[ib] - t.X[ia], dy = t.Y[ib] - t.Y[ia];
          a[k] = -dy; b[k] = dx;
          e[k] = (long long)dx * (oy - t.Y[ia]) - (long long)dy * (ox - t.X[ia]);
          const bool is_tl = dy < 0 || (dy == 0 && dx > 0);   // left edge, or top edge: pixels exactly on it are covered
          if (is_tl) tl |= 1 << k;
          // largest value of the edge function over the tile's pixel centres
          const long long ex = a[k] > 0 ? (long long)a[k] * ((xmax - px0) * 256) : 0;
          const long long ey = b[k] > 0 ? (long long)b[k] * ((ymax - py0) * 256) : 0;
          const long long emax = e[k] + ex + ey;
          if (emax < 0 || (emax == 0 && !is_tl)) keep = false;
        }
        e0 = e[0]; e1 = e[1]; e2 = e[2]; a0 = a[0]; a1 = a[1]; a2 = a[2]; b0 = b[0]; b1 = b[1]; b2 = b[2];
        z0 = t.z0; dz1 = t.dz1; dz2 = t.dz2; inv_area = t.inv_area; id = t.id;
        // bounding box of the covered pixel centres, clamped to the tile
        const int minX = min(t.X[0], min(t.X[1], t.X[2])), maxX = max(t.X[0], max(t.X[1], t.X[2]));
        const int minY = min(t.Y[0], min(t.Y[1], t.Y[2])), maxY = max(t.Y[0], max(t.Y[1], t.Y[2]));
        const int bx0 = max((minX - 128 + 255) >> 8, px0) - px0, bx1 = min((maxX - 128) >> 8, xmax) - px0;
        const int by0 = max((minY - 128 + 255) >> 8, py0) - py0, by1 = min((maxY - 128) >> 8, ymax) - py0;
        if (bx0 > bx1 || by0 > by1) keep = false;
        box = bx0 | (bx1 << 8) | (by0 << 16) | (by1 << 24);
        // 32-bit evaluation is exact when every edge value over the tile and both factors of the 24-bit multiplies fit
        bool narrow = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const long long reach = (long long)(abs(a[k]) + abs(b[k])) * (PBR_TILE * 256);
          const long long mag = e[k] < 0 ? -e[k] : e[k];
          narrow = narrow && abs(a[k]) < (1 << 23) && abs(b[k]) < (1 << 23) && mag + reach < (1ll << 31) - 1;
        }
        if (narrow) tl |= 8;
      }
      // Small triangles first, four at a time.  Serving a triangle to all 64 lanes costs the same ~60 instructions of
      // lane reads and scalar set-up whether it covers three pixels or three hundred, at the rate of a wavefront that
      // is alone on its SIMD, and an 8 x 8 block per step is mostly empty lanes for a triangle a few pixels wide.  Here
      // every lane that holds a small triangle writes its sixteen parameters to LDS once; then each group of sixteen
      // lanes takes every fourth staged triangle, reads its parameters back as a broadcast into vector registers and
      // walks its bounding box in 4 x 4 blocks, independently of the other three groups.  Same integers, same f32
      // operations per pixel as the one-triangle sweep below: the keys offered to ds_min_u64 are the same.
#ifdef PBR_NO_SMALL_SWEEP
      const bool small = false;
#else
      const bool small = keep && (tl & 8) && (((box >> 8) & 255) - (box & 255)) < kSmallBoxPixels && ((box >> 24) - ((box >> 16) & 255)) < kSmallBoxPixels;
#endif
      // The smallest of them, one per lane: when a good part of the wavefront holds triangles whose boxes are a few
      // pixels, every such lane walks its own box pixel by pixel, stepping the three edge functions by their
      // increments (the same integers as the products: nothing overflows between two pixel centres of the tile).
      // Sixty-four triangles then cost what the largest box among them costs, about thirty instructions a pixel,
      // where the groups of sixteen below spend forty and more per triangle.
      const int box_w = ((box >> 8) & 255) - (box & 255) + 1, box_h = (box >> 24) - ((box >> 16) & 255) + 1;
      const bool tiny_one = small && box_w * box_h <= PBR_TINY_AREA;
      const bool tiny = tiny_one && __popcll(__ballot(tiny_one)) >= PBR_TINY_MIN_LANES;
      if (tiny) {
        const int bx0 = box & 255, bx1 = (box >> 8) & 255, by1 = box >> 24;
        int x = bx0, y = (box >> 16) & 255;
        const int c1 = (tl & 2) ? 0 : -1, c2 = (tl & 4) ? 0 : -1;
        int r0 = (int)e0 + ((tl & 1) ? 0 : -1) + __mul24(a0, x * 256) + __mul24(b0, y * 256);
        int r1 = (int)e1 + __mul24(a1, x * 256) + __mul24(b1, y * 256), r2 = (int)e2 + __mul24(a2, x * 256) + __mul24(b2, y * 256);
        int f0 = r0, f1 = r1, f2 = r2;
        const uint32_t key_id = id + 1u;
        for (;;) {
          if ((f0 | (f1 + c1) | (f2 + c2)) >= 0) {
            const float l1 = mul_rn((float)f1, inv_area), l2 = mul_rn((float)f2, inv_area);
            const float z = add_rn(add_rn(z0, mul_rn(l1, dz1)), mul_rn(l2, dz2));
            atomicMin(&s_key[y * PBR_TILE + x], ((unsigned long long)depth_key(z) << 32) | key_id);
          }
          if (x < bx1) { ++x; f0 += a0 * 256; f1 += a1 * 256; f2 += a2 * 256; }
          else if (y < by1) { ++y; x = bx0; r0 += b0 * 256; r1 += b1 * 256; r2 += b2 * 256; f0 = r0; f1 = r1; f2 = r2; }
          else break;
        }
      }
      const unsigned long long small_mask = __ballot(small && !tiny);
      if (small_mask) {
        const int n_small = __popcll(small_mask), rank = __popcll(small_mask & ((1ull << lane) - 1ull));
        const int group = lane >> 4, gl = lane & 15, gx = gl & 3, gy = gl >> 2;
        for (int base = 0; base < n_small; base += kSmallBatch) {
          if (small && !tiny && rank >= base && rank < base + kSmallBatch) {
            uint4* rec = s_tri[wave][rank - base];
            const int c1 = (tl & 2) ? 0 : -1, c2 = (tl & 4) ? 0 : -1;
            rec[0] = make_uint4((uint32_t)((int)e0 + ((tl & 1) ? 0 : -1)), (uint32_t)(int)e1, (uint32_t)(int)e2, (uint32_t)a0);
            rec[1] = make_uint4((uint32_t)a1, (uint32_t)a2, (uint32_t)b0, (uint32_t)b1);
            rec[2] = make_uint4((uint32_t)b2, __float_as_uint(z0), __float_as_uint(dz1), __float_as_uint(dz2));
            rec[3] = make_uint4(__float_as_uint(inv_area), id + 1u, (uint32_t)box, (uint32_t)(c1 & 1) | ((uint32_t)(c2 & 1) << 1));
          }
          wave_sync();
          const int count = min(kSmallBatch, n_small - base);
          int slot = group;
          bool have = false;
          // f*: the three edge functions at this lane's pixel of the current block, r* at its pixel of the row of
          // blocks' first block.  They are stepped by four pixels (unsigned arithmetic: a lane beyond the box may run
          // out of the 32 bits the box's own pixels are known to fit, and its values are never looked at).
          int A0 = 0, A1 = 0, A2 = 0, B0 = 0, B1 = 0, B2 = 0, C1 = 0, C2 = 0, bx0 = 0, bx1 = 0, by1 = 0, bx = 0, by = 0;
          uint32_t f0 = 0, f1 = 0, f2 = 0, r0 = 0, r1 = 0, r2 = 0;
          float Z0 = 0, DZ1 = 0, DZ2 = 0, IA = 0;
          uint32_t ID = 0;
          for (;;) {
            if (!have && slot < count) {
              const uint4* rec = s_tri[wave][slot];
              const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
              A0 = (int)q0.w; A1 = (int)q1.x; A2 = (int)q1.y; B0 = (int)q1.z; B1 = (int)q1.w;
              B2 = (int)q2.x; Z0 = __uint_as_float(q2.y); DZ1 = __uint_as_float(q2.z); DZ2 = __uint_as_float(q2.w);
              IA = __uint_as_float(q3.x); ID = q3.y;
              bx0 = (int)(q3.z & 255u); bx1 = (int)((q3.z >> 8) & 255u); by = (int)((q3.z >> 16) & 255u); by1 = (int)(q3.z >> 24);
              C1 = -(int)(q3.w & 1u); C2 = -(int)((q3.w >> 1) & 1u);
              bx = bx0;
              const int xs = (bx0 + gx) * 256, ys = (by + gy) * 256;
              r0 = q0.x + (uint32_t)__mul24(A0, xs) + (uint32_t)__mul24(B0, ys);
              r1 = q0.y + (uint32_t)__mul24(A1, xs) + (uint32_t)__mul24(B1, ys);
              r2 = q0.z + (uint32_t)__mul24(A2, xs) + (uint32_t)__mul24(B2, ys);
              f0 = r0; f1 = r1; f2 = r2;
              slot += 4;
              have = true;
            }
            if (!__ballot(have)) break;
            if (have) {
              const int x = bx + gx, y = by + gy;
              if (x <= bx1 && y <= by1 && (int)(f0 | (f1 + (uint32_t)C1) | (f2 + (uint32_t)C2)) >= 0) {
                const float l1 = mul_rn((float)(int)f1, IA), l2 = mul_rn((float)(int)f2, IA);
                const float z = add_rn(add_rn(Z0, mul_rn(l1, DZ1)), mul_rn(l2, DZ2));
                atomicMin(&s_key[y * PBR_TILE + x], ((unsigned long long)depth_key(z) << 32) | ID);
              }
              bx += 4;
              if (bx <= bx1) { f0 += (uint32_t)A0 << 10; f1 += (uint32_t)A1 << 10; f2 += (uint32_t)A2 << 10; }
              else {
                bx = bx0; by += 4; have = by <= by1;
                r0 += (uint32_t)B0 << 10; r1 += (uint32_t)B1 << 10; r2 += (uint32_t)B2 << 10;
                f0 = r0; f1 = r1; f2 = r2;
              }
            }
          }
          wave_sync();   // the next round overwrites the staging
        }
      }
      unsigned long long todo = __ballot(keep && !small);   // (a tiny triangle is a small one)
      const int lx = lane & 7, ly = lane >> 3;
#ifdef PBR_SURFACE_TRACE
      if (chunk == 0) { RASTER_TRACE(4, wall_clock64()); RASTER_TRACE(5, (unsigned long long)__popcll(todo)); }
      unsigned long long trace_blocks = 0;
#endif
      while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long long E0 = bcast64(e0, src), E1 = bcast64(e1, src), E2 = bcast64(e2, src);
        const int A0 = bcast(a0, src), A1 = bcast(a1, src), A2 = bcast(a2, src), B0 = bcast(b0, src), B1 = bcast(b1, src), B2 = bcast(b2, src);
        const int TL = bcast(tl, src), BOX = bcast(box, src);
        const float Z0 = __int_as_float(bcast(__float_as_int(z0), src)), DZ1 = __int_as_float(bcast(__float_as_int(dz1), src)),
                    DZ2 = __int_as_float(bcast(__float_as_int(dz2), src)), IA = __int_as_float(bcast(__float_as_int(inv_area), src));
        const uint32_t ID = (uint32_t)bcast((int)id, src) + 1u;
        const int bx0 = BOX & 255, bx1 = (BOX >> 8) & 255, by0 = (BOX >> 16) & 255, by1 = (BOX >> 24) & 255;
        const long long bias0 = (TL & 1) ? 0 : -1, bias1 = (TL & 2) ? 0 : -1, bias2 = (TL & 4) ? 0 : -1;   // e > 0, or e >= 0 on top-left edges
        if (TL & 8) {
          // common case: small on-screen triangle, 32-bit edge functions with 24-bit multiplies
          const int e0n = (int)E0 + (int)bias0, e1n = (int)E1, e2n = (int)E2;
          const int c1 = (int)bias1, c2 = (int)bias2;
          for (int yy = by0; yy <= by1; yy += 8)
            for (int xx = bx0; xx <= bx1; xx += 8) {
#ifdef PBR_SURFACE_TRACE
              ++trace_blocks;
#endif
              const int x = xx + lx, y = yy + ly;
              if (x > bx1 || y > by1) continue;
              const int xs = x * 256, ys = y * 256;
              const int f0 = e0n + __mul24(A0, xs) + __mul24(B0, ys);
              const int f1 = e1n + __mul24(A1, xs) + __mul24(B1, ys);
              const int f2 = e2n + __mul24(A2, xs) + __mul24(B2, ys);
              if ((f0 | (f1 + c1) | (f2 + c2)) < 0) continue;
              const float l1 = mul_rn((float)f1, IA), l2 = mul_rn((float)f2, IA);
              const float z = add_rn(add_rn(Z0, mul_rn(l1, DZ1)), mul_rn(l2, DZ2));
              atomicMin(&s_key[y * PBR_TILE + x], ((unsigned long long)depth_key(z) << 32) | ID);
            }
        } else {
          for (int yy = by0; yy <= by1; yy += 8)
            for (int xx = bx0; xx <= bx1; xx += 8) {
              const int x = xx + lx, y = yy + ly;
              if (x > bx1 || y > by1) continue;
              const long long f0 = E0 + (long long)A0 * (x * 256) + (long long)B0 * (y * 256);
              const long long f1 = E1 + (long long)A1 * (x * 256) + (long long)B1 * (y * 256);
              const long long f2 = E2 + (long long)A2 * (x * 256) + (long long)B2 * (y * 256);
              if (((f0 + bias0) | (f1 + bias1) | (f2 + bias2)) < 0) continue;
              const float l1 = mul_rn((float)f1, IA), l2 = mul_rn((float)f2, IA);
              const float z = add_rn(add_rn(Z0, mul_rn(l1, DZ1)), mul_rn(l2, DZ2));
              atomicMin(&s_key[y * PBR_TILE + x], ((unsigned long long)depth_key(z) << 32) | ID);
            }
        }
      }
#ifdef PBR_SURFACE_TRACE
      if (chunk == 0) { RASTER_TRACE(6, wall_clock64()); RASTER_TRACE(7, trace_blocks); }
#endif
    }
    __syncthreads();
  }
  if (tid == 0 && clear_count) tile_count[tile] = 0u;   // leave the bin counters clear for the next frame
  RASTER_TRACE(1, wall_clock64());

  // write the tile out: thread t holds pixels 4t .. 4t+3 of the row-major tile
  const int ty = tid >> 3, tx = (tid & 7) * 4;
  const int y = py0 + ty, x = px0 + tx;
  const bool inside = y <= ymax && x <= xmax;
  uint32_t ids[4] = {0u, 0u, 0u, 0u};
  float zs[4] = {1.0f, 1.0f, 1.0f, 1.0f};
  if (n != 0u && inside) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned long long key = s_key[ty * PBR_TILE + tx + k];
      ids[k] = (x + k <= xmax) ? (uint32_t)key : 0u;
      zs[k] = key_depth((uint32_t)(key >> 32));
    }
  }
  if (inside) {
    const size_t o = (size_t)(y - p->y0) * p->W + x;
    const bool wide = x + 3 <= xmax && (o & 3) == 0;
    // the visibility buffer is only written when someone will read it (the caller, the statistics, the
    // general-sampler resolve); the specialised resolve works from the quad lists
    if (write_vis) {
      if (wide) {
        *reinterpret_cast<uint4*>(vis + o) = make_uint4(ids[0], ids[1], ids[2], ids[3]);
        if (depth) *reinterpret_cast<float4*>(depth + o) = make_float4(zs[0], zs[1], zs[2], zs[3]);
      } else {
        for (int k = 0; k < 4; ++k)
          if (x + k <= xmax) { vis[o + k] = ids[k]; if (depth) depth[o + k] = zs[k]; }
      }
    }
    // The background pixels of the tile get their final colour here (skybox.wgsl:24-33): most tiles hold no
    // geometry at all, and their workgroups would otherwise have nothing to do.
    // (tiles below sky_from_tile were already filled by the workgroups riding along in the vertex and setup launches)
    if (tile >= sky_from_tile && !(ids[0] && ids[1] && ids[2] && ids[3])) {
      uint32_t c[4];
      sky_pixels4(p, sky, x, y, c);
      if (wide && !(ids[0] | ids[1] | ids[2] | ids[3])) {
        store_colour4(rgba8 + o, c);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (x + k <= xmax && !ids[k]) rgba8[o + k] = c[k];
      }
    }
  }
  RASTER_TRACE(2, wall_clock64());
  if (n == 0u) return;

  // Work lists for the shading resolve, one entry per 2x2 pixel quad that holds a covered pixel.  A quad
  // whose four pixels show one triangle goes to the uniform list (position, id): its derivatives are
  // differences between its own four invocations.  Any other quad goes to the mixed list (position,
  // four ids): each of its covered pixels has to evaluate its triangle at the neighbouring pixels
  // itself.  The thread of an even row owns the two quads it shares with the thread one row below
  // (eight lanes further on); the lists are appended with two atomics per tile.
  uint32_t below[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) below[k] = (uint32_t)__shfl_down((int)ids[k], 8);
  const bool owner = inside && !(ty & 1);
  bool uni[2], mix[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint32_t a = ids[2 * q], b = ids[2 * q + 1], c = below[2 * q], d = below[2 * q + 1];
    const bool any = owner && (a | b | c | d) != 0u;
    uni[q] = any && a == b && a == c && a == d;
    mix[q] = any && !uni[q];
  }
  const unsigned long long bu0 = __ballot(uni[0]), bu1 = __ballot(uni[1]), bm0 = __ballot(mix[0]), bm1 = __ballot(mix[1]);
  const unsigned long long lower = (1ull << lane) - 1ull;
  if (lane == 0) s_pairs[wave] = (uint32_t)(__popcll(bu0) + __popcll(bu1)) | ((uint32_t)(__popcll(bm0) + __popcll(bm1)) << 16);
  __syncthreads();
  if (tid == 0) {
    const uint32_t tu = (s_pairs[0] & 0xffffu) + (s_pairs[1] & 0xffffu) + (s_pairs[2] & 0xffffu) + (s_pairs[3] & 0xffffu);
    const uint32_t tm = (s_pairs[0] >> 16) + (s_pairs[1] >> 16) + (s_pairs[2] >> 16) + (s_pairs[3] >> 16);
    // the tile appends to its segment of the lists (see PBR_LIST_SEGMENTS).  The segment's two counters are neighbours
    // on an 8-byte boundary: one 64-bit atomic reserves room in both (two 32-bit ones are issued one after the
    // other's answer, 2 us each at the end of every tile with geometry, the frame's fullest tile included)
    const uint32_t segment = (uint32_t)tile % PBR_LIST_SEGMENTS;
    unsigned long long at = 0ull;
    if (tu | tm)
      at = atomicAdd(reinterpret_cast<unsigned long long*>(counters + PBR_SEGMENT_FIRST + PBR_SEGMENT_WORDS * segment), (unsigned long long)tu | ((unsigned long long)tm << 32));
    const uint32_t first = segment * quad_segment_cap((uint32_t)(p->tiles_x * p->tiles_y));
    s_base[0] = first + (uint32_t)at; s_base[1] = first + (uint32_t)(at >> 32);
  }
  __syncthreads();
  uint32_t at_u = s_base[0], at_m = s_base[1];
  for (int w = 0; w < wave; ++w) { at_u += s_pairs[w] & 0xffffu; at_m += s_pairs[w] >> 16; }
  at_u += (uint32_t)(__popcll(bu0 & lower) + __popcll(bu1 & lower));
  at_m += (uint32_t)(__popcll(bm0 & lower) + __popcll(bm1 & lower));
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint32_t pos = (uint32_t)((x >> 1) + q) | ((uint32_t)((y - p->y0) >> 1) << 16);
    if (uni[q]) { lists.uniform[at_u] = make_uint2(pos, ids[2 * q]); ++at_u; }
    if (mix[q]) { lists.mixed_pos[at_m] = pos; lists.mixed_ids[at_m] = make_uint4(ids[2 * q], ids[2 * q + 1], below[2 * q], below[2 * q + 1]); ++at_m; }
  }
}

__global__ void __launch_bounds__(kRasterThreads) raster_kernel(const FrameParams frame_params, const SetupTri* __restrict__ setup,
                                                     uint32_t* __restrict__ tile_count, const uint32_t* __restrict__ bins,
                                                     const uint2* __restrict__ overflow, uint32_t overflow_cap, uint32_t* __restrict__ vis,
                                                     float* __restrict__ depth, uint32_t* __restrict__ counters, uint32_t* __restrict__ next_counters,
                                                     QuadLists lists, int write_vis, DevCubeRgb sky, uint32_t* __restrict__ rgba8, int sky_from_tile,
                                                     LiveTiles live, int listed_rows, int listed_parts) {
  // parameters travel as kernel arguments: no per-frame copy to device memory
  raster_tile(&frame_params, (int)blockIdx.y, setup, tile_count, bins, overflow, overflow_cap, vis, depth, counters, next_counters, lists, write_vis, sky, rgba8,
              sky_from_tile, live, listed_rows, listed_parts);
}

// The same launch for several views of one scene.  The grid is (tiles_x, n_views * (listed_rows + tiles_y)): the rows
// of workgroups that take the views' listed (fullest) tiles come first, view after view, so that no view's fullest
// tile queues behind another view's thousands of skybox tiles; then the views' own rows.  Neither the visibility nor
// the depth buffer is written on this path.
struct RasterViews { FrameParams fp[PBR_MAX_VIEWS]; RasterTargets ra[PBR_MAX_VIEWS]; };
static_assert(sizeof(RasterViews) + sizeof(DevCubeRgb) + 64 <= 4096, "kernel arguments are limited to 4 KB");
__global__ void __launch_bounds__(kRasterThreads) raster_views_kernel(const RasterViews rv, int n_views, DevCubeRgb sky, int sky_from_tile, int listed_rows,
                                                                      int tiles_y) {
  int row = (int)blockIdx.y, view;
  const int listed_all = n_views * listed_rows;
  if (row < listed_all) {
    view = (row >= listed_rows) + (row >= 2 * listed_rows) + (row >= 3 * listed_rows);   // n_views <= 4: no division
    row -= view * listed_rows;
  } else {
    row -= listed_all;
    view = (row >= tiles_y) + (row >= 2 * tiles_y) + (row >= 3 * tiles_y);
    row += listed_rows - view * tiles_y;
  }
  const RasterTargets& ra = rv.ra[view];
  raster_tile(&rv.fp[view], row, ra.setup, ra.tile_count, ra.bins, ra.overflow, ra.overflow_cap, nullptr, nullptr, ra.counters, ra.next_counters, ra.lists, 0,
              sky, ra.rgba8, sky_from_tile, ra.live, listed_rows, 1);
}

// ---------------------------------------------------------------------------------------------
// The rasteriser of the multisampled pipeline (multisample.count = 4, renderer.ts:492-501): the same
// tile scheme with four keys per pixel, one per sample point of the standard pattern
// (0.375, 0.125), (0.875, 0.375), (0.125, 0.625), (0.625, 0.875).  Coverage, depth and the depth
// test are per sample.  It writes four triangle ids per pixel ([y][x][sample]); the resolve that
// follows shades each triangle of a pixel once and blends by coverage.
// ---------------------------------------------------------------------------------------------
__constant__ int kSampleX[4] = {96, 224, 32, 160};   // 1/256 pixel, from the pixel's top-left corner
__constant__ int kSampleY[4] = {32, 96, 160, 224};

__global__ void __launch_bounds__(256) raster_msaa_kernel(const FrameParams frame_params, const SetupTri* __restrict__ setup,
                                                          uint32_t* __restrict__ tile_count, const uint32_t* __restrict__ bins,
                                                          const uint2* __restrict__ overflow, uint32_t overflow_cap, uint32_t* __restrict__ vis,
                                                          float* __restrict__ depth, uint32_t* __restrict__ counters, uint32_t* __restrict__ next_counters,
                                                          uint32_t* __restrict__ tile_live, int write_empty) {
  const FrameParams* __restrict__ p = &frame_params;
  __shared__ unsigned long long s_key[PBR_TILE * PBR_TILE * 4];
  const int tile = blockIdx.y * p->tiles_x + blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x == 0 && blockIdx.y == 0) rearm_counters(next_counters, tid);
  const int px0 = blockIdx.x * PBR_TILE, py0 = p->y0 + blockIdx.y * PBR_TILE;
  const uint32_t n = tile_count[tile];
  const uint32_t n_row = min(n, (uint32_t)PBR_BIN_CAP);
  const uint32_t n_entries = n_row + (n > PBR_BIN_CAP ? min(counters[1], overflow_cap) : 0u);
  const int xmax = min(px0 + PBR_TILE, p->W) - 1, ymax = min(py0 + PBR_TILE, p->y1) - 1;
  const unsigned long long clear_key = (unsigned long long)depth_key(1.0f) << 32;

  if (n != 0u) {
    for (int i = tid; i < PBR_TILE * PBR_TILE * 4; i += 256) s_key[i] = clear_key;
    __syncthreads();
    const long long ox = (long long)px0 * 256, oy = (long long)py0 * 256;   // the tile's corner: sample offsets are added per sample
    for (uint32_t chunk = 0; chunk < n_entries; chunk += 256) {
      bool keep = false;
      long long e0 = 0, e1 = 0, e2 = 0;
      int a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0, tl = 0, box = 0;
      float z0 = 0, dz1 = 0, dz2 = 0, inv_area = 0;
      uint32_t id = 0;
      const uint32_t entry = chunk + 4u * lane + wave;
      uint32_t slot = 0xffffffffu;
      if (entry < n_row) slot = bins[(size_t)tile * PBR_BIN_CAP + entry];
      else if (entry < n_entries) { const uint2 pr = overflow[entry - n_row]; if (pr.x == (uint32_t)tile) slot = pr.y; }
      if (slot != 0xffffffffu) {
        const SetupTri t = setup[slot];
        keep = true;
        long long e[3]; int a[3], b[3];
        bool narrow = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int ia = (k + 1) % 3, ib = (k + 2) % 3;
          const int dx = t.X[ib] - t.X[ia], dy = t.Y[ib] - t.Y[ia];
          a[k] = -dy; b[k] = dx;
          e[k] = (long long)dx * (oy - t.Y[ia]) - (long long)dy * (ox - t.X[ia]);
          if (dy < 0 || (dy == 0 && dx > 0)) tl |= 1 << k;
          // an upper bound of the edge function over the tile's sample points (which lie inside the pixel squares)
          const long long ex = a[k] > 0 ? (long long)a[k] * ((xmax - px0 + 1) * 256) : 0;
          const long long ey = b[k] > 0 ? (long long)b[k] * ((ymax - py0 + 1) * 256) : 0;
          if (e[k] + ex + ey < 0) keep = false;
          const long long reach = (long long)(abs(a[k]) + abs(b[k])) * (PBR_TILE * 256);
          const long long mag = e[k] < 0 ? -e[k] : e[k];
          narrow = narrow && abs(a[k]) < (1 << 23) && abs(b[k]) < (1 << 23) && mag + reach < (1ll << 31) - 1;
        }
        e0 = e[0]; e1 = e[1]; e2 = e[2]; a0 = a[0]; a1 = a[1]; a2 = a[2]; b0 = b[0]; b1 = b[1]; b2 = b[2];
        z0 = t.z0; dz1 = t.dz1; dz2 = t.dz2; inv_area = t.inv_area; id = t.id;
        const int minX = min(t.X[0], min(t.X[1], t.X[2])), maxX = max(t.X[0], max(t.X[1], t.X[2]));
        const int minY = min(t.Y[0], min(t.Y[1], t.Y[2])), maxY = max(t.Y[0], max(t.Y[1], t.Y[2]));
        const int bx0 = max(minX >> 8, px0) - px0, bx1 = min(maxX >> 8, xmax) - px0;
        const int by0 = max(minY >> 8, py0) - py0, by1 = min(maxY >> 8, ymax) - py0;
        if (bx0 > bx1 || by0 > by1) keep = false;
        box = bx0 | (bx1 << 8) | (by0 << 16) | (by1 << 24);
        if (narrow) tl |= 8;
      }
      unsigned long long todo = __ballot(keep);
      // a lane is one sample of one pixel: the wavefront sweeps the bounding box in 4 x 4 pixel blocks
      const int sm = lane & 3, lx = (lane >> 2) & 3, ly = lane >> 4;
      const int sx = kSampleX[sm], sy = kSampleY[sm];
      while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const long long E0 = bcast64(e0, src), E1 = bcast64(e1, src), E2 = bcast64(e2, src);
        const int A0 = bcast(a0, src), A1 = bcast(a1, src), A2 = bcast(a2, src), B0 = bcast(b0, src), B1 = bcast(b1, src), B2 = bcast(b2, src);
        const int TL = bcast(tl, src), BOX = bcast(box, src);
        const float Z0 = __int_as_float(bcast(__float_as_int(z0), src)), DZ1 = __int_as_float(bcast(__float_as_int(dz1), src)),
                    DZ2 = __int_as_float(bcast(__float_as_int(dz2), src)), IA = __int_as_float(bcast(__float_as_int(inv_area), src));
        const uint32_t ID = (uint32_t)bcast((int)id, src) + 1u;
        const int bx0 = BOX & 255, bx1 = (BOX >> 8) & 255, by0 = (BOX >> 16) & 255, by1 = (BOX >> 24) & 255;
        const int c0 = (TL & 1) ? 0 : -1, c1 = (TL & 2) ? 0 : -1, c2 = (TL & 4) ? 0 : -1;   // e > 0, or e >= 0 on top-left edges
        if (TL & 8) {
          // 32-bit edge functions with 24-bit multiplies are exact for this triangle over this tile
          const int e0n = (int)E0 + c0 + __mul24(A0, sx) + __mul24(B0, sy), e1n = (int)E1 + __mul24(A1, sx) + __mul24(B1, sy),
                    e2n = (int)E2 + __mul24(A2, sx) + __mul24(B2, sy);
          for (int yy = by0; yy <= by1; yy += 4)
            for (int xx = bx0; xx <= bx1; xx += 4) {
              const int x = xx + lx, y = yy + ly;
              if (x > bx1 || y > by1) continue;
              const int xs = x * 256, ys = y * 256;
              const int f0 = e0n + __mul24(A0, xs) + __mul24(B0, ys);
              const int f1 = e1n + __mul24(A1, xs) + __mul24(B1, ys);
              const int f2 = e2n + __mul24(A2, xs) + __mul24(B2, ys);
              if ((f0 | (f1 + c1) | (f2 + c2)) < 0) continue;
              const float l1 = mul_rn((float)f1, IA), l2 = mul_rn((float)f2, IA);
              const float z = add_rn(add_rn(Z0, mul_rn(l1, DZ1)), mul_rn(l2, DZ2));
              atomicMin(&s_key[(y * PBR_TILE + x) * 4 + sm], ((unsigned long long)depth_key(z) << 32) | ID);
            }
        } else {
          for (int yy = by0; yy <= by1; yy += 4)
            for (int xx = bx0; xx <= bx1; xx += 4) {
              const int x = xx + lx, y = yy + ly;
              if (x > bx1 || y > by1) continue;
              const int xs = x * 256 + sx, ys = y * 256 + sy;
              const long long f0 = E0 + (long long)A0 * xs + (long long)B0 * ys;
              const long long f1 = E1 + (long long)A1 * xs + (long long)B1 * ys;
              const long long f2 = E2 + (long long)A2 * xs + (long long)B2 * ys;
              if (((f0 + c0) | (f1 + c1) | (f2 + c2)) < 0) continue;
              const float l1 = mul_rn((float)f1, IA), l2 = mul_rn((float)f2, IA);
              const float z = add_rn(add_rn(Z0, mul_rn(l1, DZ1)), mul_rn(l2, DZ2));
              atomicMin(&s_key[(y * PBR_TILE + x) * 4 + sm], ((unsigned long long)depth_key(z) << 32) | ID);
            }
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0) { tile_count[tile] = 0u; tile_live[tile] = n; }   // leave the bin counters clear for the next frame
  // The resolve skips tiles that no triangle was binned to, so their entries are only written when the caller reads them.
  if (n == 0u && !write_empty) return;

  // write the tile out: thread t holds pixels 4t .. 4t+3 of the row-major tile, four samples each
  const int ty = tid >> 3, tx = (tid & 7) * 4;
  const int y = py0 + ty, x = px0 + tx;
  if (y > ymax || x > xmax) return;
  const size_t o = ((size_t)(y - p->y0) * p->W + x) * 4;
  for (int k = 0; k < 4 && x + k <= xmax; ++k) {
    uint4 ids = make_uint4(0u, 0u, 0u, 0u);
    float4 zs = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (n != 0u) {
      const unsigned long long* key = &s_key[(ty * PBR_TILE + tx + k) * 4];
      ids = make_uint4((uint32_t)key[0], (uint32_t)key[1], (uint32_t)key[2], (uint32_t)key[3]);
      zs = make_float4(key_depth((uint32_t)(key[0] >> 32)), key_depth((uint32_t)(key[1] >> 32)), key_depth((uint32_t)(key[2] >> 32)),
                       key_depth((uint32_t)(key[3] >> 32)));
    }
    *reinterpret_cast<uint4*>(vis + o + 4 * k) = ids;
    if (depth) *reinterpret_cast<float4*>(depth + o + 4 * k) = zs;
  }
}

}  // namespace

#ifdef PBR_SURFACE_TRACE
extern "C" int pbr_debug_setup_trace(unsigned long long* out, size_t bytes) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_setup_trace), bytes < sizeof(g_setup_trace) ? bytes : sizeof(g_setup_trace));
}
extern "C" int pbr_debug_raster_trace(unsigned long long* out, size_t bytes) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_raster_trace), bytes < sizeof(g_raster_trace) ? bytes : sizeof(g_raster_trace));
}
#endif

void launch_vertex_stage(hipStream_t s, const FrameParams& p, const float* vsoa, const float* inst_model, const float* inst_normal,
                         VertexOut* out, int n_verts, int n_instances, const SkyShare& share_in) {
  const int per_instance = (n_verts + 255) / 256;
  SkyShare share = share_in;
  share.own_blocks = per_instance * n_instances;
  hipLaunchKernelGGL(vertex_kernel, dim3(share.own_blocks + share.count), dim3(256), 0, s, p, vsoa, inst_model, inst_normal, out, n_verts, per_instance,
                     share);
}

void launch_setup_and_bin(hipStream_t s, const FrameParams& p, const VertexSource& src, const uint32_t* indices, SetupTri* setup, TriRecord* records,
                          uint32_t setup_cap, uint32_t* tile_count, uint32_t* bins, uint2* overflow, uint32_t overflow_cap, uint32_t* counters,
                          const LiveTiles& live, int n_tris, int n_instances, const SkyShare& share_in) {
  // tile_count is zero on entry (the raster kernel of the previous frame clears it) and so are the
  // frame's append counters (re-armed by the raster kernel of the previous frame)
  BinTargets bt;
  bt.setup = setup; bt.setup_cap = setup_cap; bt.tile_count = tile_count; bt.bins = bins; bt.overflow = overflow; bt.overflow_cap = overflow_cap;
  bt.counters = counters; bt.live = live; bt.records = records;
  bt.vsoa = src.vsoa; bt.inst_model = src.inst_model; bt.inst_normal = src.inst_normal;
  for (int k = 0; k < 3; ++k) { bt.mesh_lo[k] = src.mesh_lo[k]; bt.mesh_hi[k] = src.mesh_hi[k]; }
  const int per_instance = (n_tris + 255) / 256;
  SkyShare share = share_in;
  share.own_blocks = per_instance * n_instances;
  hipLaunchKernelGGL(setup_kernel, dim3(share.own_blocks + share.count), dim3(256), 0, s, p, src.vout, indices, bt, n_tris, per_instance, share);
}

void launch_raster(hipStream_t s, const FrameParams& p, const SetupTri* setup, uint32_t* tile_count, const uint32_t* bins, const uint2* overflow,
                   uint32_t overflow_cap, uint32_t* vis, float* depth, uint32_t* counters, uint32_t* next_counters, const QuadLists& lists, bool write_vis,
                   const DevCubeRgb& sky_bordered, uint8_t* rgba8, int sky_from_tile, int tiles_x, int tiles_y, const LiveTiles& live, bool split_listed) {
  const int n_tiles = tiles_x * tiles_y, listed_blocks = std::min(n_tiles, (int)live.cap);
  const int listed_parts = split_listed ? 2 : 1;
  const int listed_rows = (listed_parts * listed_blocks + tiles_x - 1) / tiles_x;
  hipLaunchKernelGGL(raster_kernel, dim3(tiles_x, listed_rows + tiles_y), dim3(kRasterThreads), 0, s, p, setup, tile_count, bins, overflow, overflow_cap, vis,
                     depth, counters, next_counters, lists, write_vis ? 1 : 0, sky_bordered, reinterpret_cast<uint32_t*>(rgba8), sky_from_tile, live, listed_rows, listed_parts);
}

void launch_raster_msaa(hipStream_t s, const FrameParams& p, const SetupTri* setup, uint32_t* tile_count, const uint32_t* bins, const uint2* overflow,
                        uint32_t overflow_cap, uint32_t* vis, float* depth, uint32_t* counters, uint32_t* next_counters, uint32_t* tile_live,
                        bool write_empty_tiles, int tiles_x, int tiles_y, const LiveTiles& live) {
  hipLaunchKernelGGL(raster_msaa_kernel, dim3(tiles_x, tiles_y), dim3(256), 0, s, p, setup, tile_count, bins, overflow, overflow_cap, vis, depth, counters,
                     next_counters, tile_live, write_empty_tiles ? 1 : 0);
}

void launch_setup_views(hipStream_t s, const ViewArgs* views, int n_views, const uint32_t* indices, int n_tris, int n_instances, const SkyShare& share_in) {
  const int per_instance = (n_tris + 255) / 256;
  SkyShare share = share_in;
  share.own_blocks = per_instance * n_instances;
  SetupViews sv = {};
  for (int v = 0; v < n_views; ++v) { sv.fp[v] = views[v].fp; sv.bt[v] = views[v].bt; sv.rgba8[v] = views[v].ra.rgba8; }
  hipLaunchKernelGGL(setup_views_kernel, dim3(share.own_blocks + share.count, n_views), dim3(256), 0, s, sv, static_cast<const VertexOut*>(nullptr), indices, n_tris,
                     per_instance, share);
}

void launch_raster_views(hipStream_t s, const ViewArgs* views, int n_views, const DevCubeRgb& sky_bordered, int sky_from_tile, int tiles_x, int tiles_y,
                         uint32_t live_cap) {
  const int n_tiles = tiles_x * tiles_y, listed_blocks = std::min(n_tiles, (int)live_cap);
  const int listed_rows = (listed_blocks + tiles_x - 1) / tiles_x;
  RasterViews rv = {};
  for (int v = 0; v < n_views; ++v) { rv.fp[v] = views[v].fp; rv.ra[v] = views[v].ra; }
  hipLaunchKernelGGL(raster_views_kernel, dim3(tiles_x, n_views * (listed_rows + tiles_y)), dim3(kRasterThreads), 0, s, rv, n_views, sky_bordered, sky_from_tile,
                     listed_rows, tiles_y);
}
