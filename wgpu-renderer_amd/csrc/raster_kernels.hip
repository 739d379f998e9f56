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

namespace {

// ---------------------------------------------------------------------------------------------
// Vertex stage.  Attribute streams are SoA (px[], py[], pz[], u[], v[], nx[], ny[], nz[]) so a
// wavefront's 64 loads of one attribute are one contiguous 256-byte request.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vertex_kernel(const FrameParams* __restrict__ p, const float* __restrict__ vsoa,
                                                     const float* __restrict__ inst_model, const float* __restrict__ inst_normal,
                                                     VertexOut* __restrict__ out, int n_verts) {
  const int inst = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_verts) return;
  float model[16], nmat[16], pvm[16];
  const float* msrc = p->has_instance_mats ? inst_model + 16 * inst : p->u.model;
  const float* nsrc = p->has_instance_mats ? inst_normal + 16 * inst : p->u.normal;
#pragma unroll
  for (int k = 0; k < 16; ++k) { model[k] = msrc[k]; nmat[k] = nsrc[k]; }
  mat4_mul_exact(p->pv, model, pvm);   // (projection * view) * model
  const size_t n = (size_t)n_verts;
  const float x = vsoa[i], y = vsoa[n + i], z = vsoa[2 * n + i];
  const float tu = vsoa[3 * n + i], tv = vsoa[4 * n + i];
  const float nx = vsoa[5 * n + i], ny = vsoa[6 * n + i], nz = vsoa[7 * n + i];
  VertexOut o;
  o.clip = mat4_vec_exact(pvm, x, y, z, 1.0f);
  float4 w = mat4_vec_exact(model, x, y, z, 1.0f);
  float4 nn = mat4_vec_exact(nmat, nx, ny, nz, 0.0f);
  o.world_u = make_float4(w.x, w.y, w.z, tu);
  o.nrm_v = make_float4(nn.x, nn.y, nn.z, tv);
  out[(size_t)inst * n + i] = o;
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
  // pixel centres px*256+128 within [min, max]
  const int minx = max((minX - 128 + 255) >> 8, 0), maxx = min((maxX - 128) >> 8, p->W - 1);
  const int miny = max((minY - 128 + 255) >> 8, p->y0), maxy = min((maxY - 128) >> 8, p->y1 - 1);
  if (minx > maxx || miny > maxy) return false;
  box.tx0 = minx / PBR_TILE; box.tx1 = maxx / PBR_TILE;
  box.ty0 = (miny - p->y0) / PBR_TILE; box.ty1 = (maxy - p->y0) / PBR_TILE;
  return true;
}

PBR_DEV void emit_triangle(uint32_t id, float4 a, float4 b, float4 c, const FrameParams* p, SetupTri* setup, uint32_t setup_cap,
                           uint32_t* tile_count, uint32_t* counters) {
  Snapped s0 = snap_vertex(a, p->W, p->H), s1 = snap_vertex(b, p->W, p->H), s2 = snap_vertex(c, p->W, p->H);
  const long long area2 = (long long)(s1.X - s0.X) * (s2.Y - s0.Y) - (long long)(s2.X - s0.X) * (s1.Y - s0.Y);
  if (area2 >= 0) return;   // clockwise on screen or degenerate: culled (cullMode "back", frontFace "ccw")
  SetupTri t;
  t.X[0] = s0.X; t.Y[0] = s0.Y; t.X[1] = s2.X; t.Y[1] = s2.Y; t.X[2] = s1.X; t.Y[2] = s1.Y;   // reorder: edge functions positive inside
  t.z0 = s0.z; t.dz1 = sub_rn(s2.z, s0.z); t.dz2 = sub_rn(s1.z, s0.z);
  t.inv_area = div_rn(1.0f, (float)(-area2));
  t.id = id; t.pad = 0;
  TileBox box;
  if (!setup_bounds(t, p, box)) return;
  const uint32_t slot = atomicAdd(&counters[0], 1u);
  if (slot >= setup_cap) { counters[2] = 1u; return; }
  setup[slot] = t;
  for (int ty = box.ty0; ty <= box.ty1; ++ty)
    for (int tx = box.tx0; tx <= box.tx1; ++tx) atomicAdd(&tile_count[ty * p->tiles_x + tx], 1u);
}

__global__ void __launch_bounds__(256) setup_kernel(const FrameParams* __restrict__ p, const VertexOut* __restrict__ vout,
                                                    const uint16_t* __restrict__ indices, SetupTri* __restrict__ setup, uint32_t setup_cap,
                                                    uint32_t* __restrict__ tile_count, uint32_t* __restrict__ counters, int n_tris) {
  const int inst = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_tris) return;
  const size_t base = (size_t)inst * p->n_verts;
  const float4 a = vout[base + indices[3 * t]].clip, b = vout[base + indices[3 * t + 1]].clip, c = vout[base + indices[3 * t + 2]].clip;
  const uint32_t id = (uint32_t)inst * (uint32_t)n_tris + (uint32_t)t;
  const float4 v[3] = {a, b, c};
  bool inside_all = true;
#pragma unroll
  for (int pl = 0; pl < 6; ++pl) {
    int outside = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) outside += plane_dist(pl, v[k]) < 0.0f;
    if (outside == 3) return;
    if (outside) inside_all = false;
  }
  {
    int l = 0, r = 0, bt = 0, tp = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { l += v[k].x < -v[k].w; r += v[k].x > v[k].w; bt += v[k].y < -v[k].w; tp += v[k].y > v[k].w; }
    if (l == 3 || r == 3 || bt == 3 || tp == 3) return;   // beside the viewport: no pixel centre can be covered
  }
  if (inside_all) { emit_triangle(id, a, b, c, p, setup, setup_cap, tile_count, counters); return; }
  // Rare path: Sutherland-Hodgman against the six planes, then a fan from the first vertex.
  float4 poly[12], tmp[12];
  int n = 3;
  poly[0] = a; poly[1] = b; poly[2] = c;
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
  for (int k = 1; k + 1 < n; ++k) emit_triangle(id, poly[0], poly[k], poly[k + 1], p, setup, setup_cap, tile_count, counters);
}

// Exclusive scan of the per-tile counts by one workgroup: each thread scans a contiguous chunk,
// the chunk totals are scanned through LDS.
__global__ void __launch_bounds__(1024) tile_scan_kernel(const uint32_t* __restrict__ count, uint32_t* __restrict__ offset,
                                                         uint32_t* __restrict__ cursor, uint32_t* __restrict__ counters, int n) {
  __shared__ uint32_t s_sum[1024];
  const int tid = threadIdx.x;
  const int chunk = (n + 1023) / 1024;
  const int begin = tid * chunk, end = min(begin + chunk, n);
  uint32_t sum = 0;
  for (int i = begin; i < end; ++i) sum += count[i];
  s_sum[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    uint32_t v = tid >= d ? s_sum[tid - d] : 0u;
    __syncthreads();
    s_sum[tid] += v;
    __syncthreads();
  }
  uint32_t run = s_sum[tid] - sum;
  for (int i = begin; i < end; ++i) { offset[i] = run; cursor[i] = 0u; run += count[i]; }
  if (tid == 1023) counters[1] = s_sum[1023];
}

__global__ void __launch_bounds__(256) bin_fill_kernel(const FrameParams* __restrict__ p, const SetupTri* __restrict__ setup, uint32_t setup_cap,
                                                       const uint32_t* __restrict__ tile_offset, uint32_t* __restrict__ tile_cursor,
                                                       uint32_t* __restrict__ bin_list, uint32_t bin_cap, uint32_t* __restrict__ counters) {
  const uint32_t n = min(counters[0], setup_cap);
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const SetupTri t = setup[i];
  TileBox box;
  if (!setup_bounds(t, p, box)) return;
  for (int ty = box.ty0; ty <= box.ty1; ++ty)
    for (int tx = box.tx0; tx <= box.tx1; ++tx) {
      const int tile = ty * p->tiles_x + tx;
      const uint32_t pos = tile_offset[tile] + atomicAdd(&tile_cursor[tile], 1u);
      if (pos < bin_cap) bin_list[pos] = i; else counters[2] = 2u;
    }
}

// ---------------------------------------------------------------------------------------------
// Tile rasteriser.  One 256-thread workgroup owns a 32x32 pixel tile; each thread owns a 2x2
// quad and keeps its four depth / id pairs in registers, so the depth buffer never touches
// memory during rasterisation.  The tile's bin is consumed in chunks of 256 triangles: every
// thread fetches one triangle, tests it exactly against the tile, and the survivors are compacted
// into an LDS-resident list with a wavefront ballot and prefix count; then all threads walk that
// list (LDS broadcast reads) and test their pixels with 64-bit edge functions.
// ---------------------------------------------------------------------------------------------
struct LdsTri {
  long long e[3];      // edge functions at the tile's first pixel centre
  int a[3], b[3];      // d e / d x, d e / d y per 1/256 pixel
  float z0, dz1, dz2, inv_area;
  uint32_t id;
  uint32_t tl;         // bit k set: edge k is a top or left edge (covers pixels exactly on it)
};

__global__ void __launch_bounds__(256) raster_kernel(const FrameParams* __restrict__ p, const SetupTri* __restrict__ setup,
                                                     const uint32_t* __restrict__ tile_count, const uint32_t* __restrict__ tile_offset,
                                                     const uint32_t* __restrict__ bin_list, uint32_t* __restrict__ vis, float* __restrict__ depth) {
  __shared__ LdsTri s_tri[256];
  __shared__ uint32_t s_wave_count[4];
  const int tile = blockIdx.y * p->tiles_x + blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px0 = blockIdx.x * PBR_TILE, py0 = p->y0 + blockIdx.y * PBR_TILE;   // tile origin in pixels
  const int qx = (tid & 15) * 2, qy = (tid >> 4) * 2;                           // this thread's quad inside the tile
  const uint32_t n = tile_count[tile], base = tile_offset[tile];
  const long long ox = (long long)px0 * 256 + 128, oy = (long long)py0 * 256 + 128;
  const int xmax = min(px0 + PBR_TILE, p->W) - 1, ymax = min(py0 + PBR_TILE, p->y1) - 1;

  float zmin[4] = {1.0f, 1.0f, 1.0f, 1.0f};
  uint32_t best[4] = {0u, 0u, 0u, 0u};

  for (uint32_t chunk = 0; chunk < n; chunk += 256) {
    // phase 1: fetch, exact tile test, compact
    bool keep = false;
    LdsTri lt;
    if (chunk + tid < n) {
      const SetupTri t = setup[bin_list[base + chunk + tid]];
      keep = true;
      lt.tl = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int ia = (k + 1) % 3, ib = (k + 2) % 3;
        const int dx = t.X[ib] - t.X[ia], dy = t.Y[ib] - t.Y[ia];
        lt.a[k] = -dy; lt.b[k] = dx;
        lt.e[k] = (long long)dx * (oy - t.Y[ia]) - (long long)dy * (ox - t.X[ia]);
        const bool tl = dy < 0 || (dy == 0 && dx > 0);
        if (tl) lt.tl |= 1u << k;
        // largest value of the edge function over the tile's pixel centres
        const long long ex = lt.a[k] > 0 ? (long long)lt.a[k] * ((xmax - px0) * 256) : 0;
        const long long ey = lt.b[k] > 0 ? (long long)lt.b[k] * ((ymax - py0) * 256) : 0;
        const long long emax = lt.e[k] + ex + ey;
        if (emax < 0 || (emax == 0 && !tl)) keep = false;
      }
      lt.z0 = t.z0; lt.dz1 = t.dz1; lt.dz2 = t.dz2; lt.inv_area = t.inv_area; lt.id = t.id;
    }
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) s_wave_count[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t wave_base = 0, m = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wave) wave_base += s_wave_count[w]; m += s_wave_count[w]; }
    if (keep) s_tri[wave_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = lt;
    __syncthreads();

    // phase 2: every thread walks the compacted list
    for (uint32_t j = 0; j < m; ++j) {
      const LdsTri& t = s_tri[j];
      long long e[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) e[k] = t.e[k] + (long long)t.a[k] * (qx * 256) + (long long)t.b[k] * (qy * 256);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int sx = s & 1, sy = s >> 1;
        long long f[3];
        bool in = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          f[k] = e[k] + (sx ? (long long)t.a[k] * 256 : 0) + (sy ? (long long)t.b[k] * 256 : 0);
          in = in && (f[k] > 0 || (f[k] == 0 && ((t.tl >> k) & 1u)));
        }
        if (in) {
          const float l1 = mul_rn((float)f[1], t.inv_area), l2 = mul_rn((float)f[2], t.inv_area);
          const float z = add_rn(add_rn(t.z0, mul_rn(l1, t.dz1)), mul_rn(l2, t.dz2));
          if (z < zmin[s] || (z == zmin[s] && best[s] != 0u && t.id + 1u < best[s])) { zmin[s] = z; best[s] = t.id + 1u; }
        }
      }
    }
    __syncthreads();
  }

  const int rows_w = p->W;
#pragma unroll
  for (int sy = 0; sy < 2; ++sy) {
    const int y = py0 + qy + sy;
    if (y > ymax) continue;
    const int x = px0 + qx;
    const size_t o = (size_t)(y - p->y0) * rows_w + x;
    if (x + 1 <= xmax && (o & 1) == 0) {
      *reinterpret_cast<uint2*>(vis + o) = make_uint2(best[sy * 2], best[sy * 2 + 1]);
      if (depth) *reinterpret_cast<float2*>(depth + o) = make_float2(zmin[sy * 2], zmin[sy * 2 + 1]);
    } else {
      if (x <= xmax) { vis[o] = best[sy * 2]; if (depth) depth[o] = zmin[sy * 2]; }
      if (x + 1 <= xmax) { vis[o + 1] = best[sy * 2 + 1]; if (depth) depth[o + 1] = zmin[sy * 2 + 1]; }
    }
  }
}

}  // namespace

void launch_vertex_stage(hipStream_t s, const FrameParams* p, const float* vsoa, const float* inst_model, const float* inst_normal,
                         VertexOut* out, int n_verts, int n_instances) {
  dim3 grid((n_verts + 255) / 256, n_instances);
  hipLaunchKernelGGL(vertex_kernel, grid, dim3(256), 0, s, p, vsoa, inst_model, inst_normal, out, n_verts);
}

void launch_setup_and_bin(hipStream_t s, const FrameParams* p, const VertexOut* vout, const uint16_t* indices, SetupTri* setup,
                          uint32_t setup_cap, uint32_t* tile_count, uint32_t* tile_offset, uint32_t* tile_cursor, uint32_t* bin_list,
                          uint32_t bin_cap, uint32_t* counters, int n_tris, int n_instances, int n_tiles) {
  (void)hipMemsetAsync(tile_count, 0, sizeof(uint32_t) * (size_t)n_tiles, s);
  (void)hipMemsetAsync(counters, 0, sizeof(uint32_t) * 4, s);
  dim3 grid((n_tris + 255) / 256, n_instances);
  hipLaunchKernelGGL(setup_kernel, grid, dim3(256), 0, s, p, vout, indices, setup, setup_cap, tile_count, counters, n_tris);
  hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, tile_count, tile_offset, tile_cursor, counters, n_tiles);
  hipLaunchKernelGGL(bin_fill_kernel, dim3((setup_cap + 255) / 256), dim3(256), 0, s, p, setup, setup_cap, tile_offset, tile_cursor, bin_list,
                     bin_cap, counters);
}

void launch_raster(hipStream_t s, const FrameParams* p, const SetupTri* setup, const uint32_t* tile_count, const uint32_t* tile_offset,
                   const uint32_t* bin_list, uint32_t* vis, float* depth, int tiles_x, int tiles_y) {
  hipLaunchKernelGGL(raster_kernel, dim3(tiles_x, tiles_y), dim3(256), 0, s, p, setup, tile_count, tile_offset, bin_list, vis, depth);
}
