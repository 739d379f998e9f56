// Node N-API addon over the C ABI of include/pbr355.h: the layer the headless `Renderer`
// (renderer.js) uses where the reference's Renderer (src/renderer.ts) uses a GPUDevice.  Each
// export forwards to one pbr_* entry point; a non-zero status becomes `throw new Error(...)` with
// the library's message, which is the reference's own error convention (renderer.ts:140-231).
// Plain N-API C calls only (no node-addon-api); typed-array arguments are passed by pointer and
// copied by the library during the call.
#include <node_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "pbr355.h"

extern "C" int pbr_host_jpeg_decode(const uint8_t* data, size_t len, uint8_t** rgba_out, int32_t* w, int32_t* h, char* err, size_t err_len);
extern "C" void pbr_host_free(void* p);

namespace {

#define NAPI_OK_OR_RETURN(call)                                            \
  do {                                                                     \
    if ((call) != napi_ok) {                                               \
      napi_throw_error(env, nullptr, "N-API call failed: " #call);         \
      return nullptr;                                                      \
    }                                                                      \
  } while (0)

struct Args {
  napi_env env;
  size_t argc = 12;
  napi_value argv[12];
  bool ok = false;
  Args(napi_env e, napi_callback_info info, size_t need, const char* usage) : env(e) {
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok) { napi_throw_error(env, nullptr, "cannot read arguments"); return; }
    if (argc < need) { napi_throw_type_error(env, nullptr, usage); return; }
    ok = true;
  }
  bool is_nullish(size_t i) const {
    if (i >= argc) return true;
    napi_valuetype t;
    napi_typeof(env, argv[i], &t);
    return t == napi_undefined || t == napi_null;
  }
  bool ctx(size_t i, PbrContext** out) const {
    void* p = nullptr;
    if (napi_get_value_external(env, argv[i], &p) != napi_ok || !p) { napi_throw_type_error(env, nullptr, "argument is not a live renderer context"); return false; }
    *out = *static_cast<PbrContext**>(p);
    if (!*out) { napi_throw_error(env, nullptr, "renderer context was destroyed"); return false; }
    return true;
  }
  bool i32(size_t i, int32_t* out) const {
    if (napi_get_value_int32(env, argv[i], out) != napi_ok) { napi_throw_type_error(env, nullptr, "expected an integer argument"); return false; }
    return true;
  }
  bool flag(size_t i, bool* out) const {
    *out = false;
    if (is_nullish(i)) return true;
    napi_value b;
    if (napi_coerce_to_bool(env, argv[i], &b) != napi_ok || napi_get_value_bool(env, b, out) != napi_ok) { napi_throw_type_error(env, nullptr, "expected a boolean argument"); return false; }
    return true;
  }
  // a TypedArray of the given element type (or, for bytes, also an ArrayBuffer / Buffer)
  // element type of a typed-array argument (napi_int8_array when the value is not a typed array)
  napi_typedarray_type element_type(size_t i) const {
    bool is_ta = false;
    napi_is_typedarray(env, argv[i], &is_ta);
    napi_typedarray_type type = napi_int8_array; napi_value ab; size_t off, count; void* data;
    if (is_ta) napi_get_typedarray_info(env, argv[i], &type, &count, &data, &ab, &off);
    return type;
  }
  bool typed(size_t i, napi_typedarray_type want, void** data, size_t* count) const {
    bool is_ta = false;
    napi_is_typedarray(env, argv[i], &is_ta);
    if (is_ta) {
      napi_typedarray_type type; napi_value ab; size_t off;
      if (napi_get_typedarray_info(env, argv[i], &type, count, data, &ab, &off) != napi_ok) { napi_throw_type_error(env, nullptr, "bad typed array"); return false; }
      if (type != want && !(want == napi_uint8_array && type == napi_uint8_clamped_array)) { napi_throw_type_error(env, nullptr, "typed array has the wrong element type"); return false; }
      return true;
    }
    bool is_ab = false;
    napi_is_arraybuffer(env, argv[i], &is_ab);
    if (is_ab && want == napi_uint8_array) {
      if (napi_get_arraybuffer_info(env, argv[i], data, count) != napi_ok) { napi_throw_type_error(env, nullptr, "bad ArrayBuffer"); return false; }
      return true;
    }
    napi_throw_type_error(env, nullptr, "expected a typed array");
    return false;
  }
};

napi_value undefined_value(napi_env env) {
  napi_value u;
  napi_get_undefined(env, &u);
  return u;
}

// status -> exception carrying pbr_last_error
bool check(napi_env env, PbrContext* ctx, int rc) {
  if (rc == PBR_OK) return true;
  const char* msg = pbr_last_error(ctx);
  char code[16];
  snprintf(code, sizeof(code), "PBR_%d", -rc);
  napi_throw_error(env, code, (msg && *msg) ? msg : "libpbr355 call failed");
  return false;
}

void finalize_ctx(napi_env, void* data, void*) {
  PbrContext** slot = static_cast<PbrContext**>(data);
  if (*slot) pbr_destroy(*slot);
  delete slot;
}

// create(deviceOrdinal) -> context handle                      [requestAdapter/requestDevice, renderer.ts:144-198]
napi_value Create(napi_env env, napi_callback_info info) {
  Args a(env, info, 1, "create(deviceOrdinal)");
  if (!a.ok) return nullptr;
  int32_t ordinal;
  if (!a.i32(0, &ordinal)) return nullptr;
  PbrContext* ctx = nullptr;
  const int rc = pbr_create(ordinal, &ctx);
  if (rc != PBR_OK) { check(env, nullptr, rc); return nullptr; }
  PbrContext** slot = new PbrContext*(ctx);
  napi_value ext;
  if (napi_create_external(env, slot, finalize_ctx, nullptr, &ext) != napi_ok) {
    pbr_destroy(ctx);
    delete slot;
    napi_throw_error(env, nullptr, "cannot wrap the renderer context");
    return nullptr;
  }
  return ext;
}

napi_value Destroy(napi_env env, napi_callback_info info) {
  Args a(env, info, 1, "destroy(ctx)");
  if (!a.ok) return nullptr;
  void* p = nullptr;
  if (napi_get_value_external(env, a.argv[0], &p) != napi_ok || !p) { napi_throw_type_error(env, nullptr, "argument is not a renderer context"); return nullptr; }
  PbrContext** slot = static_cast<PbrContext**>(p);
  if (*slot) { pbr_destroy(*slot); *slot = nullptr; }
  return undefined_value(env);
}

napi_value Synchronize(napi_env env, napi_callback_info info) {
  Args a(env, info, 1, "synchronize(ctx)");
  PbrContext* ctx;
  if (!a.ok || !a.ctx(0, &ctx)) return nullptr;
  if (!check(env, ctx, pbr_synchronize(ctx))) return nullptr;
  return undefined_value(env);
}

// uploadMesh(ctx, Float32Array vertices, Uint16Array | Uint32Array indices[, instanceCount, Float32Array models, Float32Array normals])
//                                                               [createVertexBuffer/createIndexBuffer, renderer.ts:504-540]
napi_value UploadMesh(napi_env env, napi_callback_info info) {
  Args a(env, info, 3, "uploadMesh(ctx, vertices, indices[, instanceCount, models, normals])");
  PbrContext* ctx;
  if (!a.ok || !a.ctx(0, &ctx)) return nullptr;
  void *v, *idx, *models = nullptr, *normals = nullptr;
  size_t nv, ni, nm = 0, nn = 0;
  const bool wide = a.element_type(2) == napi_uint32_array;
  if (!a.typed(1, napi_float32_array, &v, &nv) || !a.typed(2, wide ? napi_uint32_array : napi_uint16_array, &idx, &ni)) return nullptr;
  int32_t instances = 1;
  if (!a.is_nullish(3) && !a.i32(3, &instances)) return nullptr;
  if (!a.is_nullish(4) && !a.typed(4, napi_float32_array, &models, &nm)) return nullptr;
  if (!a.is_nullish(5) && !a.typed(5, napi_float32_array, &normals, &nn)) return nullptr;
  if (nv % 8) { napi_throw_range_error(env, nullptr, "vertices must hold 8 floats per vertex"); return nullptr; }
  if (models && (instances < 1 || nm != (size_t)instances * 16 || nn != nm)) { napi_throw_range_error(env, nullptr, "instance matrices must hold 16 floats per instance"); return nullptr; }
  const int rc = wide ? pbr_upload_mesh_u32(ctx, static_cast<float*>(v), (int32_t)(nv / 8), static_cast<uint32_t*>(idx), (int32_t)ni, instances,
                                            static_cast<float*>(models), static_cast<float*>(normals))
                      : pbr_upload_mesh(ctx, static_cast<float*>(v), (int32_t)(nv / 8), static_cast<uint16_t*>(idx), (int32_t)ni, instances,
                                        static_cast<float*>(models), static_cast<float*>(normals));
  if (!check(env, ctx, rc)) return nullptr;
  return undefined_value(env);
}

// uploadTexture(ctx, slot, Uint8Array rgba, width, height, isSrgb, genMips)   [createTexture2DFromSource, renderer.ts:1039-1101]
napi_value UploadTexture(napi_env env, napi_callback_info info) {
  Args a(env, info, 7, "uploadTexture(ctx, slot, rgba, width, height, isSrgb, genMips)");
  PbrContext* ctx;
  if (!a.ok || !a.ctx(0, &ctx)) return nullptr;
  int32_t slot, w, h;
  void* px; size_t n;
  bool srgb, mips;
  if (!a.i32(1, &slot) || !a.typed(2, napi_uint8_array, &px, &n) || !a.i32(3, &w) || !a.i32(4, &h) || !a.flag(5, &srgb) || !a.flag(6, &mips)) return nullptr;
  if (w <= 0 || h <= 0 || n != (size_t)w * h * 4) { napi_throw_range_error(env, nullptr, "rgba must hold width*height*4 bytes"); return nullptr; }
  if (!check(env, ctx, pbr_upload_texture(ctx, slot, static_cast<uint8_t*>(px), w, h, srgb ? 1 : 0, mips ? 1 : 0))) return nullptr;
  return undefined_value(env);
}

// uploadHdr(ctx, Float32Array rgba32f, width, height)          [generateCubemap's writeTexture, renderer.ts:1449, :1520]
napi_value UploadHdr(napi_env env, napi_callback_info info) {
  Args a(env, info, 4, "uploadHdr(ctx, data, width, height)");
  PbrContext* ctx;
  if (!a.ok || !a.ctx(0, &ctx)) return nullptr;
  void* px; size_t n; int32_t w, h;
  if (!a.typed(1, napi_float32_array, &px, &n) || !a.i32(2, &w) || !a.i32(3, &h)) return nullptr;
  if (w <= 0 || h <= 0 || n != (size_t)w * h * 4) { napi_throw_range_error(env, nullptr, "data must hold width*height*4 floats"); return nullptr; }
  if (!check(env, ctx, pbr_upload_hdr(ctx, static_cast<float*>(px), w, h))) return nullptr;
  return undefined_value(env);
}

template <int (*Fn)(PbrContext*, int32_t)>
napi_value IblSized(napi_env env, napi_callback_info info) {
  Args a(env, info, 2, "(ctx, size)");
  PbrContext* ctx; int32_t size;
  if (!a.ok || !a.ctx(0, &ctx) || !a.i32(1, &size)) return nullptr;
  if (!check(env, ctx, Fn(ctx, size))) return nullptr;
  return undefined_value(env);
}

// iblPrefilter(ctx, size, mips, samples[, shardIndex, shardCount])   [generateCubemap(PREFILTER), renderer.ts:1537-1616]
napi_value IblPrefilter(napi_env env, napi_callback_info info) {
  Args a(env, info, 4, "iblPrefilter(ctx, size, mips, samples[, shardIndex, shardCount])");
  PbrContext* ctx; int32_t size, mips, samples, shard = 0, count = 1;
  if (!a.ok || !a.ctx(0, &ctx) || !a.i32(1, &size) || !a.i32(2, &mips) || !a.i32(3, &samples)) return nullptr;
  if (!a.is_nullish(4) && !a.i32(4, &shard)) return nullptr;
  if (!a.is_nullish(5) && !a.i32(5, &count)) return nullptr;
  if (!check(env, ctx, pbr_ibl_prefilter(ctx, size, mips, samples, shard, count))) return nullptr;
  return undefined_value(env);
}

// iblDownload(ctx, which, level) -> { width, height, layers, data }  (Float32Array, or Uint16Array of f16 bits for the LUT)
napi_value IblDownload(napi_env env, napi_callback_info info) {
  Args a(env, info, 3, "iblDownload(ctx, which, level)");
  PbrContext* ctx; int32_t which, level;
  if (!a.ok || !a.ctx(0, &ctx) || !a.i32(1, &which) || !a.i32(2, &level)) return nullptr;
  int32_t w, h, layers; size_t bytes;
  if (!check(env, ctx, pbr_ibl_level_info(ctx, which, level, &w, &h, &layers, &bytes))) return nullptr;
  void* mem = nullptr;
  napi_value ab, arr, obj, v;
  NAPI_OK_OR_RETURN(napi_create_arraybuffer(env, bytes, &mem, &ab));
  if (!check(env, ctx, pbr_ibl_download(ctx, which, level, mem, bytes))) return nullptr;
  if (which == PBR_MAP_BRDF_LUT) NAPI_OK_OR_RETURN(napi_create_typedarray(env, napi_uint16_array, bytes / 2, ab, 0, &arr));
  else NAPI_OK_OR_RETURN(napi_create_typedarray(env, napi_float32_array, bytes / 4, ab, 0, &arr));
  NAPI_OK_OR_RETURN(napi_create_object(env, &obj));
  napi_create_int32(env, w, &v); napi_set_named_property(env, obj, "width", v);
  napi_create_int32(env, h, &v); napi_set_named_property(env, obj, "height", v);
  napi_create_int32(env, layers, &v); napi_set_named_property(env, obj, "layers", v);
  napi_set_named_property(env, obj, "data", arr);
  return obj;
}

// drawFrame(ctx, uniforms(ArrayBuffer|Uint8Array of sizeof(PbrFrameUniforms)), width, height, y0, y1, Uint8Array rgba[, Uint32Array vis, wantStats])
//   -> undefined, or the PbrFrameStats fields when wantStats is set        [drawFrame's render pass, renderer.ts:1998-2013]
napi_value DrawFrame(napi_env env, napi_callback_info info) {
  Args a(env, info, 7, "drawFrame(ctx, uniforms, width, height, y0, y1, rgba[, vis, wantStats])");
  PbrContext* ctx;
  if (!a.ok || !a.ctx(0, &ctx)) return nullptr;
  void *u, *out, *vis = nullptr; size_t ulen, olen, vlen = 0;
  int32_t w, h, y0, y1; bool want_stats;
  if (!a.typed(1, napi_uint8_array, &u, &ulen) || !a.i32(2, &w) || !a.i32(3, &h) || !a.i32(4, &y0) || !a.i32(5, &y1) ||
      !a.typed(6, napi_uint8_array, &out, &olen))
    return nullptr;
  if (!a.is_nullish(7) && !a.typed(7, napi_uint32_array, &vis, &vlen)) return nullptr;
  if (!a.flag(8, &want_stats)) return nullptr;
  if (ulen != sizeof(PbrFrameUniforms)) { napi_throw_range_error(env, nullptr, "uniforms must be sizeof(PbrFrameUniforms) = 496 bytes"); return nullptr; }
  if (w <= 0 || h <= 0 || y0 < 0 || y1 <= y0 || y1 > h) { napi_throw_range_error(env, nullptr, "bad frame size or row range"); return nullptr; }
  const size_t px = (size_t)w * (size_t)(y1 - y0);
  PbrFrameUniforms uniforms;
  std::memcpy(&uniforms, u, sizeof(uniforms));
  const size_t samples = (uniforms.sampler_index & 16u) ? 4 : 1;   // the multisampled pipeline reports one id per sample
  if (olen != px * 4 || (vis && vlen != px * samples)) { napi_throw_range_error(env, nullptr, "output arrays do not match the row range"); return nullptr; }
  PbrFrameStats st;
  if (!check(env, ctx, pbr_draw_frame(ctx, &uniforms, w, h, y0, y1, static_cast<uint8_t*>(out), static_cast<uint32_t*>(vis), want_stats ? &st : nullptr)))
    return nullptr;
  if (!want_stats) return undefined_value(env);
  napi_value obj, v;
  NAPI_OK_OR_RETURN(napi_create_object(env, &obj));
  const struct { const char* name; double value; } fields[] = {
      {"msVertex", st.ms_vertex}, {"msSetup", st.ms_setup}, {"msRaster", st.ms_raster}, {"msResolve", st.ms_resolve}, {"msTotal", st.ms_total},
      {"coveredPixels", (double)st.covered_pixels}, {"spilledRefs", (double)st.binned_refs}, {"setupTriangles", (double)st.setup_triangles},
      {"uniformQuads", (double)st.uniform_quads}, {"mixedQuads", (double)st.mixed_quads}};
  for (const auto& f : fields) { napi_create_double(env, f.value, &v); napi_set_named_property(env, obj, f.name, v); }
  return obj;
}

// decodeJpeg(Uint8Array|Buffer) -> { width, height, data: Uint8Array rgba }   [createImageBitmap, utils.ts:13-18]
napi_value DecodeJpeg(napi_env env, napi_callback_info info) {
  Args a(env, info, 1, "decodeJpeg(bytes)");
  if (!a.ok) return nullptr;
  void* src; size_t n;
  if (!a.typed(0, napi_uint8_array, &src, &n)) return nullptr;
  uint8_t* rgba = nullptr; int32_t w = 0, h = 0; char err[200];
  if (pbr_host_jpeg_decode(static_cast<uint8_t*>(src), n, &rgba, &w, &h, err, sizeof(err)) != 0) { napi_throw_error(env, nullptr, err); return nullptr; }
  void* mem = nullptr;
  napi_value ab, arr, obj, v;
  const size_t bytes = (size_t)w * h * 4;
  if (napi_create_arraybuffer(env, bytes, &mem, &ab) != napi_ok) { pbr_host_free(rgba); napi_throw_error(env, nullptr, "cannot allocate the image"); return nullptr; }
  std::memcpy(mem, rgba, bytes);
  pbr_host_free(rgba);
  NAPI_OK_OR_RETURN(napi_create_typedarray(env, napi_uint8_array, bytes, ab, 0, &arr));
  NAPI_OK_OR_RETURN(napi_create_object(env, &obj));
  napi_create_int32(env, w, &v); napi_set_named_property(env, obj, "width", v);
  napi_create_int32(env, h, &v); napi_set_named_property(env, obj, "height", v);
  napi_set_named_property(env, obj, "data", arr);
  return obj;
}

napi_value Init(napi_env env, napi_value exports) {
  const struct { const char* name; napi_callback fn; } fns[] = {
      {"create", Create}, {"destroy", Destroy}, {"synchronize", Synchronize}, {"uploadMesh", UploadMesh}, {"uploadTexture", UploadTexture},
      {"uploadHdr", UploadHdr}, {"iblEquirectToCube", IblSized<pbr_ibl_equirect_to_cube>}, {"iblIrradiance", IblSized<pbr_ibl_irradiance>},
      {"iblBrdfLut", IblSized<pbr_ibl_brdf_lut>}, {"iblPrefilter", IblPrefilter}, {"iblDownload", IblDownload}, {"drawFrame", DrawFrame},
      {"decodeJpeg", DecodeJpeg}};
  for (const auto& f : fns) {
    napi_value fn;
    if (napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &fn) != napi_ok || napi_set_named_property(env, exports, f.name, fn) != napi_ok) {
      napi_throw_error(env, nullptr, "cannot register addon functions");
      return nullptr;
    }
  }
  napi_value v;
  napi_create_uint32(env, (uint32_t)sizeof(PbrFrameUniforms), &v);
  napi_set_named_property(env, exports, "UNIFORM_BYTES", v);
  const struct { const char* name; int value; } consts[] = {
      {"TEX_ALBEDO", PBR_TEX_ALBEDO}, {"TEX_METALLIC_ROUGHNESS", PBR_TEX_METALLIC_ROUGHNESS}, {"TEX_NORMAL", PBR_TEX_NORMAL},
      {"TEX_EMISSIVE", PBR_TEX_EMISSIVE}, {"TEX_OCCLUSION", PBR_TEX_OCCLUSION}, {"MAP_SKYBOX", PBR_MAP_SKYBOX}, {"MAP_IRRADIANCE", PBR_MAP_IRRADIANCE},
      {"MAP_PREFILTER", PBR_MAP_PREFILTER}, {"MAP_BRDF_LUT", PBR_MAP_BRDF_LUT}, {"MAP_HDR", PBR_MAP_HDR}};
  for (const auto& c : consts) { napi_create_int32(env, c.value, &v); napi_set_named_property(env, exports, c.name, v); }
  return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
