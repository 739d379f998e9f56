// Oracle (test infrastructure): restatement of the wgpu-matrix 3.3.0 operations the reference
// calls (package.json:29, package-lock.json:4905-4908; the library source is not vendored in the
// reference) and of the orbit camera in src/camera.ts.  Conventions: column-major Float32Array,
// perspective maps depth to [0,1], lookAt returns a view matrix, scale/translate post-multiply.
// JavaScript evaluates in f64 and rounds on the store into a Float32Array; the same is done here.
// Parity unpinned: no reference test holds a matrix value.
#include "oracle.h"
#include <cmath>
#include <cstring>

namespace {
inline float f(double x) { return (float)x; }

void v3_normalize(const float* v, float* out) {
  double x = v[0], y = v[1], z = v[2];
  double len = std::sqrt(x * x + y * y + z * z);
  if (len > 0.00001) { out[0] = f(x / len); out[1] = f(y / len); out[2] = f(z / len); }
  else { out[0] = out[1] = out[2] = 0.f; }
}
void v3_cross(const float* a, const float* b, float* out) {
  double ax = a[0], ay = a[1], az = a[2], bx = b[0], by = b[1], bz = b[2];
  float t0 = f(ay * bz - az * by), t1 = f(az * bx - ax * bz), t2 = f(ax * by - ay * bx);
  out[0] = t0; out[1] = t1; out[2] = t2;
}
void v3_sub(const float* a, const float* b, float* out) {
  for (int i = 0; i < 3; ++i) out[i] = f((double)a[i] - (double)b[i]);
}
void quat_axis_angle(const float* axis, double angle, float* q) {
  double h = angle * 0.5, s = std::sin(h);
  q[0] = f(s * axis[0]); q[1] = f(s * axis[1]); q[2] = f(s * axis[2]); q[3] = f(std::cos(h));
}
void quat_mul(const float* a, const float* b, float* out) {
  double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
  out[0] = f(ax * bw + aw * bx + ay * bz - az * by);
  out[1] = f(ay * bw + aw * by + az * bx - ax * bz);
  out[2] = f(az * bw + aw * bz + ax * by - ay * bx);
  out[3] = f(aw * bw - ax * bx - ay * by - az * bz);
}
void v3_transform_quat(const float* v, const float* q, float* out) {
  double qx = q[0], qy = q[1], qz = q[2], w2 = (double)q[3] * 2;
  double x = v[0], y = v[1], z = v[2];
  double uvX = qy * z - qz * y, uvY = qz * x - qx * z, uvZ = qx * y - qy * x;
  out[0] = f(x + uvX * w2 + (qy * uvZ - qz * uvY) * 2);
  out[1] = f(y + uvY * w2 + (qz * uvX - qx * uvZ) * 2);
  out[2] = f(z + uvZ * w2 + (qx * uvY - qy * uvX) * 2);
}
double deg2rad(double d) { return d * M_PI / 180.0; }
}  // namespace

extern "C" {

void orc_mat4_perspective(float fovy, float aspect, float z_near, float z_far, float* o) {
  std::memset(o, 0, 16 * sizeof(float));
  double fl = std::tan(M_PI * 0.5 - 0.5 * (double)fovy);
  o[0] = f(fl / aspect);
  o[5] = f(fl);
  o[11] = -1.f;
  if (std::isfinite(z_far)) {
    double range_inv = 1.0 / ((double)z_near - (double)z_far);
    o[10] = f(z_far * range_inv);
    o[14] = f((double)z_far * z_near * range_inv);
  } else {
    o[10] = -1.f;
    o[14] = -z_near;
  }
}

void orc_mat4_look_at(const float* eye, const float* target, const float* up, float* o) {
  float x[3], y[3], z[3], t[3];
  v3_sub(eye, target, t); v3_normalize(t, z);
  v3_cross(up, z, t);     v3_normalize(t, x);
  v3_cross(z, x, t);      v3_normalize(t, y);
  o[0] = x[0]; o[1] = y[0]; o[2] = z[0]; o[3] = 0;
  o[4] = x[1]; o[5] = y[1]; o[6] = z[1]; o[7] = 0;
  o[8] = x[2]; o[9] = y[2]; o[10] = z[2]; o[11] = 0;
  o[12] = f(-((double)x[0] * eye[0] + (double)x[1] * eye[1] + (double)x[2] * eye[2]));
  o[13] = f(-((double)y[0] * eye[0] + (double)y[1] * eye[1] + (double)y[2] * eye[2]));
  o[14] = f(-((double)z[0] * eye[0] + (double)z[1] * eye[1] + (double)z[2] * eye[2]));
  o[15] = 1;
}

void orc_mat4_mul(const float* a, const float* b, float* o) {
  float r[16];
  for (int c = 0; c < 4; ++c)
    for (int row = 0; row < 4; ++row) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += (double)a[k * 4 + row] * (double)b[c * 4 + k];
      r[c * 4 + row] = f(s);
    }
  std::memcpy(o, r, sizeof r);
}

int orc_mat4_inverse(const float* m, float* o) {
  double a[16], inv[16];
  for (int i = 0; i < 16; ++i) a[i] = m[i];
  inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
  inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
  inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
  inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
  inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
  inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
  inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
  inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
  inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
  inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
  inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
  inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
  inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
  inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
  inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
  inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
  double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
  if (det == 0) { std::memset(o, 0, 16 * sizeof(float)); return -1; }
  double d = 1.0 / det;
  for (int i = 0; i < 16; ++i) o[i] = f(inv[i] * d);
  return 0;
}

void orc_mat4_transpose(const float* m, float* o) {
  float r[16];
  for (int c = 0; c < 4; ++c) for (int row = 0; row < 4; ++row) r[c * 4 + row] = m[row * 4 + c];
  std::memcpy(o, r, sizeof r);
}

void orc_mat4_from_quat(const float* q, float* o) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double x2 = x + x, y2 = y + y, z2 = z + z;
  double xx = x * x2, yx = y * x2, yy = y * y2, zx = z * x2, zy = z * y2, zz = z * z2, wx = w * x2, wy = w * y2, wz = w * z2;
  o[0] = f(1 - yy - zz); o[1] = f(yx + wz);     o[2] = f(zx - wy);      o[3] = 0;
  o[4] = f(yx - wz);     o[5] = f(1 - xx - zz); o[6] = f(zy + wx);      o[7] = 0;
  o[8] = f(zx + wy);     o[9] = f(zy - wx);     o[10] = f(1 - xx - yy); o[11] = 0;
  o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 1;
}

void orc_mat4_scale(const float* m, const float* v, float* o) {
  float r[16];
  for (int c = 0; c < 3; ++c) for (int row = 0; row < 4; ++row) r[c * 4 + row] = f((double)v[c] * m[c * 4 + row]);
  for (int row = 0; row < 4; ++row) r[12 + row] = m[12 + row];
  std::memcpy(o, r, sizeof r);
}

void orc_mat4_translate(const float* m, const float* v, float* o) {
  float r[16];
  std::memcpy(r, m, sizeof r);
  for (int row = 0; row < 4; ++row)
    r[12 + row] = f((double)m[row] * v[0] + (double)m[4 + row] * v[1] + (double)m[8 + row] * v[2] + (double)m[12 + row]);
  std::memcpy(o, r, sizeof r);
}

void orc_node_matrix(const float* scale3, const float* translation3, const float* rotation4,
                     const float* matrix16, float* out16) {
  if (matrix16) { std::memcpy(out16, matrix16, 16 * sizeof(float)); return; }   // glTF.ts:113-114
  float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (scale3) orc_mat4_scale(m, scale3, m);                                     // glTF.ts:118-120
  if (translation3) orc_mat4_translate(m, translation3, m);                     // glTF.ts:122-124
  if (rotation4) { float r[16]; orc_mat4_from_quat(rotation4, r); orc_mat4_mul(m, r, m); }   // glTF.ts:126-129
  std::memcpy(out16, m, 16 * sizeof(float));
}

void orc_camera_init(OrcCamera* c) {
  c->eye[0] = 0; c->eye[1] = 0; c->eye[2] = 5;
  c->target[0] = c->target[1] = c->target[2] = 0;
  c->up[0] = 0; c->up[1] = 1; c->up[2] = 0;
  c->right[0] = 1; c->right[1] = 0; c->right[2] = 0;
  c->distance = 5.0f;
}

void orc_camera_rotate(OrcCamera* c, float dx, float dy) {
  const double amp = 0.1;
  c->target[0] = c->target[1] = c->target[2] = 0;
  double yaw = -(double)dx * amp, pitch = -(double)dy * amp;
  float q_yaw[4], q_pitch[4], rot[4], back[3], t[3], nb[3];
  quat_axis_angle(c->up, deg2rad(yaw), q_yaw);
  v3_sub(c->eye, c->target, t); v3_normalize(t, back);
  v3_cross(c->up, back, t); v3_transform_quat(t, q_yaw, c->right);
  quat_axis_angle(c->right, deg2rad(pitch), q_pitch);
  quat_mul(q_yaw, q_pitch, rot);
  v3_transform_quat(back, rot, nb);
  v3_cross(nb, c->right, t); v3_normalize(t, c->up);
  for (int i = 0; i < 3; ++i) c->eye[i] = f((double)c->target[i] + f((double)nb[i] * (double)c->distance));
}

void orc_camera_dolly(OrcCamera* c, float dx, float dy) {
  c->distance = f((double)c->distance - ((double)dx + (double)dy) * 0.01);   // JS number; kept f32 here
  float t[3], back[3];
  v3_sub(c->eye, c->target, t); v3_normalize(t, back);
  for (int i = 0; i < 3; ++i) c->eye[i] = f((double)c->target[i] + f((double)back[i] * (double)c->distance));
}

void orc_camera_track(OrcCamera* c, float dx, float dy) {
  float pan[3] = {0, 0, 0};
  for (int i = 0; i < 3; ++i) pan[i] = f((double)pan[i] + (double)c->right[i] * (-(double)dx * 0.01));
  for (int i = 0; i < 3; ++i) pan[i] = f((double)pan[i] + (double)c->up[i] * ((double)dy * 0.01));
  for (int i = 0; i < 3; ++i) { c->target[i] = f((double)c->target[i] + pan[i]); c->eye[i] = f((double)c->eye[i] + pan[i]); }
}

void orc_camera_view(const OrcCamera* c, float* out16) { orc_mat4_look_at(c->eye, c->target, c->up, out16); }

}  // extern "C"
