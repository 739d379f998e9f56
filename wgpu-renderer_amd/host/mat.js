'use strict';
// Column-major f32 matrix / quaternion / vector helpers with the conventions of wgpu-matrix 3.x,
// which the reference imports (package.json:29) but does not ship.  Call sites served:
// renderer.ts:1837, :1846-1851, :1867; glTF.ts:114-128; camera.ts:39-42, :73, :164-201.
// Arithmetic runs in doubles and is rounded once when stored into a Float32Array, as in that library.

const utils = {
  degToRad(degrees) { return degrees * Math.PI / 180; },
};

const vec3 = {
  create(x, y, z) { return new Float32Array([x || 0, y || 0, z || 0]); },
  fromValues(x, y, z) { return new Float32Array([x, y, z]); },
  add(a, b) { return new Float32Array([a[0] + b[0], a[1] + b[1], a[2] + b[2]]); },
  sub(a, b) { return new Float32Array([a[0] - b[0], a[1] - b[1], a[2] - b[2]]); },
  mulScalar(a, k) { return new Float32Array([a[0] * k, a[1] * k, a[2] * k]); },
  addScaled(a, b, k) { return new Float32Array([a[0] + b[0] * k, a[1] + b[1] * k, a[2] + b[2] * k]); },
  dot(a, b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; },
  cross(a, b) {
    return new Float32Array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]);
  },
  normalize(v) {
    const length = Math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (length > 0.00001) return new Float32Array([v[0] / length, v[1] / length, v[2] / length]);
    return new Float32Array(3);
  },
  transformQuat(v, q) {
    const qx = q[0], qy = q[1], qz = q[2], w2 = q[3] * 2;
    const x = v[0], y = v[1], z = v[2];
    const uvX = qy * z - qz * y, uvY = qz * x - qx * z, uvZ = qx * y - qy * x;
    return new Float32Array([
      x + uvX * w2 + (qy * uvZ - qz * uvY) * 2,
      y + uvY * w2 + (qz * uvX - qx * uvZ) * 2,
      z + uvZ * w2 + (qx * uvY - qy * uvX) * 2]);
  },
};

const quat = {
  fromAxisAngle(axis, angle) {
    const half = angle * 0.5, s = Math.sin(half);
    return new Float32Array([s * axis[0], s * axis[1], s * axis[2], Math.cos(half)]);
  },
  mul(a, b) {
    const ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
    return new Float32Array([
      ax * bw + aw * bx + ay * bz - az * by,
      ay * bw + aw * by + az * bx - ax * bz,
      az * bw + aw * bz + ax * by - ay * bx,
      aw * bw - ax * bx - ay * by - az * bz]);
  },
};

const mat4 = {
  identity() { return new Float32Array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]); },
  create(...values) { const m = new Float32Array(16); m.set(values.slice(0, 16)); return m; },
  // depth mapped to [0, 1] (WebGPU clip space)
  perspective(fovy, aspect, zNear, zFar) {
    const m = new Float32Array(16);
    const f = Math.tan(Math.PI * 0.5 - 0.5 * fovy);
    m[0] = f / aspect;
    m[5] = f;
    m[11] = -1;
    if (Number.isFinite(zFar)) {
      const rangeInv = 1 / (zNear - zFar);
      m[10] = zFar * rangeInv;
      m[14] = zFar * zNear * rangeInv;
    } else {
      m[10] = -1;
      m[14] = -zNear;
    }
    return m;
  },
  // a view matrix looking from eye towards target
  lookAt(eye, target, up) {
    const z = vec3.normalize(vec3.sub(eye, target));
    const x = vec3.normalize(vec3.cross(up, z));
    const y = vec3.normalize(vec3.cross(z, x));
    return new Float32Array([x[0], y[0], z[0], 0, x[1], y[1], z[1], 0, x[2], y[2], z[2], 0,
      -vec3.dot(x, eye), -vec3.dot(y, eye), -vec3.dot(z, eye), 1]);
  },
  mul(a, b) {
    const m = new Float32Array(16);
    for (let c = 0; c < 4; ++c) {
      for (let r = 0; r < 4; ++r) {
        m[4 * c + r] = a[r] * b[4 * c] + a[4 + r] * b[4 * c + 1] + a[8 + r] * b[4 * c + 2] + a[12 + r] * b[4 * c + 3];
      }
    }
    return m;
  },
  transpose(a) {
    const m = new Float32Array(16);
    for (let c = 0; c < 4; ++c) for (let r = 0; r < 4; ++r) m[4 * c + r] = a[4 * r + c];
    return m;
  },
  // Laplace expansion over 2x2 minors of the top and bottom row pairs
  inverse(a) {
    const a00 = a[0], a10 = a[1], a20 = a[2], a30 = a[3], a01 = a[4], a11 = a[5], a21 = a[6], a31 = a[7];
    const a02 = a[8], a12 = a[9], a22 = a[10], a32 = a[11], a03 = a[12], a13 = a[13], a23 = a[14], a33 = a[15];
    const s0 = a00 * a11 - a10 * a01, s1 = a00 * a12 - a10 * a02, s2 = a00 * a13 - a10 * a03;
    const s3 = a01 * a12 - a11 * a02, s4 = a01 * a13 - a11 * a03, s5 = a02 * a13 - a12 * a03;
    const c5 = a22 * a33 - a32 * a23, c4 = a21 * a33 - a31 * a23, c3 = a21 * a32 - a31 * a22;
    const c2 = a20 * a33 - a30 * a23, c1 = a20 * a32 - a30 * a22, c0 = a20 * a31 - a30 * a21;
    const d = 1 / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0);
    return new Float32Array([
      (a11 * c5 - a12 * c4 + a13 * c3) * d, (-a10 * c5 + a12 * c2 - a13 * c1) * d, (a10 * c4 - a11 * c2 + a13 * c0) * d, (-a10 * c3 + a11 * c1 - a12 * c0) * d,
      (-a01 * c5 + a02 * c4 - a03 * c3) * d, (a00 * c5 - a02 * c2 + a03 * c1) * d, (-a00 * c4 + a01 * c2 - a03 * c0) * d, (a00 * c3 - a01 * c1 + a02 * c0) * d,
      (a31 * s5 - a32 * s4 + a33 * s3) * d, (-a30 * s5 + a32 * s2 - a33 * s1) * d, (a30 * s4 - a31 * s2 + a33 * s0) * d, (-a30 * s3 + a31 * s1 - a32 * s0) * d,
      (-a21 * s5 + a22 * s4 - a23 * s3) * d, (a20 * s5 - a22 * s2 + a23 * s1) * d, (-a20 * s4 + a21 * s2 - a23 * s0) * d, (a20 * s3 - a21 * s1 + a22 * s0) * d]);
  },
  fromQuat(q) {
    const x = q[0], y = q[1], z = q[2], w = q[3];
    const x2 = x + x, y2 = y + y, z2 = z + z;
    const xx = x * x2, yx = y * x2, yy = y * y2, zx = z * x2, zy = z * y2, zz = z * z2, wx = w * x2, wy = w * y2, wz = w * z2;
    return new Float32Array([1 - yy - zz, yx + wz, zx - wy, 0, yx - wz, 1 - xx - zz, zy + wx, 0, zx + wy, zy - wx, 1 - xx - yy, 0, 0, 0, 0, 1]);
  },
  // post-multiplying forms: m * S(v), m * T(v)
  scale(m, v) {
    const out = new Float32Array(m);
    for (let c = 0; c < 3; ++c) for (let r = 0; r < 4; ++r) out[4 * c + r] = m[4 * c + r] * v[c];
    return out;
  },
  translate(m, v) {
    const out = new Float32Array(m);
    for (let r = 0; r < 4; ++r) out[12 + r] = m[r] * v[0] + m[4 + r] * v[1] + m[8 + r] * v[2] + m[12 + r];
    return out;
  },
};
mat4.multiply = mat4.mul;
mat4.invert = mat4.inverse;
quat.multiply = quat.mul;

module.exports = { mat4, quat, vec3, utils };
