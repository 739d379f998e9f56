"""Column-major f32 matrix / quaternion helpers with the conventions of wgpu-matrix 3.3.0.

The reference imports `mat4`, `quat`, `vec3` and `utils` from wgpu-matrix (package.json:29); the
library itself is not part of the reference checkout.  JavaScript evaluates each expression in
f64 and rounds when the result is stored into a Float32Array; the functions here do the same
(Python floats are f64, `_f32` rounds).  Call sites they serve: renderer.ts:1837, :1846-1851,
:1867; glTF.ts:114-128; camera.ts:73, :164-201.
"""
import math

import numpy as np


def _f32(values):
    return np.asarray(values, dtype=np.float64).astype(np.float32)


def deg_to_rad(d):
    return d * math.pi / 180.0


def identity():
    return np.eye(4, dtype=np.float32).reshape(16)


def perspective(fovy, aspect, z_near, z_far):
    """mat4.perspective: depth mapped to [0, 1] (WebGPU clip space)."""
    f = math.tan(math.pi * 0.5 - 0.5 * fovy)
    m = [0.0] * 16
    m[0] = f / aspect
    m[5] = f
    m[11] = -1.0
    if math.isfinite(z_far):
        range_inv = 1.0 / (z_near - z_far)
        m[10] = z_far * range_inv
        m[14] = z_far * z_near * range_inv
    else:
        m[10] = -1.0
        m[14] = -z_near
    return _f32(m)


def _normalize3(v):
    x, y, z = (float(c) for c in v)
    length = math.sqrt(x * x + y * y + z * z)
    if length > 0.00001:
        return _f32([x / length, y / length, z / length])
    return _f32([0.0, 0.0, 0.0])


def _cross3(a, b):
    ax, ay, az = (float(c) for c in a)
    bx, by, bz = (float(c) for c in b)
    return _f32([ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx])


def _sub3(a, b):
    return _f32([float(a[i]) - float(b[i]) for i in range(3)])


def _dot3(a, b):
    return sum(float(a[i]) * float(b[i]) for i in range(3))


def look_at(eye, target, up):
    """mat4.lookAt: a view matrix looking from eye to target."""
    z = _normalize3(_sub3(eye, target))
    x = _normalize3(_cross3(up, z))
    y = _normalize3(_cross3(z, x))
    return _f32([x[0], y[0], z[0], 0, x[1], y[1], z[1], 0, x[2], y[2], z[2], 0,
                 -_dot3(x, eye), -_dot3(y, eye), -_dot3(z, eye), 1])


def mul(a, b):
    """mat4.mul: a . b, each element summed left to right in f64."""
    a = [float(x) for x in a]
    b = [float(x) for x in b]
    out = [0.0] * 16
    for c in range(4):
        for r in range(4):
            out[4 * c + r] = a[r] * b[4 * c] + a[4 + r] * b[4 * c + 1] + a[8 + r] * b[4 * c + 2] + a[12 + r] * b[4 * c + 3]
    return _f32(out)


def inverse(m):
    """mat4.inverse by Laplace expansion over the 2x2 minors of the top and bottom row pairs (f64)."""
    (a00, a10, a20, a30, a01, a11, a21, a31, a02, a12, a22, a32, a03, a13, a23, a33) = (float(x) for x in m)
    s0 = a00 * a11 - a10 * a01; s1 = a00 * a12 - a10 * a02; s2 = a00 * a13 - a10 * a03
    s3 = a01 * a12 - a11 * a02; s4 = a01 * a13 - a11 * a03; s5 = a02 * a13 - a12 * a03
    c5 = a22 * a33 - a32 * a23; c4 = a21 * a33 - a31 * a23; c3 = a21 * a32 - a31 * a22
    c2 = a20 * a33 - a30 * a23; c1 = a20 * a32 - a30 * a22; c0 = a20 * a31 - a30 * a21
    d = 1 / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0)
    return _f32([
        (a11 * c5 - a12 * c4 + a13 * c3) * d, (-a10 * c5 + a12 * c2 - a13 * c1) * d, (a10 * c4 - a11 * c2 + a13 * c0) * d, (-a10 * c3 + a11 * c1 - a12 * c0) * d,
        (-a01 * c5 + a02 * c4 - a03 * c3) * d, (a00 * c5 - a02 * c2 + a03 * c1) * d, (-a00 * c4 + a01 * c2 - a03 * c0) * d, (a00 * c3 - a01 * c1 + a02 * c0) * d,
        (a31 * s5 - a32 * s4 + a33 * s3) * d, (-a30 * s5 + a32 * s2 - a33 * s1) * d, (a30 * s4 - a31 * s2 + a33 * s0) * d, (-a30 * s3 + a31 * s1 - a32 * s0) * d,
        (-a21 * s5 + a22 * s4 - a23 * s3) * d, (a20 * s5 - a22 * s2 + a23 * s1) * d, (-a20 * s4 + a21 * s2 - a23 * s0) * d, (a20 * s3 - a21 * s1 + a22 * s0) * d])


def transpose(m):
    return np.asarray(m, dtype=np.float32).reshape(4, 4).T.copy().reshape(16)


def from_quat(q):
    x, y, z, w = (float(c) for c in q)
    x2, y2, z2 = x + x, y + y, z + z
    xx, yx, yy = x * x2, y * x2, y * y2
    zx, zy, zz = z * x2, z * y2, z * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    return _f32([1 - yy - zz, yx + wz, zx - wy, 0, yx - wz, 1 - xx - zz, zy + wx, 0,
                 zx + wy, zy - wx, 1 - xx - yy, 0, 0, 0, 0, 1])


def scale(m, v):
    m = np.asarray(m, dtype=np.float64)
    out = m.copy()
    for c in range(3):
        out[4 * c:4 * c + 4] = m[4 * c:4 * c + 4] * float(v[c])
    return out.astype(np.float32)


def translate(m, v):
    m = np.asarray(m, dtype=np.float64)
    out = m.copy()
    out[12:16] = m[0:4] * float(v[0]) + m[4:8] * float(v[1]) + m[8:12] * float(v[2]) + m[12:16]
    return out.astype(np.float32)


def quat_from_axis_angle(axis, angle):
    half = angle * 0.5
    s = math.sin(half)
    return _f32([s * float(axis[0]), s * float(axis[1]), s * float(axis[2]), math.cos(half)])


def quat_mul(a, b):
    ax, ay, az, aw = (float(c) for c in a)
    bx, by, bz, bw = (float(c) for c in b)
    return _f32([ax * bw + aw * bx + ay * bz - az * by, ay * bw + aw * by + az * bx - ax * bz,
                 az * bw + aw * bz + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def transform_quat(v, q):
    qx, qy, qz = float(q[0]), float(q[1]), float(q[2])
    w2 = float(q[3]) * 2
    x, y, z = (float(c) for c in v)
    uvx, uvy, uvz = qy * z - qz * y, qz * x - qx * z, qx * y - qy * x
    return _f32([x + uvx * w2 + (qy * uvz - qz * uvy) * 2, y + uvy * w2 + (qz * uvx - qx * uvz) * 2,
                 z + uvz * w2 + (qx * uvy - qy * uvx) * 2])


vec3_normalize = _normalize3
vec3_cross = _cross3
vec3_sub = _sub3


def vec3_add(a, b):
    return _f32([float(a[i]) + float(b[i]) for i in range(3)])


def vec3_mul_scalar(a, s):
    return _f32([float(a[i]) * float(s) for i in range(3)])


def vec3_add_scaled(a, b, s):
    return _f32([float(a[i]) + float(b[i]) * float(s) for i in range(3)])
