"""Orbit camera with the reference's surface: `Camera().view`, `.position`, and the tumble /
dolly / track updates of src/camera.ts:163-202.  The DOM listeners of camera.ts:50-69 have no
counterpart in a headless renderer; the update methods are public instead."""
import numpy as np

from . import mathx


class Camera:
    def __init__(self):
        # camera.ts:31-42
        self.rotate_amplitude = 0.1
        self.pan_amplitude = 0.01
        self.scroll_amplitude = 0.01
        self.pitch = 0.0
        self.yaw = 0.0
        self.distance = 5.0
        self.target = np.array([0.0, 0.0, 0.0], dtype=np.float32)
        self.eye = np.array([0.0, 0.0, 5.0], dtype=np.float32)
        self.up = np.array([0.0, 1.0, 0.0], dtype=np.float32)
        self.right = np.array([1.0, 0.0, 0.0], dtype=np.float32)

    @property
    def view(self):
        return mathx.look_at(self.eye, self.target, self.up)          # camera.ts:72-74

    @property
    def position(self):
        return self.eye                                                # camera.ts:76-78

    def update_camera_rotation(self, delta_x, delta_y):
        """camera.ts:163-183."""
        self.target = np.zeros(3, dtype=np.float32)
        self.yaw = -delta_x * self.rotate_amplitude
        self.pitch = -delta_y * self.rotate_amplitude
        quat_yaw = mathx.quat_from_axis_angle(self.up, mathx.deg_to_rad(self.yaw))
        back = mathx.vec3_normalize(mathx.vec3_sub(self.eye, self.target))
        self.right = mathx.transform_quat(mathx.vec3_cross(self.up, back), quat_yaw)
        quat_pitch = mathx.quat_from_axis_angle(self.right, mathx.deg_to_rad(self.pitch))
        rotation = mathx.quat_mul(quat_yaw, quat_pitch)
        new_back = mathx.transform_quat(back, rotation)
        self.up = mathx.vec3_normalize(mathx.vec3_cross(new_back, self.right))
        self.eye = mathx.vec3_add(self.target, mathx.vec3_mul_scalar(new_back, self.distance))

    def update_camera_distance(self, delta_x, delta_y):
        """camera.ts:185-189."""
        self.distance -= (delta_x + delta_y) * self.scroll_amplitude
        back = mathx.vec3_normalize(mathx.vec3_sub(self.eye, self.target))
        self.eye = mathx.vec3_add(self.target, mathx.vec3_mul_scalar(back, self.distance))

    def update_camera_target(self, delta_x, delta_y):
        """camera.ts:191-202."""
        pan = np.zeros(3, dtype=np.float32)
        pan = mathx.vec3_add_scaled(pan, self.right, -delta_x * self.pan_amplitude)
        pan = mathx.vec3_add_scaled(pan, self.up, delta_y * self.pan_amplitude)
        self.target = mathx.vec3_add(self.target, pan)
        self.eye = mathx.vec3_add(self.eye, pan)
