'use strict';
// Orbit camera with the reference's surface (src/camera.ts): `new Camera()`, `.view`, `.position`.
// The DOM listeners of camera.ts:50-69 have no counterpart in a headless renderer, so the three
// update methods they drive (camera.ts:163-202) are public here.
const { mat4, quat, vec3, utils } = require('./mat.js');

class Camera {
  constructor() {
    // camera.ts:31-42
    this.rotateAmplitude = 0.1;
    this.panAmplitude = 0.01;
    this.scrollAmplitude = 0.01;
    this.pitch = 0;
    this.yaw = 0;
    this.distance = 5;
    this.target = vec3.create(0, 0, 0);
    this.eye = vec3.create(0, 0, 5);
    this.up = vec3.create(0, 1, 0);
    this.right = vec3.create(1, 0, 0);
  }

  get view() { return mat4.lookAt(this.eye, this.target, this.up); }   // camera.ts:72-74
  get position() { return this.eye; }                                   // camera.ts:76-78

  // tumble about the origin, camera.ts:163-183
  updateCameraRotation(deltaX, deltaY) {
    this.target = vec3.create(0, 0, 0);
    this.yaw = -deltaX * this.rotateAmplitude;
    this.pitch = -deltaY * this.rotateAmplitude;
    const quatYaw = quat.fromAxisAngle(this.up, utils.degToRad(this.yaw));
    const back = vec3.normalize(vec3.sub(this.eye, this.target));
    this.right = vec3.transformQuat(vec3.cross(this.up, back), quatYaw);
    const quatPitch = quat.fromAxisAngle(this.right, utils.degToRad(this.pitch));
    const rotation = quat.mul(quatYaw, quatPitch);
    const newBack = vec3.transformQuat(back, rotation);
    this.up = vec3.normalize(vec3.cross(newBack, this.right));
    this.eye = vec3.add(this.target, vec3.mulScalar(newBack, this.distance));
  }

  // dolly, camera.ts:185-189
  updateCameraDistance(deltaX, deltaY) {
    this.distance -= (deltaX + deltaY) * this.scrollAmplitude;
    const back = vec3.normalize(vec3.sub(this.eye, this.target));
    this.eye = vec3.add(this.target, vec3.mulScalar(back, this.distance));
  }

  // track, camera.ts:191-202
  updateCameraTarget(deltaX, deltaY) {
    let pan = vec3.create(0, 0, 0);
    pan = vec3.addScaled(pan, this.right, -deltaX * this.panAmplitude);
    pan = vec3.addScaled(pan, this.up, deltaY * this.panAmplitude);
    this.target = vec3.add(this.target, pan);
    this.eye = vec3.add(this.eye, pan);
  }
}

module.exports = Camera;
