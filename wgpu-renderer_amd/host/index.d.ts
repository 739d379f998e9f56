// Type declarations of the headless host (CommonJS module `index.js`).  Names follow the reference:
// Renderer (src/renderer.ts), Camera (src/camera.ts), glTFLoader / Geometry (src/glTF.ts),
// HDRLoader / HDR (src/hdr.ts).
export type Mat4 = Float32Array;
export type Vec3 = Float32Array;

export type HDR = { width: number; height: number; data: Float32Array };

export type Layout = { format: string; offset: number };

export type Geometry = {
  model: Mat4;
  vertices: Float32Array;
  arrayStride: number;
  position: Layout;
  normal: Layout;
  texCoord: Layout;
  indices: Uint16Array | Uint32Array;   // 32-bit only from the general loader, for meshes beyond 65 536 vertices
  textures: {
    baseColorURI: string;
    metallicRoughnessURI: string;
    normalURI: string;
    emissive: [number, number, number];
    emissiveURI: string;
    occlusionURI: string;
  };
};

export class Camera {
  constructor();
  readonly view: Mat4;
  readonly position: Vec3;
  updateCameraRotation(deltaX: number, deltaY: number): void;
  updateCameraDistance(deltaX: number, deltaY: number): void;
  updateCameraTarget(deltaX: number, deltaY: number): void;
}

export class glTFLoader {
  constructor(options?: { referenceLimits?: boolean });
  load(url: string): Promise<void>;
  getGeometries(): Geometry[];
}

export class HDRLoader {
  constructor(options?: { referenceLimits?: boolean });
  readonly hdr: HDR;
  load(url: string): Promise<void>;
}

export type FrameStats = {
  msVertex: number; msSetup: number; msRaster: number; msResolve: number; msTotal: number;
  coveredPixels: number; spilledRefs: number; setupTriangles: number; uniformQuads: number; mixedQuads: number;
};

/** vis: triangle id + 1 per pixel (0 = skybox); four per pixel, in sample order, when msaaOptions.enable is set */
export type Frame = { rgba: Uint8Array; vis?: Uint32Array; stats: FrameStats };

export type RendererOptions = {
  width?: number; height?: number; device?: number; modelPath?: string; hdrPath?: string; hdr?: HDR;
  referenceLimits?: boolean; skyboxSize?: number; irradianceSize?: number; prefilterSize?: number; prefilterMips?: number; lutSize?: number;
};

export class Renderer {
  constructor(options?: RendererOptions);
  camera: Camera;
  width: number;
  height: number;
  textureOptions: { mipmaps: boolean; addressModeU: string; addressModeV: string; magFilter: string; minFilter: string };
  msaaOptions: { enable: boolean };
  pixels: Uint8Array | null;
  frameTime: number;
  render(): Promise<Frame>;
  init(): Promise<void>;
  frameUniforms(): Float32Array;
  drawFrame(options?: { y0?: number; y1?: number; wantVis?: boolean }): Frame;
  downloadMap(which: number, level?: number): { width: number; height: number; layers: number; data: Float32Array | Uint16Array };
  destroy(): void;
}
