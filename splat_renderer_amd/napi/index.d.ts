// Type declarations for splat_renderer_amd/napi/index.js — the reference's class surface
// (ath92/splat-renderer src/*.ts) over libsplat_hip.so.
export type TypedArray = Float32Array | Uint32Array | Uint8Array | Int32Array;
export class Buffer {
  readonly device: Device;
  ptr: number;
  size: number;
  destroy(): void;
  write(data: TypedArray): this;
  read<T extends TypedArray>(out: T): T;
  zero(): void;
}
export class Device {
  constructor(ordinal?: number);
  /** the SplatProjector / GPUTileBinner that last ran on this device: what TileRenderer.render composites from unless bindTileData overrides */
  lastProjector: SplatProjector | null;
  lastBinner: GPUTileBinner | null;
  queue: { writeBuffer(buffer: Buffer, offset: number, data: TypedArray): void; submit(commandBuffers?: unknown[]): void; onSubmittedWorkDone(): Promise<void> };
  createBuffer(desc: number | { size: number }): Buffer;
  createBufferFrom(data: TypedArray): Buffer;
  createCommandEncoder(): CommandEncoder;
  sync(): void;
  rankStatus(): { policy: string; atomicsOrdered: boolean; orderFaults: number };
  compositeOptions(kernel?: 'quadrant' | 'pixel' | null, ahead?: number, predict?: boolean | null): void;
  forgetCompositeHistory(): void;
  setTiming(enabled: boolean, stageMask?: number, every?: number): void;
  stageTimeStats(stage: number): { samples: number; totalMs: number };
  timingConsumed(): { staged: number; consumed: number };
  destroy(): void;
}
export interface CommandEncoder { finish(): null; }
export class Camera {
  target: Float32Array;
  distance: number;
  azimuth: number;
  elevation: number;
  fov: number;
  aspect: number;
  near: number;
  far: number;
  setAspect(aspect: number): void;
  rotate(deltaAzimuth: number, deltaElevation: number): void;
  zoom(deltaDistance: number): void;
  pan(deltaX: number, deltaY: number): void;
  getViewProjectionMatrix(): Float32Array;
  getPosition(): Float32Array;
  uniforms(width: number, height: number, time?: number): Float32Array;
}
/** src/OrbitCameraController.ts:3-75; canvas: anything with addEventListener, or null (feed the handlers synthetic events) */
export interface PointerLikeEvent {
  clientX?: number;
  clientY?: number;
  button?: number;
  deltaY?: number;
  preventDefault?(): void;
}
export class OrbitCameraController {
  constructor(camera: Camera, canvas?: { addEventListener(type: string, handler: (e: PointerLikeEvent) => void, options?: unknown): void } | null);
  onMouseDown(event: PointerLikeEvent): void;
  onMouseMove(event: PointerLikeEvent): void;
  onMouseUp(event?: PointerLikeEvent): void;
  onWheel(event: PointerLikeEvent): void;
  destroy(): void;
}
/** the render loop of src/main.ts:110-193 for the tile-raster path, headless */
export class FrameLoop {
  constructor(device: Device, numPoints: number, width: number, height: number, tileSize?: number, camera?: Camera | null,
              rendererOptions?: { footprint?: Footprint; records?: "lit" | "projected" });
  readonly camera: Camera;
  readonly renderer: Renderer;
  frame: number;
  render(propertyBuffer: Buffer | PropertyPlanes, normalsBuffer: Buffer, time?: number): Buffer;
  readPixels(): Uint8Array;
  turntable(propertyBuffer: Buffer | PropertyPlanes, normalsBuffer: Buffer, frames: number, onFrame?: (k: number, rgba8: Uint8Array) => void): void;
  destroy(): void;
}
export class PointManager {
  constructor(device: Device, scene: Float32Array | { numPoints: number; seed?: number } | SDFScene, seed?: number);
  reinitialize(): void;
  swap(): void;
  getCurrentPositionBuffer(): Buffer;
  getNextPositionBuffer(): Buffer;
  getNumPoints(): number;
  destroy(): void;
}
export interface SceneNode { type: "primitive" | "operation"; }
export class Sphere {
  constructor(params?: { id?: string; position?: ArrayLike<number>; radius?: number });
  id: string;
  position: Float32Array;
  radius: number;
}
export class Box {
  constructor(params?: { id?: string; position?: ArrayLike<number>; size?: ArrayLike<number> });
  id: string;
  position: Float32Array;
  size: Float32Array;
}
export class Torus {
  constructor(params?: { id?: string; position?: ArrayLike<number>; majorRadius?: number; minorRadius?: number });
  id: string;
  position: Float32Array;
  majorRadius: number;
  minorRadius: number;
}
export class Capsule {
  constructor(params?: { id?: string; position?: ArrayLike<number>; height?: number; radius?: number });
  id: string;
  position: Float32Array;
  height: number;
  radius: number;
}
export type Primitive = Sphere | Box | Torus | Capsule;
export class SmoothUnion { constructor(k?: number); k: number; id: string; }
export function union(a: Primitive | SceneNode, b: Primitive | SceneNode): SceneNode;
export function intersection(a: Primitive | SceneNode, b: Primitive | SceneNode): SceneNode;
export function subtraction(a: Primitive | SceneNode, b: Primitive | SceneNode): SceneNode;
export function smoothUnion(k: number, a: Primitive | SceneNode, b: Primitive | SceneNode): SceneNode;
export class SDFScene {
  setRoot(node: Primitive | SceneNode): void;
  get(id: string): Primitive | undefined;
  getPrimitives(): Primitive[];
  getRoot(): SceneNode | null;
  getOperations(): unknown[];
  getStructureHash(): string;
  program(): Float32Array;
}
export class GradientSampler {
  constructor(device: Device, scene: SDFScene, numPoints: number);
  updateSceneParameters(): void;
  rebuildIfNeeded(): void;
  evaluateGradients(enc: CommandEncoder | null, uniformBuffer: Buffer | null, positionBuffer: Buffer): void;
  getGradientBuffer(): Buffer;
  getScene(): SDFScene;
  destroy(): void;
}
export class PositionUpdater {
  constructor(device: Device, shaderCode: string | null, numPoints: number);
  updatePositions(enc: CommandEncoder | null, uniformBuffer: Buffer | null, currentPositionBuffer: Buffer, gradientBuffer: Buffer, nextPositionBuffer: Buffer): void;
}
export class CurvatureSampler {
  constructor(device: Device, scene: SDFScene, numPoints: number);
  updateSceneParameters(): void;
  rebuildIfNeeded(): void;
  computeScaleFactors(enc: CommandEncoder | null, positionBuffer: Buffer): void;
  getScaleFactorsBuffer(): Buffer;
  getCurvatureBuffer(gradientBuffer: Buffer): Buffer;
  destroy(): void;
}
export class Comm {
  static uniqueId(): Uint8Array;
  constructor(device: Device, rank: number, world: number, idBytes: Uint8Array);
  readonly rank: number;
  readonly world: number;
  allGather(shardBuffer: Buffer, gatheredBuffer: Buffer, bytesPerRank: number): void;
  destroy(): void;
}
export class BandRenderer {
  constructor(device: Device, comm: Comm | null, numPoints: number, width: number, height: number, tileSize?: number);
  row0: number;
  row1: number;
  render(uniformData: Float32Array | Buffer, propertyBuffer: Buffer, normalsBuffer: Buffer): Buffer;
  settle(): number;
  pixelRows(): [number, number];
  readPixels(): Uint8Array;
  destroy(): void;
}
export interface PropertyPlanes { posRadius: Buffer; colorOpacity: Buffer; isPlanes: true; prelit?: boolean; }
export class SplatPropertyManager {
  constructor(device: Device, numSplats: number);
  updateFromCurvature(enc: CommandEncoder | null, positionBuffer: Buffer, curvatureBuffer: Buffer): void;
  updatePlanesFromCurvature(enc: CommandEncoder | null, positionBuffer: Buffer, curvatureBuffer: Buffer): PropertyPlanes;
  setFromArrays(props: Float32Array): void;
  getPropertyBuffer(): Buffer;
  getPropertyPlanes(): PropertyPlanes;
  getLitPlanes(normalsBuffer: Buffer): PropertyPlanes;
  destroy(): void;
}
export type Footprint = "isotropic" | "disc" | 0 | 1;
export class SplatProjector {
  constructor(device: Device, numSplats: number, footprint?: Footprint);
  project(enc: CommandEncoder | null, uniformBuffer: Buffer | Float32Array, splatPropertyBuffer: Buffer, keysBuffer?: Buffer | null, payloadBuffer?: Buffer | null, paddedSize?: number, normalsBuffer?: Buffer | null): void;
  getProjectedBuffer(): Buffer;
  getRecordsBuffer(): Buffer;
  contents: "projected" | "lit";
  getDiscBuffer(): Buffer;
  destroy(): void;
}
export class DepthKeyExtractor {
  constructor(device: Device);
  extract(enc: CommandEncoder | null, projectedBuffer: Buffer, keysBuffer: Buffer, payloadBuffer: Buffer, numSplats: number, paddedSize: number): void;
  cleanupTempBuffers(): void;
}
export class RadixSorter {
  constructor(device: Device, numSplats: number);
  readonly paddedSize: number;
  sort(numKeys?: number, bitBegin?: number, bitEnd?: number): void;
  getSortedIndicesBuffer(): Buffer;
  getKeysBuffer(): Buffer;
  getPayloadBuffer(): Buffer;
  cleanupTempBuffers(): void;
  destroy(): void;
}
export class PrefixSumScanner {
  constructor(device: Device);
  scan(enc: CommandEncoder | null, inputBuffer: Buffer, outputBuffer: Buffer, numElements: number): Promise<void>;
  cleanupTempBuffers(): void;
}
export class GPUTileBinner {
  constructor(device: Device, tileSize: number);
  setFrameOrder(order: "default" | "sortFirst" | "tileFirst"): void;
  binSplats(enc: CommandEncoder | null, projectedBuffer: Buffer, sortedIndicesBuffer: Buffer, numSplats: number, screenWidth: number, screenHeight: number): Promise<void>;
  getTileOffsetsBuffer(): Buffer;
  getTileIndicesBuffer(): Buffer;
  getTileCountsBuffer(): Buffer;
  getTotalIndices(): number;
  getTileSize(): number;
  cleanupTempBuffers(): void;
  destroy(): void;
}
export class PerTileSorter {
  constructor(device: Device, validate?: boolean);
  violations: number;
  sort(enc: CommandEncoder | null, projectedBuffer: Buffer, tileListsBuffer: Buffer, tileOffsetsBuffer: Buffer, splatIndicesBuffer: Buffer, numTiles: number, maxSplatsPerTile: number, totalPairs?: number): number | undefined;
  cleanupTempBuffers(): void;
  destroy(): void;
}
export class SequentialRenderer {
  constructor(device: Device, context?: unknown, presentationFormat?: string, numSplats?: number, tileSize?: number, footprint?: Footprint);
  render(uniformData: Float32Array | Buffer, splatPropertyBuffer: Buffer, sortedIndexBuffer: Buffer, curvatureBuffer: Buffer, width: number, height: number): void;
  readPixels(): Uint8Array;
  destroy(): void;
}
export class ComputeShaderRenderer {
  constructor(device: Device, context?: unknown, presentationFormat?: string, options?: { mode?: number; earlyOut?: boolean; footprint?: Footprint; recordFormat?: number });
  recordFormat: number;
  ensureOutputTexture(width: number, height: number): void;
  render(uniformData: Float32Array, splatPropertyBuffer: Buffer, splatIndicesBuffer: Buffer, curvatureBuffer: Buffer, projectedBuffer: Buffer, tileListsBuffer: Buffer, tileOffsetsBuffer: Buffer, tileSize: number, numTilesX: number, width: number, height: number): void;
  readPixels(): Uint8Array;
  destroy(): void;
}
/** render() has the reference's eleven arguments (src/TileRenderer.ts:234-246) and needs nothing else: the projected records and
 *  tile offsets are those of the device's last SplatProjector / GPUTileBinner; bindTileData overrides them. */
export class TileRenderer extends ComputeShaderRenderer {
  constructor(device: Device, context?: unknown, presentationFormat?: string, options?: { mode?: number; earlyOut?: boolean; footprint?: Footprint; recordFormat?: number });
  bindTileData(projectedBuffer: Buffer, tileCountsBuffer: Buffer, tileOffsetsBuffer: Buffer): void;
  // @ts-ignore (the reference's TileRenderer.render: another argument list than ComputeShaderRenderer.render, and async)
  render(uniformData: Float32Array, splatPropertyBuffer: Buffer, splatIndicesBuffer: Buffer, curvatureBuffer: Buffer, tileCountsData: Uint32Array | Buffer, numTilesX: number, numTilesY: number, tileSize: number, maxSplatsPerTile: number, width: number, height: number): Promise<void>;
}
export class Renderer {
  constructor(device: Device, context?: unknown, presentationFormat?: string, numPoints?: number, tileSize?: number, options?: { footprint?: Footprint; records?: "lit" | "projected" });
  recordFormat: number;
  render(uniformData: Float32Array | Buffer, propertyBuffer: Buffer | PropertyPlanes, normalsBuffer: Buffer, scaleFactorsBuffer: Buffer | null, width: number, height: number): Buffer;
  finish(): number;
  readPixels(): Uint8Array;
  destroy(): void;
}
export const MODE_FRONT_TO_BACK: 0; export const MODE_REFERENCE_LITERAL: 1;
export const FOOTPRINT_ISOTROPIC: 0; export const FOOTPRINT_DISC: 1;
export const RECORDS_PROJECTED: 0; export const RECORDS_COMPACT: 1; export const RECORDS_LIT32: 3;
