'use strict';
/**
 * Host classes of the tile-raster hot path over the N-API addon (splat_napi.node -> libsplat_hip.so).
 *
 * Same class names, constructor arguments, verbs and error behaviour as the TypeScript classes of
 * ath92/splat-renderer (file:line below are under that repository's src/), so a call site written
 * against the reference keeps working: `device` is a Device (GPUDevice equivalent: one HIP device +
 * one stream), a GPUBuffer is a Buffer {ptr, size}, a GPUCommandEncoder argument is accepted and
 * ignored (recording = immediate enqueue on the stream).
 *
 * SOURCE AND TWIN: index.ts is the source; index.js is its twin.  index.ts is TypeScript restricted to syntax that erases
 * without moving a character of what remains — `declare` lines, `type` / `interface` statements, and annotations in the
 * headers of functions and class members (strip_types.js lists the forms) — so that the CommonJS / ES2019 file this
 * image's Node 12 loads, index.js, is `node strip_types.js index.ts > index.js`, line for line (tests/test_napi.py holds
 * the committed twin to exactly that).  No TypeScript compiler exists in this image: the annotations follow index.d.ts
 * (the hand-written public surface) and have not been through `tsc`; fields and helpers index.d.ts does not list are
 * typed from their use, `any` where that is not a single type.
 */
declare function require(name: string): any;
declare const module: { exports: any };
type TypedArray = Float32Array | Uint32Array | Uint8Array | Int32Array;
interface CommandEncoder { finish(): null; }
interface RankStatus { policy: string; atomicsOrdered: boolean; orderFaults: number }
interface StageTimeStats { samples: number; totalMs: number }
interface EntriesCounted { staged: number; consumed: number }
interface PointerLikeEvent {
  clientX?: number;
  clientY?: number;
  button?: number;
  deltaY?: number;
  preventDefault?(): void;
}
interface SceneNode { type: "primitive" | "operation"; }
interface PropertyPlanes { posRadius: Buffer_; colorOpacity: Buffer_; isPlanes: true; prelit?: boolean; }
type Footprint = "isotropic" | "disc" | 0 | 1;
const native = require('./splat_napi.node');

const U32_MAX = 0xffffffff;
const MODE_FRONT_TO_BACK = 0;
const FOOTPRINT_ISOTROPIC = 0, FOOTPRINT_DISC = 1;
const MODE_REFERENCE_LITERAL = 1;
const RECORDS_PROJECTED = 0, RECORDS_COMPACT = 1, RECORDS_LIT32 = 3;

class Buffer_ {
  declare device: Device;
  declare ptr: number;
  declare size: number;
  declare owned: boolean;
  declare hostShadow: Float32Array | null;
  constructor(device: Device, ptr: number, size: number, owned: boolean = true) {
    this.device = device;
    this.ptr = ptr;
    this.size = size;
    this.owned = owned;
    this.hostShadow = null;
  }
  destroy(): void {
    if (this.owned && this.ptr) native.buf_free(this.device.ctx, this.ptr);
    this.ptr = 0;
  }
  write(typedArray: TypedArray): this {
    native.buf_upload(this.device.ctx, this.ptr, typedArray);
    if (this.size <= 256) this.hostShadow = new Float32Array(typedArray.buffer.slice(typedArray.byteOffset, typedArray.byteOffset + typedArray.byteLength));
    return this;
  }
  read<T extends TypedArray>(typedArray: T): T {
    native.buf_download(this.device.ctx, typedArray, this.ptr);
    return typedArray;
  }
  zero(): void { native.buf_zero(this.device.ctx, this.ptr, this.size); }
}

class Device {
  declare ctx: unknown;
  declare lastProjector: SplatProjector | null;
  declare lastBinner: GPUTileBinner | null;
  declare queue: { writeBuffer(buffer: Buffer_, offset: number, data: TypedArray): void; submit(commandBuffers?: unknown[]): void; onSubmittedWorkDone(): Promise<void> };
  constructor(ordinal: number = 0) {
    this.ctx = native.ctx_create(ordinal);
    // the SplatProjector and GPUTileBinner that last ran on this device (their project() / binSplats(), or a Renderer's frame):
    // what TileRenderer.render — whose reference signature names neither — composites from unless told otherwise
    this.lastProjector = null;
    this.lastBinner = null;
    const self = this;
    this.queue = {
      writeBuffer(buffer, offset, data) {
        if (offset !== 0) throw new Error('writeBuffer: only offset 0 is supported');
        buffer.write(data);
      },
      submit() {},
      onSubmittedWorkDone() {
        self.sync();
        return Promise.resolve();
      },
    };
  }
  createBuffer(desc: number | { size: number }): Buffer_ {
    const size = typeof desc === 'number' ? desc : desc.size;
    return new Buffer_(this, native.buf_alloc(this.ctx, size), size);
  }
  createBufferFrom(typedArray: TypedArray): Buffer_ { return this.createBuffer(Math.max(typedArray.byteLength, 16)).write(typedArray); }
  createCommandEncoder(): CommandEncoder { return { finish() { return null; } }; }
  sync(): void { native.sync(this.ctx); }
  /** How this context ranks equal digits in its sort kernels (include/splat.h, NOTE on ranking).  orderFaults counts the frames
   *  whose tile lists failed the per-tile sort's order check and were rendered again: anything but 0 means the context has
   *  switched to ballot ranking for good (policy 'ballot', the slower path) and is worth a report. */
  rankStatus(): RankStatus {
    const s = native.rank_status(this.ctx);
    return { policy: ['checked', 'atomic', 'ballot'][s[0]], atomicsOrdered: s[1] === 1, orderFaults: s[2] };
  }
  /** splat_composite_options: which composite kernel this context runs and how far its builder looks ahead (ahead / predict
   *  change the schedule only: same bytes; the two kernels agree within the composite's stated tolerance).  kernel null (process
   *  default) | 'quadrant' | 'pixel'; ahead 0 (default) | 1 | 2; predict null | boolean (the look-ahead bound from the previous
   *  launch's per-tile costs).  EVERY call sets all three: null / 0 = the process default, i.e. the environment's (not "as it
   *  was"); a choice made here takes precedence over the environment variable.  host.py Device.compositeOptions. */
  compositeOptions(kernel: string | null = null, ahead: number = 0, predict: boolean | null = null): void {
    const k = kernel === null ? -1 : kernel === 'quadrant' ? 0 : kernel === 'pixel' ? 1 : NaN;
    if (Number.isNaN(k)) throw new Error("compositeOptions: kernel is null, 'quadrant' or 'pixel'");
    native.composite_options(this.ctx, k, ahead, predict === null ? -1 : predict ? 1 : 0);
  }
  /** splat_composite_forget_history: the next composite behaves like a context's first (row-major tile order, no look-ahead bound). */
  forgetCompositeHistory(): void { native.composite_forget_history(this.ctx); }
  /** splat_set_timing / _stages / _sampling: HIP-event timing of the stages in `stageMask` (bit = stage id; 0 = all), every
   *  `every`-th launch. */
  setTiming(enabled: boolean, stageMask: number = 0, every: number = 1): void {
    native.set_timing(this.ctx, enabled ? 1 : 0);
    if (enabled && stageMask) native.set_timing_stages(this.ctx, stageMask);
    if (enabled) native.set_timing_sampling(this.ctx, every);
  }
  /** splat_stage_time_stats: launches timed and their total since timing was switched on. */
  stageTimeStats(stage: number): StageTimeStats {
    const s = native.stage_time_stats(this.ctx, stage);
    return { samples: s[0], totalMs: s[1] };
  }
  /** splat_timing_consumed: list entries staged and consumed by the timed composites (counted only when asked for in setTiming's mask). */
  timingConsumed(): EntriesCounted {
    const s = native.timing_consumed(this.ctx);
    return { staged: s[0], consumed: s[1] };
  }
  destroy(): void {
    if (this.ctx) native.ctx_destroy(this.ctx);
    this.ctx = null;
  }
}

function uniformFloats(u: Float32Array | Buffer_ | ArrayLike<number>): Float32Array {
  if (u instanceof Buffer_) {
    if (!u.hostShadow) throw new Error('uniform buffer was never written');
    u = u.hostShadow;
  }
  if (!(u instanceof Float32Array)) u = Float32Array.from(u);
  return u;
}

/** src/SplatPropertyManager.ts:13-181 */
class SplatPropertyManager {
  declare device: Device;
  declare numSplats: number;
  declare propertyBuffer: Buffer_ | null;
  declare planesValid: boolean;
  declare litValid: boolean;
  declare planes: PropertyPlanes | null;
  declare lit: PropertyPlanes | null;
  declare litNormals: unknown; // (the normals buffer's device pointer the lit plane was shaded from)
  constructor(device: Device, numSplats: number) {
    this.device = device;
    this.numSplats = numSplats;
    this.propertyBuffer = device.createBuffer(numSplats * 32);
    this.propertyBuffer.write(SplatPropertyManager.defaultProperties(numSplats)); // initializeDefaults :33-50
  }
  // :33-50: position 0, radius 0.04, white, opacity 0.7 (held to an execution of the reference's loop: tests/golden/ref_host.json)
  static defaultProperties(numSplats: number): Float32Array {
    const data = new Float32Array(numSplats * 8);
    for (let i = 0; i < numSplats; i++) {
      data[i * 8 + 3] = 0.04;
      data[i * 8 + 4] = 1;
      data[i * 8 + 5] = 1;
      data[i * 8 + 6] = 1;
      data[i * 8 + 7] = 0.7;
    }
    return data;
  }
  updateFromCurvature(commandEncoder: CommandEncoder | null, positionBuffer: Buffer_, curvatureBuffer: Buffer_): void { // :153-173
    native.update_props(this.device.ctx, positionBuffer.ptr, curvatureBuffer.ptr, this.numSplats, this.propertyBuffer.ptr);
    this.planesValid = false;
    this.litValid = false;
  }
  setFromArrays(props: Float32Array): void {
    this.propertyBuffer.write(props);
    this.planesValid = false;
    this.litValid = false;
  }
  getPropertyBuffer(): Buffer_ { return this.propertyBuffer; } // :175-177
  // the MI355X-native layout: two vec4 planes {posRadius, colorOpacity}; Renderer.render takes either
  getPropertyPlanes(): PropertyPlanes {
    if (!this.planes) this.planes = { posRadius: this.device.createBuffer(this.numSplats * 16), colorOpacity: this.device.createBuffer(this.numSplats * 16), isPlanes: true };
    if (!this.planesValid) {
      native.props_to_planes(this.device.ctx, this.propertyBuffer.ptr, this.numSplats, this.planes.posRadius.ptr, this.planes.colorOpacity.ptr);
      this.planesValid = true;
    }
    return this.planes;
  }
  // the planes with the colour plane already lit by the given normals (kd = 0.85 + 0.15 max(n.l, 0), once per
  // property update instead of once per staged list entry): the frame then gathers one line less per entry
  getLitPlanes(normalsBuffer: Buffer_): PropertyPlanes {
    const p = this.getPropertyPlanes();
    if (!this.lit) this.lit = { posRadius: p.posRadius, colorOpacity: this.device.createBuffer(this.numSplats * 16), isPlanes: true, prelit: true };
    if (!this.litValid || this.litNormals !== normalsBuffer.ptr) {
      native.lit_colors(this.device.ctx, p.colorOpacity.ptr, 1, normalsBuffer.ptr, 1, this.numSplats, this.lit.colorOpacity.ptr);
      this.litValid = true;
      this.litNormals = normalsBuffer.ptr;
    }
    return this.lit;
  }
  updatePlanesFromCurvature(commandEncoder: CommandEncoder | null, positionBuffer: Buffer_, curvatureBuffer: Buffer_): PropertyPlanes {
    const p = this.getPropertyPlanes();
    native.update_props_planes(this.device.ctx, positionBuffer.ptr, curvatureBuffer.ptr, this.numSplats, p.posRadius.ptr, p.colorOpacity.ptr);
    return p;
  }
  destroy(): void { // :179-181
    this.propertyBuffer.destroy();
    if (this.planes) {
      this.planes.posRadius.destroy();
      this.planes.colorOpacity.destroy();
      this.planes = null;
    }
    if (this.lit) {
      this.lit.colorOpacity.destroy();
      this.lit = null;
    }
  }
}

/** src/SplatProjector.ts:5-203 */
function footprintCode(f: Footprint | undefined): number {
  if (f === undefined || f === null || f === 'isotropic' || f === FOOTPRINT_ISOTROPIC) return FOOTPRINT_ISOTROPIC;
  if (f === 'disc' || f === FOOTPRINT_DISC) return FOOTPRINT_DISC;
  throw new Error(`footprint must be 'isotropic' or 'disc', not ${f}`);
}
/** footprint 'disc' (extension): SequentialRenderer's oriented disc — project() then needs normalsBuffer, the bounds are
 * the disc's exact screen extent and getDiscBuffer() holds the 32-byte records the composite evaluates. */
class SplatProjector {
  declare device: Device;
  declare numSplats: number;
  declare footprint: number;
  declare projectedBuffer: Buffer_ | null;
  declare contents: "projected" | "lit";
  declare discBuffer: Buffer_ | null;
  constructor(device: Device, numSplats: number, footprint: Footprint = 'isotropic') {
    this.device = device;
    this.numSplats = numSplats;
    this.footprint = footprintCode(footprint);
    this.projectedBuffer = device.createBuffer(numSplats * 32);
    this.contents = 'projected';
    this.discBuffer = this.footprint === FOOTPRINT_DISC ? device.createBuffer(numSplats * 32) : null;
  }
  project(commandEncoder: CommandEncoder | null, uniformBuffer: Buffer_ | Float32Array, splatPropertyBuffer: Buffer_, keysBuffer: Buffer_ | null = null, payloadBuffer: Buffer_ | null = null, paddedSize: number = 0, normalsBuffer: Buffer_ | null = null): void { // :174-194
    const u = uniformFloats(uniformBuffer);
    if (u.length < 22) throw new Error('uniform block needs 22 floats (VP, eye, time, screenW, screenH)');
    const keys = keysBuffer ? keysBuffer.ptr : null, payload = payloadBuffer ? payloadBuffer.ptr : null;
    this.device.lastProjector = this;
    this.contents = 'projected';
    if (this.footprint === FOOTPRINT_DISC) {
      if (!normalsBuffer) throw new Error("SplatProjector(footprint 'disc').project needs normalsBuffer");
      native.project_disc(this.device.ctx, u, splatPropertyBuffer.ptr, 2, normalsBuffer.ptr, 1, this.numSplats, this.projectedBuffer.ptr,
        this.discBuffer.ptr, keys, payload, paddedSize);
      return;
    }
    native.project(this.device.ctx, u, splatPropertyBuffer.ptr, 2, this.numSplats, this.projectedBuffer.ptr, keys, payload, paddedSize);
  }
  // :196-198.  Throws when the last frame left the 32-byte LIT composite records {centre.xy, radius, depth | lit rgb,
  // opacity} here instead of ProjectedSplat records (Renderer records 'lit', the whole-frame facade's default): code
  // written against the reference's layout must not read those by accident — getRecordsBuffer() hands them out.
  getProjectedBuffer(): Buffer_ {
    if (this.contents === 'lit') {
      throw new Error("the projector's buffer holds lit composite records (Renderer records 'lit'), not ProjectedSplat records: " +
        "use getRecordsBuffer() and Renderer.recordFormat, or new Renderer(..., { records: 'projected' })");
    }
    return this.projectedBuffer;
  }
  getRecordsBuffer(): Buffer_ { return this.projectedBuffer; } // whatever the last frame wrote (this.contents: 'projected' | 'lit')
  getDiscBuffer(): Buffer_ {
    if (!this.discBuffer) throw new Error("getDiscBuffer: this projector was not created with footprint 'disc'");
    return this.discBuffer;
  }
  destroy(): void {
    if (this.device.lastProjector === this) this.device.lastProjector = null;
    this.projectedBuffer.destroy();
    if (this.discBuffer) this.discBuffer.destroy();
  }          // :200-202
}

/** src/DepthKeyExtractor.ts:5-115 */
class DepthKeyExtractor {
  declare device: Device;
  constructor(device: Device) { this.device = device; }
  extract(commandEncoder: CommandEncoder | null, projectedBuffer: Buffer_, keysBuffer: Buffer_, payloadBuffer: Buffer_, numSplats: number, paddedSize: number): void { // :71-109
    native.extract_keys(this.device.ctx, projectedBuffer.ptr, numSplats, paddedSize, keysBuffer.ptr, payloadBuffer.ptr);
  }
  cleanupTempBuffers(): void {}
}

/** src/RadixSorter.ts:21-301 */
class RadixSorter {
  declare device: Device;
  declare numSplats: number;
  declare handle: unknown;
  declare paddedSize: number;
  constructor(device: Device, numSplats: number) {
    this.device = device;
    this.numSplats = numSplats;
    this.handle = native.sort_create(device.ctx, numSplats);
    this.paddedSize = native.sort_capacity(this.handle);
    // :46-52
  }
  sort(numKeys: number = this.numSplats, bitBegin: number = 0, bitEnd: number = 32): void { native.sort_run(this.device.ctx, this.handle, numKeys, bitBegin, bitEnd); } // :197-264
  getSortedIndicesBuffer(): Buffer_ { return new Buffer_(this.device, native.sort_sorted_payload(this.handle), this.paddedSize * 4, false); } // :269-271
  getKeysBuffer(): Buffer_ { return new Buffer_(this.device, native.sort_keys(this.handle), this.paddedSize * 4, false); }       // :273-275
  getPayloadBuffer(): Buffer_ { return new Buffer_(this.device, native.sort_payload(this.handle), this.paddedSize * 4, false); } // :277-279
  cleanupTempBuffers(): void {}
  destroy(): void {
    if (this.handle) native.sort_destroy(this.handle);
    this.handle = null;
  }
}

/** src/PrefixSumScanner.ts:8-168 */
class PrefixSumScanner {
  declare device: Device;
  constructor(device: Device) { this.device = device; }
  async scan(commandEncoder: CommandEncoder | null, inputBuffer: Buffer_, outputBuffer: Buffer_, numElements: number): Promise<void> { // :74-87 (async in the reference because of its CPU fallback)
    native.scan_u32(this.device.ctx, inputBuffer.ptr, outputBuffer.ptr, numElements, null);
  }
  cleanupTempBuffers(): void {}
}

/** src/GPUTileBinner.ts:11-378 */
class GPUTileBinner {
  declare device: Device;
  declare tileSize: number;
  declare handle: unknown;
  declare prefixSumScanner: PrefixSumScanner;
  declare numTiles: number;
  constructor(device: Device, tileSize: number) {
    this.device = device;
    this.tileSize = tileSize;
    this.handle = native.bin_create(device.ctx, tileSize);
    this.prefixSumScanner = new PrefixSumScanner(device);
    // :49
    this.numTiles = 0;
  }
  // order of work of the whole-frame call: 'tileFirst' (bin in index order, PerTileSorter-style depth sort per tile;
  // the default), 'sortFirst' (global depth sort, bin in sorted order) or 'default'; same lists either way
  setFrameOrder(order: "default" | "sortFirst" | "tileFirst"): void { native.bin_set_frame_order(this.device.ctx, this.handle, { default: -1, sortFirst: 0, tileFirst: 1 }[order]); }
  async binSplats(commandEncoder: CommandEncoder | null, projectedBuffer: Buffer_, sortedIndicesBuffer: Buffer_, numSplats: number, screenWidth: number, screenHeight: number): Promise<void> { // :190-338
    native.bin_run(this.device.ctx, this.handle, projectedBuffer.ptr, numSplats, sortedIndicesBuffer.ptr, numSplats, screenWidth, screenHeight, 0, U32_MAX);
    this.numTiles = Math.ceil(screenWidth / this.tileSize) * Math.ceil(screenHeight / this.tileSize);
    this.device.lastBinner = this;
  }
  // the natives throw Error("... Tile offsets buffer not initialized") etc. before binSplats, as :340-359
  getTileOffsetsBuffer(): Buffer_ { return new Buffer_(this.device, native.bin_offsets(this.device.ctx, this.handle), this.numTiles * 4, false); }
  getTileIndicesBuffer(): Buffer_ {
    const p = native.bin_indices(this.device.ctx, this.handle);
    return new Buffer_(this.device, p, Math.max(4, this.getTotalIndices() * 4), false);
  }
  getTileCountsBuffer(): Buffer_ { return new Buffer_(this.device, native.bin_counts(this.device.ctx, this.handle), this.numTiles * 4, false); }
  getTotalIndices(): number { return native.bin_total(this.device.ctx, this.handle); }
  getTileSize(): number { return this.tileSize; } // :361-363
  cleanupTempBuffers(): void { this.prefixSumScanner.cleanupTempBuffers(); }
  destroy(): void {
    if (this.device.lastBinner === this) this.device.lastBinner = null;
    if (this.handle) native.bin_destroy(this.handle);
    this.handle = null;
  }
}

/** src/PerTileSorter.ts:6-223 — lists leave GPUTileBinner already in (depth key, index) order, so sort()
 * reorders nothing; with validate=true it runs the order check on the device and returns the number
 * of out-of-order neighbours (0). */
class PerTileSorter {
  declare device: Device;
  declare validate: boolean;
  declare violations: number;
  constructor(device: Device, validate: boolean = false) {
    this.device = device;
    this.validate = validate;
    this.violations = 0;
  }
  sort(commandEncoder: CommandEncoder | null, projectedBuffer: Buffer_, tileListsBuffer: Buffer_, tileOffsetsBuffer: Buffer_, splatIndicesBuffer: Buffer_, numTiles: number, maxSplatsPerTile: number, totalPairs?: number): number | undefined { // :174-213
    if (!this.validate) return undefined;
    const total = totalPairs === undefined ? splatIndicesBuffer.size / 4 : totalPairs;
    this.violations = native.validate_tile_order(this.device.ctx, projectedBuffer.ptr, tileOffsetsBuffer.ptr, numTiles, splatIndicesBuffer.ptr, total);
    return this.violations;
  }
  cleanupTempBuffers(): void {}
  destroy(): void {}
}

/** src/ComputeShaderRenderer.ts:5-469 (the canvas blit :268-338 is out of scope) */
class ComputeShaderRenderer {
  declare device: Device;
  declare mode: number;
  declare earlyOut: boolean;
  declare footprint: number;
  declare recordFormat: number;
  declare outputTexture: Buffer_ | null;
  declare width: number;
  declare height: number;
  constructor(device: Device, context: unknown = null, presentationFormat: string = 'rgba8unorm', options: { mode?: number; earlyOut?: boolean; footprint?: Footprint; recordFormat?: number } = {}) {
    this.device = device;
    this.mode = options.mode || MODE_FRONT_TO_BACK;
    this.earlyOut = options.earlyOut !== false;
    this.footprint = footprintCode(options.footprint);
    // RECORDS_LIT32: projectedBuffer in render() holds lit composite records (what a Renderer with records 'lit' leaves in its
    // projector's buffer); colours and normals are then not read
    this.recordFormat = options.recordFormat || RECORDS_PROJECTED;
    // 'disc': projectedBuffer in render() is the disc projector's getDiscBuffer()
    this.outputTexture = null;
    this.width = 0;
    this.height = 0;
  }
  ensureOutputTexture(width: number, height: number): void { // :340-360
    if (this.width !== width || this.height !== height) {
      if (this.outputTexture) this.outputTexture.destroy();
      this.outputTexture = this.device.createBuffer(width * height * 4);
      this.width = width;
      this.height = height;
    }
  }
  render(uniformData: Float32Array, splatPropertyBuffer: Buffer_, splatIndicesBuffer: Buffer_, curvatureBuffer: Buffer_, projectedBuffer: Buffer_, tileListsBuffer: Buffer_, tileOffsetsBuffer: Buffer_, tileSize: number, numTilesX: number, width: number, height: number): void { // :362-462
    if (numTilesX !== Math.ceil(width / tileSize)) throw new Error('numTilesX does not match ceil(width / tileSize)');
    this.ensureOutputTexture(width, height);
    native.composite(this.device.ctx, [this.mode, this.earlyOut ? 1 : 0, tileSize, 0, U32_MAX, this.recordFormat, 0, this.footprint], splatPropertyBuffer.ptr + 16, 2, curvatureBuffer.ptr, 1,
      projectedBuffer.ptr, splatIndicesBuffer.ptr, tileListsBuffer.ptr, tileOffsetsBuffer.ptr, width, height, this.outputTexture.ptr, null);
  }
  readPixels(): Uint8Array { return this.outputTexture.read(new Uint8Array(this.width * this.height * 4)); }
  destroy(): void {
    if (this.outputTexture) this.outputTexture.destroy();
    this.outputTexture = null;
  } // :464-468
}

/** src/TileRenderer.ts:5-355 — fronts the same composite.  render() has the reference's eleven arguments (:234-246) and runs with
 * nothing else: the projected records and the prefix-sum offsets the composite needs, which that signature does not name, are
 * those of the SplatProjector and GPUTileBinner that last ran on the device (Device.lastProjector / lastBinner); bindTileData
 * overrides them.  tileCountsData is the reference's host Uint32Array of counts per tile (its length is checked; the
 * device-resident counts are what the kernel reads) or, as an extension, the device buffer itself. */
class TileRenderer extends ComputeShaderRenderer {
  declare bound: Buffer_[] | null;
  declare formatGiven: boolean;
  constructor(device: Device, context: unknown = null, presentationFormat: string = 'rgba8unorm', options: { mode?: number; earlyOut?: boolean; footprint?: Footprint; recordFormat?: number } = {}) {
    super(device, context, presentationFormat, options);
    this.bound = null;
    this.formatGiven = options.recordFormat !== undefined;
  }
  bindTileData(projectedBuffer: Buffer_, tileCountsBuffer: Buffer_, tileOffsetsBuffer: Buffer_): void { this.bound = [projectedBuffer, tileCountsBuffer, tileOffsetsBuffer]; }
  async render(uniformData: Float32Array, splatPropertyBuffer: Buffer_, splatIndicesBuffer: Buffer_, curvatureBuffer: Buffer_, tileCountsData: Uint32Array | Buffer_, numTilesX: number, numTilesY: number, tileSize: number, maxSplatsPerTile: number, width: number, height: number): Promise<void> { // :234-348
    if (numTilesY !== Math.ceil(height / tileSize)) throw new Error('numTilesY does not match ceil(height / tileSize)');
    if (tileCountsData instanceof Uint32Array && tileCountsData.length !== numTilesX * numTilesY) throw new Error('tileCountsData does not hold one count per tile (numTilesX * numTilesY)');
    let projected, counts, offsets;
    if (this.bound) {
      projected = this.bound[0];
      counts = this.bound[1];
      offsets = this.bound[2];
    } else {
      const p = this.device.lastProjector, b = this.device.lastBinner;
      if (!p || !b) throw new Error('TileRenderer.render: no SplatProjector / GPUTileBinner has run on this device yet (and bindTileData was not called)');
      if (this.footprint === FOOTPRINT_DISC) projected = p.getDiscBuffer();
      else {
        projected = p.getRecordsBuffer();
        if (!this.formatGiven) this.recordFormat = p.contents === 'lit' ? RECORDS_LIT32 : RECORDS_PROJECTED;
      }
      counts = tileCountsData instanceof Buffer_ ? tileCountsData : b.getTileCountsBuffer();
      offsets = b.getTileOffsetsBuffer();
    }
    super.render(uniformData, splatPropertyBuffer, splatIndicesBuffer, curvatureBuffer, projected, counts, offsets, tileSize, numTilesX, width, height);
  }
}

/** src/SequentialRenderer.ts:5-321 — ordering-exact path: composites exactly the order of the caller's sorted index
 * buffer (near-to-far) with the reference's own footprint, the oriented disc of its vertex/fragment pair (:91-142),
 * evaluated per pixel through the inverse plane-to-screen homography (footprint 'disc', the default); 'isotropic'
 * composites the same order with ComputeShaderRenderer's screen-space Gaussian. */
class SequentialRenderer {
  declare device: Device;
  declare numSplats: number;
  declare tileSize: number;
  declare projector: SplatProjector;
  declare binner: GPUTileBinner;
  declare compositor: ComputeShaderRenderer;
  constructor(device: Device, context: unknown = null, presentationFormat: string = 'rgba8unorm', numSplats: number = 0, tileSize: number = 16, footprint: Footprint = 'disc') {
    this.device = device;
    this.numSplats = numSplats;
    this.tileSize = tileSize;
    this.projector = new SplatProjector(device, numSplats, footprint);
    this.binner = new GPUTileBinner(device, tileSize);
    this.compositor = new ComputeShaderRenderer(device, context, presentationFormat, { footprint });
  }
  render(uniformData: Float32Array | Buffer_, splatPropertyBuffer: Buffer_, sortedIndexBuffer: Buffer_, curvatureBuffer: Buffer_, width: number, height: number): void { // :233-314
    let u = uniformFloats(uniformData);
    if (u.length < 22) {
      const v = new Float32Array(22);
      v.set(u.subarray(0, 20));
      v[20] = width;
      v[21] = height;
      u = v;
    }
    const disc = this.projector.footprint === FOOTPRINT_DISC;
    this.projector.project(null, u, splatPropertyBuffer, null, null, 0, disc ? curvatureBuffer : null);
    native.bin_run(this.device.ctx, this.binner.handle, this.projector.getProjectedBuffer().ptr, this.numSplats, sortedIndexBuffer.ptr, this.numSplats, width, height, 0, U32_MAX);
    this.binner.numTiles = Math.ceil(width / this.tileSize) * Math.ceil(height / this.tileSize);
    this.compositor.render(u, splatPropertyBuffer, this.binner.getTileIndicesBuffer(), curvatureBuffer, disc ? this.projector.getDiscBuffer() : this.projector.getProjectedBuffer(),
      this.binner.getTileCountsBuffer(), this.binner.getTileOffsetsBuffer(), this.tileSize, Math.ceil(width / this.tileSize), width, height);
  }
  readPixels(): Uint8Array { return this.compositor.readPixels(); }
  destroy(): void {
    this.projector.destroy();
    this.binner.destroy();
    this.compositor.destroy();
  } // :316-320
}

/** src/Renderer.ts:13,250,311 — name kept as the whole-frame facade (project -> keys -> sort -> bin -> composite) */
class Renderer {
  declare device: Device;
  declare numPoints: number;
  declare tileSize: number;
  declare footprint: number;
  declare records: string;
  declare projector: SplatProjector;
  declare sorter: RadixSorter;
  declare binner: GPUTileBinner;
  declare output: Buffer_ | null;
  declare width: number;
  declare height: number;
  declare last: [Float32Array | Buffer_, Buffer_ | PropertyPlanes, Buffer_, Buffer_ | null, number, number] | null;
  declare recordFormat: number;
  constructor(device: Device, context: unknown = null, presentationFormat: string = 'rgba8unorm', numPoints: number = 0, tileSize: number = 16, options: { footprint?: Footprint; records?: "lit" | "projected" } = {}) {
    this.device = device;
    this.numPoints = numPoints;
    this.tileSize = tileSize;
    this.footprint = footprintCode(options.footprint);
    // 'disc': SequentialRenderer's oriented discs (normalsBuffer then always required)
    // records 'lit' (default, isotropic frames): the projector leaves 32-byte lit composite records (centre, radius, depth |
    // lit colour) in projector.getProjectedBuffer() and the composite gathers ONE line per staged list entry;
    // 'projected': the reference's ProjectedSplat records, colour and normal gathered per entry.  Same image.
    // (disc frames, 'lit': the lit colour rides behind each disc record inside the binner — one gathered record per staged entry)
    this.records = options.records === 'projected' ? 'projected' : 'lit';
    this.projector = new SplatProjector(device, numPoints);
    this.sorter = new RadixSorter(device, numPoints);
    this.binner = new GPUTileBinner(device, tileSize);
    this.output = null;
    this.width = 0;
    this.height = 0;
  }
  render(uniformData: Float32Array | Buffer_, propertyBuffer: Buffer_ | PropertyPlanes, normalsBuffer: Buffer_, scaleFactorsBuffer: Buffer_ | null, width: number, height: number): Buffer_ {
    this.last = [uniformData, propertyBuffer, normalsBuffer, scaleFactorsBuffer, width, height]; // (finish() may render it again)
    let u = uniformFloats(uniformData);
    if (u.length < 22) {
      const v = new Float32Array(22);
      v.set(u.subarray(0, 20));
      v[20] = width;
      v[21] = height;
      u = v;
    }
    if (this.width !== width || this.height !== height) {
      if (this.output) this.output.destroy();
      this.output = this.device.createBuffer(width * height * 4);
      this.width = width;
      this.height = height;
    }
    const small = Math.ceil(width / this.tileSize) <= 256 && Math.ceil(height / this.tileSize) <= 256;
    // what the FRAME composites from (a disc frame with 'lit': 48-byte lit disc records inside the binner) ...
    const frameFormat = this.records === 'lit' && small ? RECORDS_LIT32 : RECORDS_PROJECTED;
    // ... and what projector.getRecordsBuffer() holds after this frame — what a caller passes, with this format, to the staged
    // composite: lit composite records for an isotropic 'lit' frame, ProjectedSplat records otherwise (a disc frame's too)
    const isoLit = frameFormat === RECORDS_LIT32 && this.footprint !== FOOTPRINT_DISC;
    this.recordFormat = isoLit ? RECORDS_LIT32 : RECORDS_PROJECTED;
    this.projector.contents = isoLit ? 'lit' : 'projected';
    const cfg = [MODE_FRONT_TO_BACK, 1, this.tileSize, 0, U32_MAX, frameFormat, propertyBuffer.prelit ? 1 : 0, this.footprint];
    if (propertyBuffer.isPlanes) { // SplatPropertyManager.getPropertyPlanes()
      native.render_frame_planes(this.device.ctx, this.sorter.handle, this.binner.handle, cfg, u, propertyBuffer.posRadius.ptr, propertyBuffer.colorOpacity.ptr,
        normalsBuffer.ptr, this.numPoints, width, height, this.projector.projectedBuffer.ptr, this.output.ptr, null);
    } else {
      native.render_frame(this.device.ctx, this.sorter.handle, this.binner.handle, cfg, u,
        propertyBuffer.ptr, normalsBuffer.ptr, this.numPoints, width, height, this.projector.projectedBuffer.ptr, this.output.ptr, null);
    }
    this.binner.numTiles = Math.ceil(width / this.tileSize) * Math.ceil(height / this.tileSize);
    this.device.lastProjector = this.projector;
    this.device.lastBinner = this.binner;
    return this.output;
  }
  // Settles a sync-free frame: waits for its report (pair total, overflow and order-check flags: include/splat.h) and, if
  // the frame has to be rendered again — it outgrew the pair limit sized from the frame before it, or its tile lists
  // failed the per-tile sort's order check — does so.  Returns the frame's pair total.  Called before results are read.
  finish(): number {
    try {
      return native.bin_total(this.device.ctx, this.binner.handle);
    } catch (e) {
      if (!this.last || !/libsplat_hip -(4|8):/.test(e.message)) throw e;
      this.render(...this.last);
      return native.bin_total(this.device.ctx, this.binner.handle);
    }
  }
  readPixels(): Uint8Array {
    this.finish();
    return this.output.read(new Uint8Array(this.width * this.height * 4));
  }
  destroy(): void {
    this.projector.destroy();
    this.sorter.destroy();
    this.binner.destroy();
    if (this.output) this.output.destroy();
  }
}

/** src/Camera.ts:3-139 with gl-matrix 3.4.4 semantics (Float32Array stores, f64 arithmetic) */
class Camera {
  declare target: Float32Array;
  declare distance: number;
  declare azimuth: number;
  declare elevation: number;
  declare fov: number;
  declare aspect: number;
  declare near: number;
  declare far: number;
  declare viewProjectionMatrix: Float32Array;
  declare cameraPosition: Float32Array;
  declare isDirty: boolean;
  constructor() {
    this.target = new Float32Array([0, 0, 0]);
    this.distance = 3.0;
    this.azimuth = 0.5;
    this.elevation = 0.5;
    this.fov = 45;
    this.aspect = 1.0;
    this.near = 0.1;
    this.far = 100.0;
    this.viewProjectionMatrix = new Float32Array(16);
    this.cameraPosition = new Float32Array(3);
    this.isDirty = true;
  }
  setAspect(aspect: number): void {
    this.aspect = aspect;
    this.isDirty = true;
  }
  rotate(dAz: number, dEl: number): void {
    this.azimuth += dAz;
    this.elevation += dEl;
    const m = Math.PI / 2 - 0.01;
    this.elevation = Math.max(-m, Math.min(m, this.elevation));
    this.isDirty = true;
  }
  zoom(d: number): void {
    this.distance += d;
    this.distance = Math.max(0.5, Math.min(20.0, this.distance));
    this.isDirty = true;
  }
  pan(deltaX: number, deltaY: number): void { // :61-83 with gl-matrix's vec3 semantics (every result stored in a Float32Array)
    const f32 = (x, y, z) => new Float32Array([x, y, z]);
    const normalize = (a) => {
      let len = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
      if (len > 0) len = 1 / Math.sqrt(len);
      return f32(a[0] * len, a[1] * len, a[2] * len);
    };
    const cross = (a, b) => f32(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
    const position = this.getCameraPosition();
    const forward = normalize(f32(this.target[0] - position[0], this.target[1] - position[1], this.target[2] - position[2]));
    const right = normalize(cross(forward, f32(0, 1, 0)));
    const up = normalize(cross(right, forward));
    let offset = f32(0, 0, 0);
    offset = f32(offset[0] + right[0] * deltaX, offset[1] + right[1] * deltaX, offset[2] + right[2] * deltaX);
    // vec3.scaleAndAdd
    offset = f32(offset[0] + up[0] * deltaY, offset[1] + up[1] * deltaY, offset[2] + up[2] * deltaY);
    this.target = f32(this.target[0] + offset[0], this.target[1] + offset[1], this.target[2] + offset[2]);
    this.isDirty = true;
  }
  getCameraPosition(): Float32Array {
    const x = this.distance * Math.cos(this.elevation) * Math.sin(this.azimuth), y = this.distance * Math.sin(this.elevation), z = this.distance * Math.cos(this.elevation) * Math.cos(this.azimuth);
    return new Float32Array([this.target[0] + x, this.target[1] + y, this.target[2] + z]);
  }
  updateMatrices(): void {
    if (!this.isDirty) return;
    const eye = this.getCameraPosition();
    this.cameraPosition = eye;
    const view = new Float32Array(16), proj = new Float32Array(16);
    let z0 = eye[0] - this.target[0], z1 = eye[1] - this.target[1], z2 = eye[2] - this.target[2];
    if (Math.abs(z0) < 1e-6 && Math.abs(z1) < 1e-6 && Math.abs(z2) < 1e-6) { view[0] = view[5] = view[10] = view[15] = 1; } else {
      let len = 1 / Math.sqrt(z0 * z0 + z1 * z1 + z2 * z2);
      z0 *= len;
      z1 *= len;
      z2 *= len;
      let x0 = 1 * z2 - 0 * z1, x1 = 0 * z0 - 0 * z2, x2 = 0 * z1 - 1 * z0;
      // up = (0,1,0)
      len = Math.sqrt(x0 * x0 + x1 * x1 + x2 * x2);
      if (!len) { x0 = x1 = x2 = 0; } else { len = 1 / len; x0 *= len; x1 *= len; x2 *= len; }
      let y0 = z1 * x2 - z2 * x1, y1 = z2 * x0 - z0 * x2, y2 = z0 * x1 - z1 * x0;
      len = Math.sqrt(y0 * y0 + y1 * y1 + y2 * y2);
      if (!len) { y0 = y1 = y2 = 0; } else { len = 1 / len; y0 *= len; y1 *= len; y2 *= len; }
      view.set([x0, y0, z0, 0, x1, y1, z1, 0, x2, y2, z2, 0, -(x0 * eye[0] + x1 * eye[1] + x2 * eye[2]), -(y0 * eye[0] + y1 * eye[1] + y2 * eye[2]), -(z0 * eye[0] + z1 * eye[1] + z2 * eye[2]), 1]);
    }
    const f = 1 / Math.tan(((this.fov * Math.PI) / 180) / 2), nf = 1 / (this.near - this.far);
    proj[0] = f / this.aspect;
    proj[5] = f;
    proj[11] = -1;
    proj[10] = (this.far + this.near) * nf;
    proj[14] = 2 * this.far * this.near * nf;
    const out = this.viewProjectionMatrix;
    for (let c = 0; c < 4; c++) for (let k = 0; k < 4; k++) out[c * 4 + k] = view[c * 4] * proj[k] + view[c * 4 + 1] * proj[4 + k] + view[c * 4 + 2] * proj[8 + k] + view[c * 4 + 3] * proj[12 + k];
    this.isDirty = false;
  }
  getViewProjectionMatrix(): Float32Array {
    this.updateMatrices();
    return this.viewProjectionMatrix;
  }
  getPosition(): Float32Array {
    this.updateMatrices();
    return this.cameraPosition;
  }
  uniforms(width: number, height: number, time: number = 0): Float32Array {
    const u = new Float32Array(22);
    u.set(this.getViewProjectionMatrix(), 0);
    u.set(this.getPosition(), 16);
    u[19] = time;
    u[20] = width;
    u[21] = height;
    return u;
  }
}

/** src/PointManager.ts:41,220-252 — ping-pong position buffers.  The reference seeds points with unseeded Math.random on an
 * SDF surface (out of scope: upstream splat generation); here `scene` is a Float32Array of vec4 positions, or
 * {numPoints, seed} for a deterministic uniform cloud in [-1,1]^3, and reinitialize() uploads it again. */
// src/sdf/Primitive.ts:283-290 AS WRITTEN: centre = min + max / 2 (vec3.scaleAndAdd(_, min, max, 1 / 2)), not the midpoint — kept, so
// that seeding boxes equal the reference's; plain arrays, i.e. double precision (held to an execution of the reference's
// statements: tests/golden/ref_host.json)
function scaleAABB(aabb: [ArrayLike<number>, ArrayLike<number>], scale: number): [Float32Array, Float32Array] {
  const min = [0, 0, 0];
  const max = [0, 0, 0];
  for (let k = 0; k < 3; k++) {
    const center = aabb.min[k] + aabb.max[k] * (1 / 2);
    const currentScale = aabb.max[k] - aabb.min[k];
    min[k] = center + currentScale * (-scale / 2);
    max[k] = center + currentScale * (scale / 2);
  }
  return { min, max };
}

class PointManager {
  declare device: Device;
  declare scene: SDFScene | null;
  declare seed: number;
  declare positions: Float32Array | null;
  declare numPoints: number;
  declare buffers: Buffer_[];
  declare current: number;
  // (device, Float32Array of vec4 positions) | (device, { numPoints, seed }) | (device, SDFScene[, seed]): the reference's constructor —
  // point count from the primitives' surface areas (src/PointManager.ts:22-39), a fresh cloud on the faces of the scene's scaled box
  // at every reinitialize() (:96-189, :220-231), drawn on the device (native.sdf_seed_positions: point i a pure function of (seed, i))
  constructor(device: Device, scene: Float32Array | { numPoints: number; seed?: number } | SDFScene, seed: number = 0) {
    this.device = device;
    this.scene = null;
    this.seed = seed;
    if (scene instanceof Float32Array) {
      this.positions = scene;
      this.numPoints = scene.length / 4;
    } else if (scene && typeof scene.getPrimitives === 'function') {
      const prims = scene.getPrimitives();
      if (!prims.length) throw new Error('Scene must have at least one primitive');
      // :47-49
      this.scene = scene;
      this.numPoints = PointManager.calculatePointCount(scene); // :22-39
    } else {
      const n = scene.numPoints;
      let state = (scene.seed === undefined ? 1 : scene.seed) >>> 0 || 1;
      const next = () => {
        state ^= state << 13;
        state >>>= 0;
        state ^= state >>> 17;
        state ^= state << 5;
        state >>>= 0;
        return state / 4294967296;
      };
      // xorshift32
      this.positions = new Float32Array(n * 4);
      for (let i = 0; i < n; i++) {
        this.positions[i * 4] = next() * 2 - 1;
        this.positions[i * 4 + 1] = next() * 2 - 1;
        this.positions[i * 4 + 2] = next() * 2 - 1;
        this.positions[i * 4 + 3] = 1;
      }
      this.numPoints = n;
    }
    this.buffers = [device.createBuffer(this.numPoints * 16), device.createBuffer(this.numPoints * 16)];
    this.current = 0;
    this.reinitialize();
  }
  // the box PointManager seeds on (:96-107): the primitives' AABBs merged, scaled 1.5x by the reference's scaleAABB AS WRITTEN
  // (centre = min + max / 2: src/sdf/Primitive.ts:283-290), in double precision, then rounded to float32
  seedingBox(): [Float32Array, Float32Array] {
    const mn = [Infinity, Infinity, Infinity], mx = [-Infinity, -Infinity, -Infinity];
    for (const p of this.scene.getPrimitives()) {
      const [a, b] = p.getAABB();
      for (let k = 0; k < 3; k++) {
        mn[k] = Math.min(mn[k], a[k]);
        mx[k] = Math.max(mx[k], b[k]);
      }
    }
    const scaled = scaleAABB({ min: mn, max: mx }, 1.5);
    return [Float32Array.from(scaled.min), Float32Array.from(scaled.max)];
  }
  // :22-39: floor(30000 sqrt(area)) per primitive, clamped to [10000, 200000]; 50000 for a scene without primitives
  static calculatePointCount(scene: SDFScene): number {
    const prims = scene.getPrimitives();
    if (prims.length === 0) return 50000;
    const total = prims.reduce((t, p) => t + Math.floor(30000 * Math.sqrt(p.getSurfaceArea())), 0);
    return Math.max(10000, Math.min(total, 200000));
  }
  reinitialize(): void { // :220-231
    if (this.scene) {
      const [lo, hi] = this.seedingBox();
      native.sdf_seed_positions(this.device.ctx, lo, hi, this.numPoints, this.seed++, this.buffers[this.current].ptr);
      return;
    }
    this.buffers[this.current].write(this.positions);
  }
  getCurrentPositionBuffer(): Buffer_ { return this.buffers[this.current]; }     // :233-235
  getNextPositionBuffer(): Buffer_ { return this.buffers[1 - this.current]; }    // :236-238
  swap(): void { this.current = 1 - this.current; }                           // :240-242
  getNumPoints(): number { return this.numPoints; }                             // :244-246
  destroy(): void { this.buffers.forEach((b) => b.destroy()); }              // :248-252
}

/** SDF splat generation (src/sdf/{Primitive,Operation,Scene}.ts, src/GradientSampler.ts, src/PositionUpdater.ts,
 * src/CurvatureSampler.ts).  The reference generates a WGSL function per scene graph (sdf/CodeGenerator.ts); here the graph
 * becomes a postfix program — children first, then their operation, the order CodeGenerator's traverse() emits — that a
 * stack machine in the HIP kernels evaluates: updateSceneParameters() re-encodes it, nothing is ever recompiled. */
const SDF = { sphere: 0, box: 1, torus: 2, capsule: 3, union: 16, intersection: 17, subtraction: 18, smooth_union: 19 };
let nextPrimId = 0, nextSminId = 0;
// a primitive's box (src/sdf/Primitive.ts: getAABB): position -/+ extent in double precision, rounded to float32 like the reference's vec3
const aabb = (p, e) => [Float32Array.from([p[0] - e[0], p[1] - e[1], p[2] - e[2]]), Float32Array.from([p[0] + e[0], p[1] + e[1], p[2] + e[2]])];
class Primitive {
  declare id: string;
  declare position: Float32Array;
  constructor(id: string, position: ArrayLike<number>) {
    this.id = id || `prim_${nextPrimId++}`;
    this.position = Float32Array.from(position || [0, 0, 0]);
  }
}
class Sphere extends Primitive {
  declare radius: number;
  constructor(p: { id?: string; position?: ArrayLike<number>; radius?: number } = {}) {
    super(p.id, p.position);
    this.radius = p.radius === undefined ? 0.5 : p.radius;
  }
  getType(): string { return 'sphere'; }
  getParamNames(): string[] { return [`${this.id}_center`, `${this.id}_radius`]; }
  getParamValues(): number[] { return [...this.position, this.radius]; }
  getSurfaceArea(): number { return 4 * Math.PI * this.radius * this.radius; }
  instr(): number[] { return [SDF.sphere, ...this.position, this.radius]; }
  getAABB(): [Float32Array, Float32Array] { return aabb(this.position, [this.radius, this.radius, this.radius]); }
}
class Box extends Primitive {
  declare size: Float32Array;
  constructor(p: { id?: string; position?: ArrayLike<number>; size?: ArrayLike<number> } = {}) {
    super(p.id, p.position);
    this.size = Float32Array.from(p.size || [0.5, 0.5, 0.5]);
  }
  getType(): string { return 'box'; }
  getParamNames(): string[] { return [`${this.id}_center`, `${this.id}_size`]; }
  getParamValues(): number[] { return [...this.position, 0, ...this.size, 0]; }
  getSurfaceArea(): number {
    const w = this.size[0] * 2, h = this.size[1] * 2, d = this.size[2] * 2;
    return 2 * (w * h + w * d + h * d);
  }
  instr(): number[] { return [SDF.box, ...this.position, ...this.size]; }
  getAABB(): [Float32Array, Float32Array] { return aabb(this.position, this.size); }
}
class Torus extends Primitive {
  declare majorRadius: number;
  declare minorRadius: number;
  constructor(p: { id?: string; position?: ArrayLike<number>; majorRadius?: number; minorRadius?: number } = {}) {
    super(p.id, p.position);
    this.majorRadius = p.majorRadius === undefined ? 0.5 : p.majorRadius;
    this.minorRadius = p.minorRadius === undefined ? 0.2 : p.minorRadius;
  }
  getType(): string { return 'torus'; }
  getParamNames(): string[] { return [`${this.id}_center`, `${this.id}_radii`]; }
  getParamValues(): number[] { return [...this.position, 0, this.majorRadius, this.minorRadius, 0, 0]; }
  getSurfaceArea(): number { return 4 * Math.PI * Math.PI * this.majorRadius * this.minorRadius; }
  instr(): number[] { return [SDF.torus, ...this.position, this.majorRadius, this.minorRadius]; }
  getAABB(): [Float32Array, Float32Array] { return aabb(this.position, [this.majorRadius + this.minorRadius, this.minorRadius, this.majorRadius + this.minorRadius]); }
}
class Capsule extends Primitive {
  declare height: number;
  declare radius: number;
  constructor(p: { id?: string; position?: ArrayLike<number>; height?: number; radius?: number } = {}) {
    super(p.id, p.position);
    this.height = p.height === undefined ? 1.0 : p.height;
    this.radius = p.radius === undefined ? 0.3 : p.radius;
  }
  getType(): string { return 'capsule'; }
  getParamNames(): string[] { return [`${this.id}_center`, `${this.id}_params`]; }
  getParamValues(): number[] { return [...this.position, 0, this.height, this.radius, 0, 0]; }
  getSurfaceArea(): number { return 2 * Math.PI * this.radius * this.height + 4 * Math.PI * this.radius * this.radius; }
  instr(): number[] { return [SDF.capsule, ...this.position, this.height, this.radius]; }
  getAABB(): [Float32Array, Float32Array] { return aabb(this.position, [this.radius, this.height / 2 + this.radius, this.radius]); }
}
class Operation {
  declare type: string;
  declare params: number[];
  constructor(type: string, params: number[] = []) {
    this.type = type;
    this.params = params;
  }
  getType(): string { return this.type; }
  getParamNames(): string[] { return []; }
  getParamValues(): number[] { return this.params; }
}
class SmoothUnion extends Operation {
  declare k: number;
  declare id: string;
  constructor(k: number = 0.1) {
    super('smooth_union', [k]);
    this.k = k;
    this.id = `smin_${nextSminId++}`;
  }
  getParamNames(): string[] { return [`${this.id}_k`]; }
  getParamValues(): number[] { return [this.k]; }
}
const primitive = (p) => (p && p.type === 'primitive') || (p && p.type === 'operation') ? p : { type: 'primitive', primitive: p };
const binary = (op, a, b) => ({ type: 'operation', operation: op, children: [primitive(a), primitive(b)] });
const union = (a, b) => binary(new Operation('union'), a, b), intersection = (a, b) => binary(new Operation('intersection'), a, b);
const subtraction = (a, b) => binary(new Operation('subtraction'), a, b), smoothUnion = (k, a, b) => binary(new SmoothUnion(k), a, b);
class SDFScene { // src/sdf/Scene.ts:72-152
  declare root: SceneNode | null;
  declare primitiveMap: Map<string, Primitive>;
  constructor() {
    this.root = null;
    this.primitiveMap = new Map();
  }
  setRoot(node: Primitive | SceneNode): void {
    this.root = primitive(node);
    this.primitiveMap.clear();
    const walk = (n) => {
      if (n.type === 'primitive') this.primitiveMap.set(n.primitive.id, n.primitive);
      else n.children.forEach(walk);
    };
    walk(this.root);
  }
  get(id: string): Primitive | undefined { return this.primitiveMap.get(id); }
  getPrimitives(): Primitive[] { return Array.from(this.primitiveMap.values()); }
  getRoot(): SceneNode | null { return this.root; }
  getOperations(): unknown[] {
    const ops = [];
    const walk = (n) => {
      if (n.type === 'operation') {
        ops.push(n.operation);
        n.children.forEach(walk);
      }
    };
    if (this.root) walk(this.root);
    return ops;
  }
  getStructureHash(): string {
    const walk = (n) => (n.type === 'primitive' ? `P:${n.primitive.getType()}:${n.primitive.id}` : `O:${n.operation.getType()}:(${n.children.map(walk).join(',')})`);
    return this.root ? walk(this.root) : '';
  }
  program(): Float32Array { // Float32Array, 8 floats per instruction: [op, a0..a6]
    const rows = [];
    const walk = (n) => {
      if (n.type === 'primitive') rows.push(n.primitive.instr());
      else {
        n.children.forEach(walk);
        rows.push([SDF[n.operation.getType()], ...n.operation.getParamValues()]);
      }
    };
    if (this.root) walk(this.root);
    const out = new Float32Array(rows.length * 8);
    rows.forEach((r, k) => out.set(r, k * 8));
    return out;
  }
}
class SceneStage {
  declare device: Device;
  declare scene: SDFScene;
  declare numPoints: number;
  declare currentStructureHash: string;
  declare program: Float32Array;
  constructor(device: Device, scene: SDFScene, numPoints: number) {
    this.device = device;
    this.scene = scene;
    this.numPoints = numPoints;
    this.currentStructureHash = scene.getStructureHash();
    this.updateSceneParameters();
  }
  updateSceneParameters(): void { this.program = this.scene.program(); }
  rebuildIfNeeded(): void {
    const h = this.scene.getStructureHash();
    if (h !== this.currentStructureHash) {
      this.currentStructureHash = h;
      this.updateSceneParameters();
    }
  }
  getScene(): SDFScene { return this.scene; }
}
class GradientSampler extends SceneStage { // src/GradientSampler.ts
  declare gradientBuffer: Buffer_ | null;
  constructor(device: Device, scene: SDFScene, numPoints: number) {
    super(device, scene, numPoints);
    this.gradientBuffer = device.createBuffer(numPoints * 16);
  }
  evaluateGradients(commandEncoder: CommandEncoder | null, uniformBuffer: Buffer_ | null, positionBuffer: Buffer_): void { native.sdf_gradients(this.device.ctx, this.program, positionBuffer.ptr, this.numPoints, this.gradientBuffer.ptr); }
  getGradientBuffer(): Buffer_ { return this.gradientBuffer; }
  destroy(): void { this.gradientBuffer.destroy(); }
}
class PositionUpdater { // src/PositionUpdater.ts
  declare device: Device;
  declare numPoints: number;
  constructor(device: Device, shaderCode: string | null, numPoints: number) {
    this.device = device;
    this.numPoints = numPoints;
  }
  updatePositions(commandEncoder: CommandEncoder | null, uniformBuffer: Buffer_ | null, currentPositionBuffer: Buffer_, gradientBuffer: Buffer_, nextPositionBuffer: Buffer_): void { native.sdf_update_positions(this.device.ctx, currentPositionBuffer.ptr, gradientBuffer.ptr, this.numPoints, nextPositionBuffer.ptr); }
}
class CurvatureSampler extends SceneStage { // src/CurvatureSampler.ts
  declare scaleFactorsBuffer: Buffer_ | null;
  declare curvatureBuffer: Buffer_ | null;
  constructor(device: Device, scene: SDFScene, numPoints: number) {
    super(device, scene, numPoints);
    this.scaleFactorsBuffer = device.createBuffer(numPoints * 4);
    this.curvatureBuffer = null;
  }
  computeScaleFactors(commandEncoder: CommandEncoder | null, positionBuffer: Buffer_): void { native.sdf_scale_factors(this.device.ctx, this.program, positionBuffer.ptr, this.numPoints, this.scaleFactorsBuffer.ptr); }
  getScaleFactorsBuffer(): Buffer_ { return this.scaleFactorsBuffer; }
  // vec4(normal, scaleFactor): the curvatureData buffer SplatPropertyManager.updateFromCurvature binds (the reference's samplers write its halves apart)
  getCurvatureBuffer(gradientBuffer: Buffer_): Buffer_ {
    if (!this.curvatureBuffer) this.curvatureBuffer = this.device.createBuffer(this.numPoints * 16);
    native.sdf_curvature(this.device.ctx, gradientBuffer.ptr, this.scaleFactorsBuffer.ptr, this.numPoints, this.curvatureBuffer.ptr);
    return this.curvatureBuffer;
  }
  destroy(): void {
    this.scaleFactorsBuffer.destroy();
    if (this.curvatureBuffer) this.curvatureBuffer.destroy();
  }
}

/** src/OrbitCameraController.ts:3-75 — the same speeds and the same mapping of mouse buttons and the wheel onto
 * Camera.rotate / pan / zoom.  `canvas` is anything with addEventListener(type, handler) (a DOM canvas, a Node
 * EventEmitter adapter) or null: without one, feed the handlers synthetic events {clientX, clientY, button, deltaY}. */
class OrbitCameraController {
  declare camera: Camera;
  declare canvas: unknown; // (headless: whatever the caller passes for the reference's HTMLCanvasElement)
  declare isDragging: boolean;
  declare dragButton: number;
  declare lastMouseX: number;
  declare lastMouseY: number;
  declare rotationSpeed: number;
  declare panSpeed: number;
  declare zoomSpeed: number;
  constructor(camera: Camera, canvas: { addEventListener(type: string, handler: (e: PointerLikeEvent) => void, options?: unknown): void } | null = null) {
    this.camera = camera;
    this.canvas = canvas;
    this.isDragging = false; // :7
    this.dragButton = -1; // :8
    this.lastMouseX = 0; // :9
    this.lastMouseY = 0; // :10
    this.rotationSpeed = 0.005; // :12
    this.panSpeed = 0.002; // :13
    this.zoomSpeed = 0.001; // :14
    this.setupEventListeners();
  }
  setupEventListeners(): void { // :23-33
    if (!this.canvas || typeof this.canvas.addEventListener !== 'function') return;
    this.canvas.addEventListener('mousedown', this.onMouseDown.bind(this));
    this.canvas.addEventListener('mousemove', this.onMouseMove.bind(this));
    this.canvas.addEventListener('mouseup', this.onMouseUp.bind(this));
    this.canvas.addEventListener('wheel', this.onWheel.bind(this), { passive: false });
    this.canvas.addEventListener('contextmenu', (e) => e.preventDefault());
  }
  onMouseDown(event: PointerLikeEvent): void { // :35-40
    this.isDragging = true;
    this.dragButton = event.button;
    this.lastMouseX = event.clientX;
    this.lastMouseY = event.clientY;
  }
  onMouseMove(event: PointerLikeEvent): void { // :42-58
    if (!this.isDragging) return;
    const dx = event.clientX - this.lastMouseX;
    const dy = event.clientY - this.lastMouseY;
    if (this.dragButton === 0) { // left button: rotate
      this.camera.rotate(dx * this.rotationSpeed, -dy * this.rotationSpeed);
    } else if (this.dragButton === 1 || this.dragButton === 2) { // middle or right button: pan
      this.camera.pan(-dx * this.panSpeed, dy * this.panSpeed);
    }
    this.lastMouseX = event.clientX;
    this.lastMouseY = event.clientY;
  }
  onMouseUp(_event?: PointerLikeEvent): void { // :60-63
    this.isDragging = false;
    this.dragButton = -1;
  }
  onWheel(event: PointerLikeEvent): void { // :65-70
    if (event.preventDefault) event.preventDefault();
    const delta = event.deltaY * this.zoomSpeed;
    this.camera.zoom(delta);
  }
  destroy(): void {} // :72-74
}

/** The render loop of src/main.ts:110-193 for the tile-raster path, without a browser: per frame the camera's uniform
 * block (VP, eye, time, W, H — :126-144) and one Renderer.render call (:183-190).  Frames are enqueued back to back
 * (sync-free after the first); a frame's pixels are read only when asked for.  splat_renderer_amd/frameloop.py is the
 * same loop in Python: the two give the same images byte for byte (tests/test_napi.py). */
class FrameLoop {
  declare device: Device;
  declare width: number;
  declare height: number;
  declare camera: Camera;
  declare renderer: Renderer;
  declare frame: number;
  constructor(device: Device, numPoints: number, width: number, height: number, tileSize: number = 16, camera: Camera | null = null, rendererOptions: { footprint?: Footprint; records?: "lit" | "projected" } = {}) {
    this.device = device;
    this.width = width;
    this.height = height;
    this.camera = camera || new Camera();
    this.camera.setAspect(width / height); // resizeCanvas, main.ts:97-101
    this.renderer = new Renderer(device, null, 'rgba8unorm', numPoints, tileSize, rendererOptions);
    this.frame = 0;
  }
  // one frame with the camera as it stands; returns the output buffer (pixels stay on the device)
  render(propertyBuffer: Buffer_ | PropertyPlanes, normalsBuffer: Buffer_, time?: number): Buffer_ {
    const t = time === undefined ? this.frame / 60.0 : time;
    const out = this.renderer.render(this.camera.uniforms(this.width, this.height, t), propertyBuffer, normalsBuffer, null, this.width, this.height);
    this.frame += 1;
    return out;
  }
  readPixels(): Uint8Array { return this.renderer.readPixels(); }
  // `frames` frames of a full orbit (Camera.rotate by 2 pi / frames after each); onFrame(k, rgba8) gets every frame's pixels
  turntable(propertyBuffer: Buffer_ | PropertyPlanes, normalsBuffer: Buffer_, frames: number, onFrame?: (k: number, rgba8: Uint8Array) => void): void {
    for (let k = 0; k < frames; k++) {
      this.render(propertyBuffer, normalsBuffer);
      if (onFrame) onFrame(k, this.readPixels());
      this.camera.rotate((2.0 * Math.PI) / frames, 0.0);
    }
  }
  destroy(): void { this.renderer.destroy(); }
}

/** The multi-GPU frame's exchange (no reference counterpart: the reference is single-device): one process per GPU, an
 * RCCL communicator behind the C ABI.  Rank 0 calls Comm.uniqueId() and hands the 128 bytes to the other ranks by
 * any channel (a file, a socket, an environment variable); every rank then constructs Comm with the same bytes. */
class Comm {
  declare device: Device;
  declare rank: number;
  declare world: number;
  declare handle: unknown;
  static uniqueId(): Uint8Array { return new Uint8Array(native.comm_unique_id()); }
  constructor(device: Device, rank: number, world: number, idBytes: Uint8Array) {
    this.device = device;
    this.rank = rank;
    this.world = world;
    this.handle = native.comm_init(device.ctx, rank, world, idBytes);
  }
  allGather(shardBuffer: Buffer_, gatheredBuffer: Buffer_, bytesPerRank: number): void { native.allgather_records(this.device.ctx, this.handle, shardBuffer.ptr, gatheredBuffer.ptr, bytesPerRank); }
  destroy(): void {
    if (this.handle) native.comm_destroy(this.handle);
    this.handle = null;
  }
}

/** One rank of north_star's multi-GPU frame (SURVEY §8e): project my 1/world of the splats into 16-byte exchange records
 * {centre x, y, radius, depth}, ONE all-gather, then my band of tile rows binned, depth-sorted and composited from the
 * gathered records.  render() returns the full-size image buffer of which this rank owns pixel rows pixelRows().
 * Everything is enqueued on the device's stream: no host synchronisation inside a frame. */
class BandRenderer {
  declare device: Device;
  declare comm: Comm | null;
  declare numPoints: number;
  declare width: number;
  declare height: number;
  declare tileSize: number;
  declare per: number;
  declare first: number;
  declare count: number;
  declare row0: number;
  declare row1: number;
  declare sorter: RadixSorter;
  declare binner: GPUTileBinner;
  declare shard: Buffer_;
  declare gathered: Buffer_;
  declare output: Buffer_ | null;
  constructor(device: Device, comm: Comm | null, numPoints: number, width: number, height: number, tileSize: number = 16) {
    this.device = device;
    this.comm = comm;
    this.numPoints = numPoints;
    this.width = width;
    this.height = height;
    this.tileSize = tileSize;
    const world = comm ? comm.world : 1, rank = comm ? comm.rank : 0;
    this.per = Math.ceil(numPoints / world);
    this.first = Math.min(rank * this.per, numPoints);
    this.count = Math.min(this.per, numPoints - this.first);
    const nty = Math.ceil(height / tileSize);
    this.row0 = Math.floor(nty * rank / world);
    this.row1 = Math.floor(nty * (rank + 1) / world);
    this.sorter = new RadixSorter(device, this.per * world);
    this.binner = new GPUTileBinner(device, tileSize);
    this.shard = device.createBuffer(this.per * 16);
    this.shard.write(new Float32Array(this.per * 4).fill(NaN));
    // padding records bin nowhere
    this.gathered = world > 1 ? device.createBuffer(this.per * world * 16) : this.shard;
    this.output = device.createBuffer(width * height * 4);
    this.output.zero();
  }
  render(uniformData: Float32Array | Buffer_, propertyBuffer: Buffer_, normalsBuffer: Buffer_): Buffer_ {
    const u = uniformFloats(uniformData), d = this.device, world = this.comm ? this.comm.world : 1;
    native.project_slice_compact(d.ctx, u, propertyBuffer.ptr, 2, this.first, this.count, this.shard.ptr);
    if (world > 1) this.comm.allGather(this.shard, this.gathered, this.per * 16);
    const cfg = [MODE_FRONT_TO_BACK, 1, this.tileSize, this.row0, this.row1, RECORDS_COMPACT, 0, FOOTPRINT_ISOTROPIC];
    native.band_frame(d.ctx, this.sorter.handle, this.binner.handle, cfg, propertyBuffer.ptr, normalsBuffer.ptr, this.gathered.ptr, this.per * world,
      this.width, this.height, this.output.ptr, null);
    return this.output;
  }
  settle(): number { return native.band_settle(this.device.ctx, this.sorter.handle, this.binner.handle); }
  pixelRows(): [number, number] { return [this.row0 * this.tileSize, Math.min(this.row1 * this.tileSize, this.height)]; }
  readPixels(): Uint8Array {
    this.settle();
    return this.output.read(new Uint8Array(this.width * this.height * 4));
  }
  destroy(): void {
    this.sorter.destroy();
    this.binner.destroy();
    this.shard.destroy();
    if (this.gathered !== this.shard) this.gathered.destroy();
    this.output.destroy();
  }
}

module.exports = { native, Device, Buffer: Buffer_, Camera, OrbitCameraController, FrameLoop, PointManager, scaleAABB, Comm, BandRenderer, SDFScene, Sphere, Box, Torus, Capsule, SmoothUnion,
  union, intersection, subtraction, smoothUnion, GradientSampler, PositionUpdater, CurvatureSampler, SplatPropertyManager, SplatProjector, DepthKeyExtractor, RadixSorter, PrefixSumScanner,
  GPUTileBinner, PerTileSorter, ComputeShaderRenderer, TileRenderer, SequentialRenderer, Renderer, MODE_FRONT_TO_BACK, MODE_REFERENCE_LITERAL,
  FOOTPRINT_ISOTROPIC, FOOTPRINT_DISC, RECORDS_PROJECTED, RECORDS_COMPACT, RECORDS_LIT32 };
