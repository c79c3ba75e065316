'use strict';
// node render_frame.js <props.f32> <normals.f32> <n> <W> <H> <out.rgba8> [<order.u32> <counts.u32> <indices.u32> [<frame.rgba8> [<seq.rgba8> <discframe.rgba8>]]]
// Renders one frame through the JS host classes (stage by stage, like the reference's call order)
// and writes the raw outputs for tests/test_napi.py to compare with the oracle.
const fs = require('fs');
const sr = require('./index.js');
const [propsPath, normalsPath, nStr, wStr, hStr, outPath, orderPath, countsPath, indicesPath, framePath, seqPath, discFramePath] = process.argv.slice(2);
const n = +nStr, W = +wStr, H = +hStr;
const f32 = (p) => { const b = fs.readFileSync(p); return new Float32Array(b.buffer, b.byteOffset, b.length / 4); };
const device = new sr.Device(0);
const props = new sr.SplatPropertyManager(device, n); props.setFromArrays(f32(propsPath));
const normals = device.createBufferFrom(f32(normalsPath));
const camera = new sr.Camera(); camera.setAspect(W / H);
const uniforms = camera.uniforms(W, H);
const projector = new sr.SplatProjector(device, n), sorter = new sr.RadixSorter(device, n), extractor = new sr.DepthKeyExtractor(device);
const binner = new sr.GPUTileBinner(device, 16), renderer = new sr.ComputeShaderRenderer(device, null, 'rgba8unorm');
let threw = false;
try { binner.getTileOffsetsBuffer(); } catch (e) { threw = /not initialized/.test(e.message); }
if (!threw) throw new Error('getter before binSplats did not throw');
(async () => {
  const enc = device.createCommandEncoder();
  projector.project(enc, uniforms, props.getPropertyBuffer());
  extractor.extract(enc, projector.getProjectedBuffer(), sorter.getKeysBuffer(), sorter.getPayloadBuffer(), n, sorter.paddedSize);
  sorter.sort();
  await binner.binSplats(enc, projector.getProjectedBuffer(), sorter.getSortedIndicesBuffer(), n, W, H);
  const checker = new sr.PerTileSorter(device, true);
  const bad = checker.sort(enc, projector.getProjectedBuffer(), binner.getTileCountsBuffer(), binner.getTileOffsetsBuffer(),
    binner.getTileIndicesBuffer(), binner.numTiles, 4096, binner.getTotalIndices());
  if (bad !== 0) throw new Error('PerTileSorter validator found ' + bad + ' out-of-order neighbours');
  renderer.render(uniforms, props.getPropertyBuffer(), binner.getTileIndicesBuffer(), normals, projector.getProjectedBuffer(),
    binner.getTileCountsBuffer(), binner.getTileOffsetsBuffer(), 16, Math.ceil(W / 16), W, H);
  const stagedPixels = renderer.readPixels();
  fs.writeFileSync(outPath, Buffer.from(stagedPixels.buffer));
  // TileRenderer.render with exactly the reference's eleven arguments (src/TileRenderer.ts:234-246; tileCountsData a host
  // Uint32Array) and no bindTileData: it composites from the device's last projector and binner — the bytes of the staged composite
  const tileRenderer = new sr.TileRenderer(device, null, 'rgba8unorm');
  const tileCountsData = binner.getTileCountsBuffer().read(new Uint32Array(binner.numTiles));
  await tileRenderer.render(uniforms, props.getPropertyBuffer(), binner.getTileIndicesBuffer(), normals, tileCountsData, Math.ceil(W / 16), Math.ceil(H / 16), 16, 4096, W, H);
  const tilePixels = tileRenderer.readPixels();
  const tileRendererEqualsStaged = tilePixels.length === stagedPixels.length && tilePixels.every((v, i) => v === stagedPixels[i]);
  tileRenderer.destroy();
  if (orderPath) fs.writeFileSync(orderPath, Buffer.from(sorter.getSortedIndicesBuffer().read(new Uint32Array(n)).buffer));
  if (countsPath) fs.writeFileSync(countsPath, Buffer.from(binner.getTileCountsBuffer().read(new Uint32Array(binner.numTiles)).buffer));
  if (indicesPath) fs.writeFileSync(indicesPath, Buffer.from(binner.getTileIndicesBuffer().read(new Uint32Array(binner.getTotalIndices())).buffer));
  let framePairs = -1, recordFormat = -1, schedulesKeepTheBytes = null, kernelsWithinOneLsb = null, timed = null, bandEqualsFrame = null, bandPairs = -1, pointManagerOk = null;
  if (framePath) { // the whole-frame facade (tile-first order inside), fed the native two-plane property layout, twice (2nd: sync-free)
    const whole = new sr.Renderer(device, null, 'rgba8unorm', n, 16);
    whole.binner.setFrameOrder('tileFirst');
    for (let k = 0; k < 2; k++) whole.render(uniforms, props.getPropertyPlanes(), normals, null, W, H);
    const framePixels = whole.readPixels();
    fs.writeFileSync(framePath, Buffer.from(framePixels.buffer));
    framePairs = whole.binner.getTotalIndices();
    recordFormat = whole.recordFormat;
    // the composite's per-context options and the timing detail from JS: the per-pixel-queue kernel under two schedules (one chunk
    // of look-ahead bounded by history; two chunks, no history) gives the same bytes, and the default kernel's
    // image within 1 LSB; one composite launch timed, its entries counted
    device.compositeOptions('pixel', 1, true);
    for (let k = 0; k < 3; k++) whole.render(uniforms, props.getPropertyPlanes(), normals, null, W, H);
    const pixelA = whole.readPixels();
    device.compositeOptions('pixel', 2, false);
    device.forgetCompositeHistory();
    device.setTiming(true, (1 << 3) + 0x80000000, 1); // the bit of SPLAT_STAGE_COMPOSITE + SPLAT_TIMING_COUNT_ENTRIES
    whole.render(uniforms, props.getPropertyPlanes(), normals, null, W, H);
    const pixelB = whole.readPixels();
    schedulesKeepTheBytes = pixelB.length === pixelA.length && pixelB.every((v, i) => v === pixelA[i]);
    kernelsWithinOneLsb = pixelA.length === framePixels.length && pixelA.every((v, i) => Math.abs(v - framePixels[i]) <= 1);
    timed = Object.assign(device.stageTimeStats(3), device.timingConsumed());
    device.setTiming(false);
    device.compositeOptions(null, 0, null);
    // north_star's multi-GPU frame from JS, on the one GPU there is: a one-rank RCCL communicator behind the C ABI,
    // project my slice -> all-gather -> my band (= every tile row) from the gathered 16-byte records
    const comm = new sr.Comm(device, 0, 1, sr.Comm.uniqueId());
    const band = new sr.BandRenderer(device, comm, n, W, H, 16);
    for (let k = 0; k < 2; k++) band.render(uniforms, props.getPropertyBuffer(), normals);
    const bandPixels = band.readPixels();
    bandPairs = band.settle();
    bandEqualsFrame = bandPixels.length === framePixels.length && bandPixels.every((v, i) => v === framePixels[i]);
    // a real exchange through the communicator: gather a buffer onto a second one
    const probe = new Float32Array(1024).map((_, i) => i * 0.5), a = device.createBufferFrom(probe), b = device.createBuffer(4096);
    comm.allGather(a, b, 4096);
    const back = b.read(new Float32Array(1024));
    if (!back.every((v, i) => v === probe[i])) throw new Error('Comm.allGather did not deliver the shard');
    band.destroy(); comm.destroy(); a.destroy(); b.destroy();
    // PointManager: ping-pong position buffers
    const pos = new Float32Array(n * 4); for (let i = 0; i < n; i++) { pos[i * 4] = i; pos[i * 4 + 3] = 1; }
    const pm = new sr.PointManager(device, pos);
    const cur = pm.getCurrentPositionBuffer(), nxt = pm.getNextPositionBuffer();
    pm.swap();
    const seeded = new sr.PointManager(device, { numPoints: 100, seed: 7 });
    const sp = seeded.getCurrentPositionBuffer().read(new Float32Array(400));
    pointManagerOk = pm.getNumPoints() === n && pm.getCurrentPositionBuffer() === nxt && pm.getNextPositionBuffer() === cur &&
      cur.read(new Float32Array(n * 4)).every((v, i) => v === pos[i]) && sp.every((v, i) => (i % 4 === 3 ? v === 1 : v >= -1 && v <= 1));
    pm.destroy(); seeded.destroy();
  }
  let seqPairs = -1, discFramePairs = -1;
  if (seqPath) { // SequentialRenderer with its own footprint (the oriented disc), fed RadixSorter's near-to-far order
    const seq = new sr.SequentialRenderer(device, null, 'rgba8unorm', n);
    seq.render(uniforms, props.getPropertyBuffer(), sorter.getSortedIndicesBuffer(), normals, W, H);
    fs.writeFileSync(seqPath, Buffer.from(seq.readPixels().buffer));
    seqPairs = seq.binner.getTotalIndices();
    seq.destroy();
  }
  if (discFramePath) { // the whole-frame facade with that footprint, colour plane pre-lit
    const whole = new sr.Renderer(device, null, 'rgba8unorm', n, 16, { footprint: 'disc' });
    for (let k = 0; k < 2; k++) whole.render(uniforms, props.getLitPlanes(normals), normals, null, W, H);
    fs.writeFileSync(discFramePath, Buffer.from(whole.readPixels().buffer));
    discFramePairs = whole.binner.getTotalIndices();
  }
  console.log(JSON.stringify({ n, W, H, pairs: binner.getTotalIndices(), tileRendererEqualsStaged, framePairs, schedulesKeepTheBytes, kernelsWithinOneLsb, timed, seqPairs, discFramePairs, recordFormat, bandEqualsFrame, bandPairs,
    pointManagerOk, ranking: device.rankStatus(), uniforms: Array.from(uniforms) }));
})().catch((e) => { console.error(e); process.exit(1); });
