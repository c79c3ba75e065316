'use strict';
// node sdf_generate.js <positions.f32> <n> <out.f32>
// The producer half of the reference's frame (src/main.ts:146-180) through the JS host classes: the scene main.ts builds,
// five rounds of {evaluateGradients, updatePositions, swap}, computeScaleFactors; writes positions | gradients (of the last
// evaluation) | vec4(normal, scale) for tests/test_napi.py to compare with the oracle.  With a fourth argument: also the reference's
// own constructor, new PointManager(device, scene, seed) — point count from the scene, two clouds drawn on the device — as
// <seeded.f32> = cloud(seed) | cloud(seed + 1).
const fs = require('fs');
const sr = require('./index.js');
const [posPath, nStr, outPath, seededPath] = process.argv.slice(2);
const n = +nStr;
const b = fs.readFileSync(posPath);
const device = new sr.Device(0);
const scene = new sr.SDFScene();
const sphere1 = new sr.Sphere({ id: 'sphere1', position: [0, 0, 0], radius: 0.5 });
const box1 = new sr.Box({ id: 'box1', position: [0.6, 0, 0], size: [0.3, 0.3, 0.3] });
const sphere2 = new sr.Sphere({ id: 'sphere2', position: [0, 0.6, 0], radius: 0.25 });
scene.setRoot(sr.smoothUnion(0.1, sr.smoothUnion(0.15, sphere1, box1), sphere2)); // main.ts:85
let seededCount = 0;
if (seededPath) { // before sphere1 moves: the box is the scene's as main.ts builds it
  const spm = new sr.PointManager(device, scene, 33);
  seededCount = spm.getNumPoints();
  const clouds = new Float32Array(seededCount * 8);
  clouds.set(spm.getCurrentPositionBuffer().read(new Float32Array(seededCount * 4)), 0);
  spm.reinitialize();
  clouds.set(spm.getCurrentPositionBuffer().read(new Float32Array(seededCount * 4)), seededCount * 4);
  fs.writeFileSync(seededPath, Buffer.from(clouds.buffer));
  spm.destroy();
}
const pm = new sr.PointManager(device, new Float32Array(b.buffer, b.byteOffset, n * 4));
const gs = new sr.GradientSampler(device, scene, n), cs = new sr.CurvatureSampler(device, scene, n), pu = new sr.PositionUpdater(device, null, n);
sphere1.position[0] = 0.1; // animate, then tell the samplers (main.ts:114-120)
gs.updateSceneParameters(); cs.updateSceneParameters();
for (let i = 0; i < 5; i++) { // main.ts:149-172
  gs.evaluateGradients(null, null, pm.getCurrentPositionBuffer());
  pu.updatePositions(null, null, pm.getCurrentPositionBuffer(), gs.getGradientBuffer(), pm.getNextPositionBuffer());
  pm.swap();
}
cs.computeScaleFactors(null, pm.getCurrentPositionBuffer());
const out = new Float32Array(n * 12);
out.set(pm.getCurrentPositionBuffer().read(new Float32Array(n * 4)), 0);
out.set(gs.getGradientBuffer().read(new Float32Array(n * 4)), n * 4);
out.set(cs.getCurvatureBuffer(gs.getGradientBuffer()).read(new Float32Array(n * 4)), n * 8);
fs.writeFileSync(outPath, Buffer.from(out.buffer));
// the same producer in one launch (native.sdf_generate) from the same starting cloud: the same bits
const start = device.createBuffer(n * 16); start.write(new Float32Array(b.buffer, b.byteOffset, n * 4));
const fp = device.createBuffer(n * 16), fg = device.createBuffer(n * 16), fc = device.createBuffer(n * 16), fprops = device.createBuffer(n * 32);
sr.native.sdf_generate(device.ctx, scene.program(), null, null, 0, start.ptr, n, 5, fp.ptr, fg.ptr, fc.ptr, fprops.ptr);
const same = (x, y) => { const a = new Uint32Array(x.buffer, x.byteOffset, x.length), c = new Uint32Array(y.buffer, y.byteOffset, y.length); for (let i = 0; i < a.length; i++) if (a[i] !== c[i]) return false; return true; };
const fusedEqual = same(fp.read(new Float32Array(n * 4)), out.subarray(0, n * 4)) && same(fg.read(new Float32Array(n * 4)), out.subarray(n * 4, n * 8)) &&
  same(fc.read(new Float32Array(n * 4)), out.subarray(n * 8, n * 12));
console.log(JSON.stringify({ n, hash: scene.getStructureHash(), seededCount, fusedEqual }));
