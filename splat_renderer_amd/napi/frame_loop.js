'use strict';
// node frame_loop.js <props.f32> <normals.f32> <n> <W> <H> <outPrefix>
// The headless frame loop in the reference's own language (src/main.ts:110-193 + src/OrbitCameraController.ts): four frames
// of an orbit — Camera.rotate, then a pan, a wheel zoom and a left-button drag through OrbitCameraController — rendered
// back to back through FrameLoop; writes <outPrefix><k>.rgba8 and prints the uniform blocks.  tests/test_napi.py runs the
// same moves through splat_renderer_amd/frameloop.py and compares the images byte for byte.
const fs = require('fs');
const sr = require('./index.js');
const [propsPath, normalsPath, nStr, wStr, hStr, outPrefix] = process.argv.slice(2);
const n = +nStr;
const W = +wStr;
const H = +hStr;
const f32 = (p) => {
  const b = fs.readFileSync(p);
  return new Float32Array(b.buffer, b.byteOffset, b.length / 4);
};
const device = new sr.Device(0);
const props = device.createBufferFrom(f32(propsPath));
const normals = device.createBufferFrom(f32(normalsPath));
const loop = new sr.FrameLoop(device, n, W, H);
const ctl = new sr.OrbitCameraController(loop.camera, null);
const moves = [
  () => loop.camera.rotate((2 * Math.PI) / 4, 0.0),
  () => loop.camera.pan(0.2, -0.1),
  () => ctl.onWheel({ deltaY: 400.0 }),
  () => {
    ctl.onMouseDown({ clientX: 10, clientY: 10, button: 0 });
    ctl.onMouseMove({ clientX: 70, clientY: 40, button: 0 });
    ctl.onMouseUp({});
  },
];
const uniforms = [];
const pairs = [];
moves.forEach((move, k) => {
  uniforms.push(Array.from(loop.camera.uniforms(W, H, k / 60.0)));
  loop.render(props, normals);
  fs.writeFileSync(`${outPrefix}${k}.rgba8`, Buffer.from(loop.readPixels().buffer));
  pairs.push(loop.renderer.finish());
  move();
});
// the same frames without reading anything back in between (all sync-free): the same last image
const loop2 = new sr.FrameLoop(device, n, W, H);
const ctl2 = new sr.OrbitCameraController(loop2.camera, null);
const moves2 = [() => loop2.camera.rotate((2 * Math.PI) / 4, 0.0), () => loop2.camera.pan(0.2, -0.1), () => ctl2.onWheel({ deltaY: 400.0 })];
for (let k = 0; k < 4; k++) {
  loop2.render(props, normals);
  if (k < 3) moves2[k]();
}
fs.writeFileSync(`${outPrefix}sync_free_last.rgba8`, Buffer.from(loop2.readPixels().buffer));
// the whole-frame facade keeps lit composite records in the projector's buffer: the reference-layout getter must refuse
let refused = false;
try {
  loop.renderer.projector.getProjectedBuffer();
} catch (e) {
  refused = /lit composite records/.test(e.message);
}
console.log(JSON.stringify({ uniforms, pairs, refused, recordFormat: loop.renderer.recordFormat }));
loop.destroy();
loop2.destroy();
props.destroy();
normals.destroy();
device.destroy();
