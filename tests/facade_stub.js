'use strict';
// CPU check of host/webgpu.js: a recording stand-in for the addon shows which libcrt calls
// the facade makes for the WebGPU command stream host/webgpu_main.js submits.
const Module = require('module');
const path = require('path');
const real = Module._load;
const calls = [];
Module._load = function (req, ...rest) {
  if (!req.endsWith('crt_napi.node')) return real.call(this, req, ...rest);
  return {
    create: (ordinal) => ({ ordinal }),
    destroy: () => calls.push(['destroy']),
    uploadScene: (h, p, l, s, c, cam) => calls.push(['upload', p.byteLength, l.byteLength, s.byteLength, c.byteLength, cam.byteLength]),
    buildAccel: (h, mode) => calls.push(['accel', mode]),
    trace: (h, n) => calls.push(['trace', n]),
    readRgba8: () => new Uint8Array(16 * 16 * 4).fill(7),
  };
};
const root = path.join(__dirname, '..');
const sl = require(path.join(root, 'host', 'sceneLoader'));
const scene = sl.loadScene(path.join(root, 'scenes', 'cornell_box.json'));
scene.camera = { ...scene.camera, width: 16, height: 16 };
const packed = sl.pack(scene, undefined, path.join(root, 'scenes'));
const g = {};
require(path.join(root, 'host', 'webgpu')).install(g, {
  frames: 2,
  onDone: (device, canvas) => { device.destroy(); console.log(JSON.stringify({ calls, px: canvas.pixels[5], n: canvas.pixels.length })); },
});
require(path.join(root, 'host', 'webgpu_main')).main(g, packed);
