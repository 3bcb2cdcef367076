'use strict';
/*
 * main.js -- Node host of the compute pass: the reference's Main()/frame()
 * (src/main.js:7-624) with the browser removed and WebGPU replaced by the
 * HIP library behind the N-API addon.
 *
 *   reference (src/main.js)                       here
 *   8-9    requestAdapter / requestDevice         addon.create(device)
 *   114-393 flatten + pack + createBuffer/unmap   sceneLoader.pack + addon.uploadScene
 *   298-311 zero accumulator, sample = 0          (done by uploadScene / reset)
 *   597-611 per frame: dispatch(1); dispatch(W/8,H/8)   frame(): addon.trace(h, 1)
 *   612-617 blit to the canvas                    readRgba8() / writePPM()   (display only)
 *   620     requestAnimationFrame(frame) forever  run(spp): spp frames, optionally fused
 */
const fs = require('fs');
const path = require('path');
const sceneLoader = require('./sceneLoader');

let addon;
function loadAddon() {
  if (!addon) addon = require(path.join(__dirname, '..', 'addon', 'crt_napi.node')); // throws if not built
  return addon;
}

const ACCEL = { none: 0, brute: 0, bvh2: 1, bvh: 1, lbvh: 2 };

function Main(options = {}) {
  const a = loadAddon();
  const scene = options.scene || sceneLoader.loadScene(options.sceneFile);
  if (options.width) scene.camera = { ...scene.camera, width: options.width, height: options.height || options.width };
  const packed = sceneLoader.pack(scene, options.cie, options.sceneFile ? path.dirname(path.resolve(options.sceneFile)) : undefined);
  const { width, height } = packed;

  const device = a.create(options.device || 0);
  a.uploadScene(device, packed.primitives, packed.lights, packed.spectra, packed.cie, packed.camera);
  if (options.tile) a.setTile(device, ...options.tile);
  a.buildAccel(device, ACCEL[options.accel || 'bvh2']);

  // one reference frame = { sample++ ; trace } (main.js:598-611)
  const frame = () => a.trace(device, 1);
  const run = (spp, fused = true) => {
    if (fused) a.trace(device, spp); // same result as spp frames (summed in sample order)
    else for (let i = 0; i < spp; i++) frame();
    a.sync(device);
  };

  return {
    width, height, packed, device, frame, run,
    // Promise forms: the job runs on a worker thread, the event loop stays free (the reference's frame() is
    // fire-and-forget too: queue.submit, src/main.js:618-620).  Jobs of one device run in call order; the blocking
    // calls throw ERR_CRT_BUSY while any are pending.
    frameAsync: (n = 1) => a.traceAsync(device, n),
    syncAsync: () => a.syncAsync(device),
    readRgba8Async: () => a.readRgba8Async(device),
    readAccumAsync: () => a.readAccumAsync(device),
    sync: () => a.sync(device),
    reset: () => a.reset(device),
    get sample() { return a.sampleCount(device); },
    readAccum: () => a.readAccum(device),
    readRgba8: () => a.readRgba8(device),
    counters: () => a.counters(device),
    enableCounters: (on) => a.enableCounters(device, !!on),
    lastTraceMs: () => a.lastTraceMs(device),
    accelStats: () => a.accelStats(device),
    destroy: () => a.destroy(device),
  };
}

// Binary PPM of the rgba8 framebuffer (row 0 = top, like the reference's blit).
function writePPM(file, rgba, width, height) {
  const out = Buffer.alloc(width * height * 3);
  for (let i = 0, j = 0; i < width * height * 4; i += 4, j += 3) {
    out[j] = rgba[i]; out[j + 1] = rgba[i + 1]; out[j + 2] = rgba[i + 2];
  }
  fs.writeFileSync(file, Buffer.concat([Buffer.from(`P6\n${width} ${height}\n255\n`), out]));
}

module.exports = { Main, writePPM, loadAddon };
