#!/usr/bin/env node
'use strict';
/*
 * display_loop.js -- the reference's frame loop as a display loop sees it (src/main.js:597-620: one compute pass per
 * sample, then the blit of THAT frame, then requestAnimationFrame): every frame index is shown exactly once, in order,
 * while the renderer runs ahead of the display by `lag` frames (64: a cohort of 16 samples plus the time its longest paths take to
 * retire, DESIGN.md 5.1) -- far enough for small calls to be merged into cohorts and for no read to wait for its frame.
 *
 *   frame k:   trace(1)                       request sample k          (returns at once: the call only notes the sample)
 *              readSampleRgba8(k - lag)       show sample k - lag       (from the ring of the last F frames; waits only for
 *                                                                        the batch that holds it, never flushes)
 *
 *   node host/display_loop.js [--scene f.json | --packed prefix] [--width W --height H] [--frames 640] [--lag 64] [--ring 128] [--check 1]
 *   --check 1: every shown frame is compared with a synchronous trace(1); sync(); readRgba8() loop on a second context.
 * Prints one JSON line: ms per shown frame, frames shown, whether every index appeared once and in order.
 */
const path = require('path');
const { loadAddon } = require('./main');
const sceneLoader = require('./sceneLoader');

const args = {};
for (let i = 2; i < process.argv.length; i++) {
  const k = process.argv[i];
  if (k.startsWith('--')) args[k.slice(2)] = (i + 1 < process.argv.length && !process.argv[i + 1].startsWith('--')) ? process.argv[++i] : true;
}
const num = (k, d) => (k in args ? Number(args[k]) : d);
const a = loadAddon();
let packed;
if (args.packed) {
  // buffers packed elsewhere (tools/dump_packed.py writes the synthetic benchmark scenes): <prefix>.{primitives,lights,camera,spectra,cie}.bin
  const fs = require('fs');
  const rd = (n) => { const b = fs.readFileSync(`${args.packed}.${n}.bin`); return b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength); };
  const camera = new Float32Array(rd('camera'));
  packed = { primitives: rd('primitives'), lights: rd('lights'), camera, spectra: new Float32Array(rd('spectra')), cie: new Float32Array(rd('cie')),
    width: camera[11], height: camera[12] };
} else {
  const scene = sceneLoader.loadScene(args.scene);
  if (args.width) scene.camera = { ...scene.camera, width: num('width'), height: num('height', num('width')) };
  packed = sceneLoader.pack(scene, undefined, args.scene ? path.dirname(path.resolve(args.scene)) : undefined);
}
const frames = num('frames', 640), lag = num('lag', 64), ring = num('ring', 128), check = num('check', 0);

function make() {
  const h = a.create(num('device', 0));
  a.uploadScene(h, packed.primitives, packed.lights, packed.spectra, packed.cie, packed.camera);
  a.buildAccel(h, 1);
  return h;
}
const h = make();
a.setOption(h, 'frame_ring', ring);
let ref = null;
if (check) ref = make();

let shown = 0, inOrder = true, equal = true;
// the display's frame buffer: one page-locked array, reused frame after frame (the readback then runs at PCIe speed)
const frame = new Uint8Array(packed.width * packed.height * 4);
a.pinHost(frame);
const show = (k) => {
  a.readSampleRgba8Into(h, k, frame);                        // the display step: frame k, complete, as a synced loop would show it
  shown++;
  if (ref) {
    a.trace(ref, 1); a.sync(ref);
    const want = a.readRgba8(ref);
    if (a.sampleCount(ref) !== k) inOrder = false;
    for (let i = 0; i < want.length; i++) if (want[i] !== frame[i]) { equal = false; break; }
  }
};
const t0 = process.hrtime.bigint();
for (let k = 1; k <= frames; k++) {
  a.trace(h, 1);
  if (k > lag) show(k - lag);
}
for (let k = Math.max(1, frames - lag + 1); k <= frames; k++) show(k);   // the tail: the last frames are shown as they complete
const ms = Number(process.hrtime.bigint() - t0) / 1e6;
console.log(JSON.stringify({ width: packed.width, height: packed.height, frames, lag, ring, shown, every_index_once_in_order: shown === frames && inOrder,
  equal_to_synced_loop: ref ? equal : null, ms_per_frame: ms / frames, latest: a.latestSample(h) }));
a.unpinHost(frame);
a.destroy(h);
if (ref) a.destroy(ref);
