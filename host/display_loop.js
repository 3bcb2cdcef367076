#!/usr/bin/env node
'use strict';
/*
 * display_loop.js -- the reference's frame loop as a display loop sees it (src/main.js:597-620: one compute pass per
 * sample, then the blit of THAT frame, then requestAnimationFrame): every frame index is shown exactly once, in order,
 * while the renderer runs ahead of the display by `lag` frames -- far enough for small calls to be merged into
 * cohorts (DESIGN.md 5.1) and for no read to wait for work that has not been requested yet.
 *
 *   frame k:   trace(1)                       request sample k          (returns at once: the call only notes the sample)
 *              readSampleRgba8(k - lag)       show sample k - lag       (from the ring of the last F frames; waits only for
 *                                                                        the batch that holds it, never flushes)
 *
 *   node host/display_loop.js [--scene f.json] [--width W --height H] [--frames 320] [--lag 32] [--ring 64] [--check 1]
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
const scene = sceneLoader.loadScene(args.scene);
if (args.width) scene.camera = { ...scene.camera, width: num('width'), height: num('height', num('width')) };
const packed = sceneLoader.pack(scene, undefined, args.scene ? path.dirname(path.resolve(args.scene)) : undefined);
const frames = num('frames', 320), lag = num('lag', 32), ring = num('ring', 64), check = num('check', 0);

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
const show = (k) => {
  const frame = a.readSampleRgba8(h, k);                     // the display step: frame k, complete, as a synced loop would show it
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
a.destroy(h);
if (ref) a.destroy(ref);
