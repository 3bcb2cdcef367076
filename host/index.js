#!/usr/bin/env node
'use strict';
// CLI: node host/index.js [--scene file.json] [--width W --height H] [--spp N] [--accel bvh2|lbvh|none]
//                         [--out image.ppm] [--dump prefix] [--pack-only prefix]
const fs = require('fs');
const { Main, writePPM } = require('./main');
const sceneLoader = require('./sceneLoader');

const args = {};
for (let i = 2; i < process.argv.length; i++) {
  const k = process.argv[i];
  if (k.startsWith('--')) args[k.slice(2)] = (i + 1 < process.argv.length && !process.argv[i + 1].startsWith('--')) ? process.argv[++i] : true;
}
const num = (k, d) => (k in args ? Number(args[k]) : d);

if (args['pack-only']) { // dump the packed host buffers (used by the packer parity test; needs no GPU)
  const scene = sceneLoader.loadScene(args.scene);
  if (args.width) scene.camera = { ...scene.camera, width: num('width'), height: num('height', num('width')) };
  const p = sceneLoader.pack(scene, undefined, args.scene ? require('path').dirname(require('path').resolve(args.scene)) : undefined);
  const pre = args['pack-only'];
  fs.writeFileSync(`${pre}.primitives.bin`, Buffer.from(p.primitives));
  fs.writeFileSync(`${pre}.patches.bin`, Buffer.from(p.patches));
  fs.writeFileSync(`${pre}.lights.bin`, Buffer.from(p.lights));
  fs.writeFileSync(`${pre}.camera.bin`, Buffer.from(p.camera.buffer));
  fs.writeFileSync(`${pre}.spectra.bin`, Buffer.from(p.spectra.buffer));
  fs.writeFileSync(`${pre}.cie.bin`, Buffer.from(p.cie.buffer));
  console.log(JSON.stringify({ nprim: p.primitives.byteLength / 80, nlight: p.lights.byteLength / 80, keyIndex: p.keyIndex }));
  process.exit(0);
}

const spp = num('spp', 16);
const r = Main({ sceneFile: args.scene, width: args.width ? num('width') : undefined, height: args.height ? num('height') : undefined,
  accel: args.accel || 'bvh2', device: num('device', 0) });
r.enableCounters(true);
const t0 = process.hrtime.bigint();
r.run(spp, !args.unfused);
const dt = Number(process.hrtime.bigint() - t0) / 1e9;
const c = r.counters();
const rgba = r.readRgba8();
if (args.out) writePPM(args.out, rgba, r.width, r.height);
if (args.dump) {
  fs.writeFileSync(`${args.dump}.rgba8.bin`, Buffer.from(rgba.buffer));
  fs.writeFileSync(`${args.dump}.accum.bin`, Buffer.from(r.readAccum().buffer));
}
console.log(JSON.stringify({ width: r.width, height: r.height, spp, sample: r.sample, seconds: dt,
  rays: c[0], mrays_per_s: c[0] / dt / 1e6, kernel_ms: r.lastTraceMs()[0] }));
r.destroy();
