#!/usr/bin/env node
'use strict';
/*
 * multi.js -- the Node host of the multi-GPU configurations (BASELINE configs 4 and 5 x 8): the frame partitioned by
 * rows across N GPUs of one node, ONE worker process per GPU, the path's only exchange the gather of the finished
 * strips -- an RCCL all-gather over xGMI issued by libcrt (crt_gather, include/crt.h "Multi-GPU").
 *
 * The reference has one GPUDevice and no such step (src/main.js:8-9); what each worker runs is the reference's own
 * sequence -- Main(): create, upload, then frame() in a loop (src/main.js:597-620) -- on its rows of the frame.
 *
 *   parent   forks N workers BEFORE anything touches a GPU (it never does itself), relays the communicator id that
 *            rank 0 makes to the others, collects their reports
 *   worker   create(device = rank) -> commInit(id, rank, N) -> uploadScene -> commPartition(band) -> buildAccel
 *            -> { trace(spp); gather(rgba8) } x frames -> sync -> gather(rgba8 | accum); rank 0 writes the image
 *
 *   node host/multi.js --gpus 8 [--scene file.json] [--width W --height H] [--spp 64] [--frames 4] [--band 8]
 *                      [--accel bvh2|lbvh] [--out frame.ppm] [--local]
 *   --local: all ranks in THIS process on device 0, joined by the in-process transport (CRT_COMM_LOCAL: device-to-device
 *            copies stand where the RCCL all-gather does) -- how the path is exercised on a one-GPU box.
 */
const path = require('path');
const { fork } = require('child_process');
const { writePPM, loadAddon } = require('./main');
const sceneLoader = require('./sceneLoader');

const ACCEL = { none: 0, brute: 0, bvh2: 1, bvh: 1, lbvh: 2 };
const GATHER_RGBA8 = 1, GATHER_ACCUM = 2;

function parseArgs(argv) {
  const args = {};
  for (let i = 0; i < argv.length; i++) {
    const k = argv[i];
    if (k.startsWith('--')) args[k.slice(2)] = (i + 1 < argv.length && !argv[i + 1].startsWith('--')) ? argv[++i] : true;
  }
  return args;
}

function packScene(opt) {
  const scene = sceneLoader.loadScene(opt.scene);
  if (opt.width) scene.camera = { ...scene.camera, width: Number(opt.width), height: Number(opt.height || opt.width) };
  return sceneLoader.pack(scene, undefined, opt.scene ? path.dirname(path.resolve(opt.scene)) : undefined);
}

// One rank's part: the reference's Main() + frame() loop on this rank's rows.
function setupRank(a, opt, id, rank, world, device) {
  const packed = packScene(opt);
  const h = a.create(device);
  a.commInit(h, id, rank, world);
  a.uploadScene(h, packed.primitives, packed.lights, packed.spectra, packed.cie, packed.camera);
  a.commPartition(h, Number(opt.band !== undefined ? opt.band : 8));
  a.buildAccel(h, ACCEL[opt.accel || 'bvh2']);
  return { h, packed };
}

function runFrames(a, ranks, opt) {
  const spp = Number(opt.spp || 16), frames = Number(opt.frames || 2);
  const t0 = process.hrtime.bigint();
  for (let f = 0; f < frames; f++) {
    for (const r of ranks) a.trace(r.h, spp);                    // asynchronous: the frame's paths finish under the next frame's
    // the display step of a frame (src/main.js:612-617 blits; here the strips are shipped): the latest COMPLETE frame
    for (const r of ranks) if (f > 0) a.gather(r.h, GATHER_RGBA8);
  }
  for (const r of ranks) a.sync(r.h);
  for (const r of ranks) a.gather(r.h, GATHER_RGBA8 | GATHER_ACCUM);
  const rgba = a.readFrameRgba8(ranks[0].h);                     // (an all-gather: every rank holds the frame; rank 0 reads it)
  const seconds = Number(process.hrtime.bigint() - t0) / 1e9;
  return { rgba, seconds, samples: spp * frames };
}

function worker() {
  const opt = JSON.parse(process.argv[3]);
  const rank = Number(opt.rank), world = Number(opt.world);
  const a = loadAddon();
  const start = (idBuf) => {
    try {
      const id = new Uint8Array(idBuf);
      const r = setupRank(a, opt, id, rank, world, opt.oneGpu ? 0 : rank);
      const res = runFrames(a, [r], opt);
      const info = a.commInfo(r.h);
      if (rank === 0 && opt.out) writePPM(opt.out, res.rgba, r.packed.width, r.packed.height);
      let sum = 0;
      for (let i = 0; i < res.rgba.length; i += 97) sum = (sum + res.rgba[i] * (i % 251 + 1)) >>> 0;
      process.send({ done: true, rank, rows: info[3], seconds: res.seconds, samples: res.samples, checksum: sum, width: r.packed.width, height: r.packed.height });
      a.destroy(r.h);
      process.exit(0);
    } catch (e) {
      process.send({ error: String(e && e.message || e), rank });
      process.exit(1);
    }
  };
  process.on('message', (m) => { if (m.id) start(Buffer.from(m.id, 'base64')); });
  if (rank === 0) {
    // rank 0 makes the communicator id (ncclGetUniqueId inside libcrt) and hands it to the parent
    try { process.send({ id: Buffer.from(a.commUniqueId(false)).toString('base64') }); }
    catch (e) { process.send({ error: String(e && e.message || e), rank }); process.exit(1); }
  }
}

function parent(args) {
  const world = Number(args.gpus || 1);
  if (args.local) {
    // every rank in this process, device 0, in-process transport
    const a = loadAddon();
    const id = new Uint8Array(a.commUniqueId(true));
    const ranks = [];
    for (let k = 0; k < world; k++) ranks.push(setupRank(a, args, id, k, world, Number(args.device || 0)));
    const res = runFrames(a, ranks, args);
    if (args.out) writePPM(args.out, res.rgba, ranks[0].packed.width, ranks[0].packed.height);
    if (args.dump) require('fs').writeFileSync(args.dump, Buffer.from(res.rgba.buffer));
    console.log(JSON.stringify({ gpus: world, transport: 'local', width: ranks[0].packed.width, height: ranks[0].packed.height,
      samples: res.samples, seconds: res.seconds, rows: ranks.map((r) => a.commInfo(r.h)[3]) }));
    for (const r of ranks) a.destroy(r.h);
    return;
  }
  // one worker per GPU, forked before any GPU call (this process never makes one)
  const env = { ...process.env, HSA_ENABLE_IPC_MODE_LEGACY: process.env.HSA_ENABLE_IPC_MODE_LEGACY || '0' };
  const kids = [];
  const reports = [];
  let failed = false;
  for (let k = 0; k < world; k++) {
    const opt = { ...args, rank: k, world, oneGpu: !!args['one-gpu'] };
    const c = fork(__filename, ['--worker', JSON.stringify(opt)], { env });
    c.on('message', (m) => {
      if (m.id) for (const o of kids) o.send({ id: m.id });            // rank 0's id to every rank (itself included)
      else if (m.done) {
        reports[m.rank] = m;
        if (reports.filter(Boolean).length === world) {
          const sums = new Set(reports.map((r) => r.checksum));
          console.log(JSON.stringify({ gpus: world, transport: 'rccl', width: reports[0].width, height: reports[0].height, samples: reports[0].samples,
            seconds: Math.max(...reports.map((r) => r.seconds)), rows: reports.map((r) => r.rows), every_rank_holds_the_same_frame: sums.size === 1 }));
        }
      } else if (m.error) { failed = true; console.error(`rank ${m.rank}: ${m.error}`); for (const o of kids) o.kill(); }
    });
    c.on('exit', (code) => { if (code && !failed) { failed = true; console.error(`rank ${k} exited with ${code}`); } if (failed) process.exitCode = 1; });
    kids.push(c);
  }
}

if (process.argv[2] === '--worker') worker();
else parent(parseArgs(process.argv.slice(2)));
