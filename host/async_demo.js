#!/usr/bin/env node
'use strict';
// The asynchronous N-API entry points: a 64-spp trace + sync + readback run on a worker thread while the event
// loop keeps turning (the reference's frame() is fire-and-forget: queue.submit, src/main.js:618-620); the image is the
// synchronous one, byte for byte.   node host/async_demo.js [--size N] [--spp N]
const { Main } = require('./main');
const args = {};
for (let i = 2; i < process.argv.length; i += 2) args[process.argv[i].slice(2)] = Number(process.argv[i + 1]);
const size = args.size || 512, spp = args.spp || 64;

(async () => {
  const r = Main({ width: size, height: size });
  r.run(spp);
  const want = Buffer.from(r.readRgba8().buffer.slice(0));
  r.reset();
  let ticks = 0, running = true;
  const spin = () => { if (running) { ticks++; setImmediate(spin); } };
  spin();
  const p1 = r.frameAsync(spp), p2 = r.syncAsync(), p3 = r.readRgba8Async();
  let busy = false;
  try { r.readRgba8(); } catch (e) { busy = e.code === 'ERR_CRT_BUSY'; }
  await p1; await p2;
  const got = await p3;
  running = false;
  const sample = r.sample;
  let rejected = false;
  try { await r.frameAsync(1).then(() => r.syncAsync()); r.destroy(); await r.syncAsync(); } catch (e) { rejected = true; }
  console.log(JSON.stringify({ ticks, busy, equal: Buffer.compare(Buffer.from(got.buffer), want) === 0, sample, rejected }));
})().catch((e) => { console.error(e); process.exit(1); });
