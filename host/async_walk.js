#!/usr/bin/env node
'use strict';
// A seeded random walk over the asynchronous N-API entry points of ONE context: traceAsync / readSampleRgba8Async /
// readRgba8Async / syncAsync promises queued without waiting for the ones before (the addon runs a context's jobs in call
// order on a worker thread), invalid requests in between (their promise rejects, the queue goes on), blocking calls refused
// while jobs are pending.  Every frame a promise delivers is compared with the frame a synchronous trace(1); sync();
// readRgba8() loop shows at that sample index.      node host/async_walk.js [--size N] [--epochs N] [--seed N]
const { Main, loadAddon } = require('./main');
const args = {};
for (let i = 2; i < process.argv.length; i += 2) args[process.argv[i].slice(2)] = Number(process.argv[i + 1]);
const size = args.size || 96, epochs = args.epochs || 6, K = 32, RING = 24;
let seed = (args.seed || 20261007) >>> 0;
const rnd = () => { seed = (Math.imul(seed, 1664525) + 1013904223) >>> 0; return seed / 4294967296; };
const same = (a, b) => Buffer.compare(Buffer.from(a.buffer, a.byteOffset, a.byteLength), Buffer.from(b.buffer, b.byteOffset, b.byteLength)) === 0;

(async () => {
  const a = loadAddon();
  const ref = Main({ width: size, height: size });
  const want = [null];
  for (let k = 1; k <= K; k++) { ref.frame(); ref.sync(); want.push(Uint8Array.from(ref.readRgba8())); }
  ref.destroy();
  const r = Main({ width: size, height: size });
  const h = r.device;
  let checked = 0, rejected = 0, busy = 0;
  for (let e = 0; e < epochs; e++) {
    a.setOption(h, 'frame_ring', RING); a.setOption(h, 'wf_cohort', [1, 4, 16][Math.floor(rnd() * 3)]);
    a.setOption(h, 'wf_ring', [2, 3, 32][Math.floor(rnd() * 3)]);
    r.reset();
    let total = 0;
    const pending = [];
    const steps = 8 + Math.floor(rnd() * 24);
    for (let s = 0; s < steps; s++) {
      const n = [1, 1, 1, 2, 3][Math.floor(rnd() * 5)];
      if (total + n > K) break;
      pending.push(a.traceAsync(h, n)); total += n;
      const what = rnd();
      if (what < 0.35) {
        const lo = Math.max(1, total - RING + 1), k = lo + Math.floor(rnd() * (total - lo + 1));
        pending.push(a.readSampleRgba8Async(h, k).then((img) => { if (!same(img, want[k])) throw new Error(`frame ${k} differs (epoch ${e})`); checked++; }));
      } else if (what < 0.45) {
        pending.push(a.readSampleRgba8Async(h, total + 5).then(() => { throw new Error('a frame that was never requested was delivered'); }, () => { rejected++; }));
      } else if (what < 0.55) {
        const t = total;
        pending.push(a.readRgba8Async(h).then((img) => { if (!same(img, want[t])) throw new Error(`readRgba8Async after ${t} samples differs (epoch ${e})`); checked++; }));
      } else if (what < 0.65) {
        try { a.latestSample(h); if (pending.length) throw new Error('a blocking call ran while jobs were pending'); } catch (err) { if (err.code === 'ERR_CRT_BUSY') busy++; else throw err; }
      } else if (what < 0.75) {
        await Promise.all(pending.splice(0));                      // the caller catches up
      }
    }
    pending.push(a.syncAsync(h));
    await Promise.all(pending);
    if (a.sampleCount(h) !== total) throw new Error(`sample count ${a.sampleCount(h)} != ${total}`);
    if (total && !same(a.readRgba8(h), want[total])) throw new Error(`final frame differs (epoch ${e})`);
  }
  r.destroy();
  console.log(JSON.stringify({ epochs, checked, rejected, busy, ok: true }));
})().catch((e) => { console.error(e); process.exit(1); });
