'use strict';
/*
 * webgpu_main.js -- a host that talks WebGPU only (navigator.gpu, GPUBufferUsage, command
 * encoders, requestAnimationFrame), making the kinds of calls the reference's Main() makes
 * (src/main.js:8-621): five mapped-at-creation storage buffers, an rgba8 storage texture,
 * a nine-entry bind group for the path tracer, a one-entry bind group for the sample counter,
 * and per frame {counter pass, path-trace pass, blit pass} in one submit.  It knows nothing
 * of the addon: run it with host/webgpu.js installed and the passes execute in libcrt.
 *
 *   node host/webgpu_main.js --scene scenes/cornell_box.json --size 128 --frames 8 --out f.ppm
 */
const path = require('path');
const sceneLoader = require('./sceneLoader');

async function main(g, packed, shaderText = { trace: '/* ComputeShader.wgsl */', count: '/* UpdateVariables.wgsl */' }) {
  const adapter = await g.navigator.gpu.requestAdapter();
  const device = await adapter.requestDevice();
  device.addEventListener('uncapturederror', (e) => { throw e.error; });
  const canvas = g.document.getElementById('canvas');
  canvas.width = packed.width; canvas.height = packed.height;
  const context = canvas.getContext('webgpu');
  context.configure({ device, format: 'bgra8unorm' });

  const storage = (bytes) => {
    const b = device.createBuffer({ size: bytes.byteLength, usage: g.GPUBufferUsage.STORAGE, mappedAtCreation: true });
    new Uint8Array(b.getMappedRange()).set(new Uint8Array(bytes.buffer || bytes, bytes.byteOffset || 0, bytes.byteLength));
    b.unmap();
    return b;
  };
  const zeros = (n) => storage(new Uint8Array(n));
  const npix = packed.width * packed.height;
  const framebuffer = device.createTexture({ size: { width: packed.width, height: packed.height }, format: 'rgba8unorm',
    usage: g.GPUTextureUsage.STORAGE_BINDING | g.GPUTextureUsage.TEXTURE_BINDING });
  const buffers = [zeros(npix * 16), zeros(4), zeros(16), // b1 accumulator, b2 sample, b3 unused slot
    storage(packed.camera), storage(packed.cie), storage(packed.spectra), storage(packed.lights), storage(packed.primitives)];

  const vis = g.GPUShaderStage.COMPUTE;
  const traceLayout = device.createBindGroupLayout({ entries: [{ binding: 0, visibility: vis, storageTexture: { format: 'rgba8unorm' } },
    ...buffers.map((_, i) => ({ binding: i + 1, visibility: vis, buffer: { type: i < 2 ? 'storage' : 'read-only-storage' } }))] });
  const traceGroup = device.createBindGroup({ layout: traceLayout, entries: [{ binding: 0, resource: framebuffer.createView() },
    ...buffers.map((b, i) => ({ binding: i + 1, resource: { buffer: b } }))] });
  const countLayout = device.createBindGroupLayout({ entries: [{ binding: 0, visibility: vis, buffer: { type: 'storage' } }] });
  const countGroup = device.createBindGroup({ layout: countLayout, entries: [{ binding: 0, resource: { buffer: buffers[1] } }] });
  const pipe = (layout, code, entryPoint) => device.createComputePipeline({
    layout: device.createPipelineLayout({ bindGroupLayouts: [layout] }), compute: { module: device.createShaderModule({ code }), entryPoint } });
  const tracePipe = pipe(traceLayout, shaderText.trace, 'main');
  const countPipe = pipe(countLayout, shaderText.count, 'main');

  const frame = () => {
    const enc = device.createCommandEncoder();
    let pass = enc.beginComputePass();
    pass.setPipeline(countPipe); pass.setBindGroup(0, countGroup); pass.dispatchWorkgroups(1); pass.end();
    pass = enc.beginComputePass();
    pass.setPipeline(tracePipe); pass.setBindGroup(0, traceGroup);
    pass.dispatchWorkgroups(Math.ceil(packed.width / 8), Math.ceil(packed.height / 8)); pass.end();
    const blit = enc.beginRenderPass({ colorAttachments: [{ view: context.getCurrentTexture().createView(), loadOp: 'clear', storeOp: 'store' }] });
    blit.draw(3); blit.end();
    device.queue.submit([enc.finish()]);
    g.requestAnimationFrame(frame);
  };
  g.requestAnimationFrame(frame);
  return device;
}

if (require.main === module) {
  const arg = (name, dflt) => { const i = process.argv.indexOf(name); return i < 0 ? dflt : process.argv[i + 1]; };
  const file = arg('--scene', path.join(__dirname, '..', 'scenes', 'cornell_box.json'));
  const scene = sceneLoader.loadScene(file);
  const size = parseInt(arg('--size', '0'), 10);
  if (size) scene.camera = { ...scene.camera, width: size, height: size };
  const packed = sceneLoader.pack(scene, undefined, path.dirname(path.resolve(file)));
  const g = {};
  require('./webgpu').install(g, { frames: parseInt(arg('--frames', '4'), 10), accel: arg('--accel', 'bvh'),
    onDone: (device, canvas) => {
      const out = arg('--out'), dump = arg('--dump');
      if (out) require('./main').writePPM(out, canvas.pixels, packed.width, packed.height);
      if (dump) require('fs').writeFileSync(dump, Buffer.from(canvas.pixels));
      console.log(JSON.stringify({ width: packed.width, height: packed.height, frames: parseInt(arg('--frames', '4'), 10) }));
      device.destroy();
    } });
  main(g, packed).catch((e) => { console.error(e); process.exit(1); });
}

module.exports = { main };
