'use strict';
/*
 * webgpu.js -- a WebGPU-shaped facade over the HIP addon: just the calls the reference's
 * Main() makes (src/main.js:8-621), so that host code written against `navigator.gpu` can
 * drive libcrt without knowing about it.  SURVEY.md 8(f)-4.
 *
 * What is real: storage buffers (createBuffer / getMappedRange / unmap), bind groups, compute
 * pipelines, command encoding and queue.submit.  On submit, a compute pass whose bind group
 * has the path tracer's nine entries (ComputeShader.wgsl:1-9) becomes
 *     uploadScene(b8 primitives, b7 lights, b6 spectra, b5 CIE, b4 camera)   (first time / after a change)
 *     trace(1)                                                                (every dispatch)
 * and a pass with the one-entry bind group (UpdateVariables.wgsl) is the `sample++` that
 * trace() already includes.  The rgba8 storage texture (b0) receives the framebuffer after
 * each submit.  What is a stub: render pipelines, samplers, the canvas context (the blit pass,
 * src/main.js:612-617, is display only) -- `canvas.pixels` holds the last frame instead.
 */
const path = require('path');

const GPUBufferUsage = { MAP_READ: 1, MAP_WRITE: 2, COPY_SRC: 4, COPY_DST: 8, INDEX: 16, VERTEX: 32, UNIFORM: 64, STORAGE: 128, INDIRECT: 256, QUERY_RESOLVE: 512 };
const GPUTextureUsage = { COPY_SRC: 1, COPY_DST: 2, TEXTURE_BINDING: 4, STORAGE_BINDING: 8, RENDER_ATTACHMENT: 16 };
const GPUShaderStage = { VERTEX: 1, FRAGMENT: 2, COMPUTE: 4 };

class Buffer {
  constructor(desc) {
    this.size = desc.size; this.usage = desc.usage; this.data = new ArrayBuffer(desc.size);
    this.mapped = !!desc.mappedAtCreation; this.version = 0;
  }
  getMappedRange() { if (!this.mapped) throw new Error('buffer is not mapped'); return this.data; }
  unmap() { this.mapped = false; this.version++; }
}
class Texture {
  constructor(desc) { this.width = desc.size.width; this.height = desc.size.height; this.format = desc.format; this.pixels = new Uint8Array(this.width * this.height * 4); }
  createView() { return { texture: this }; }
}

class ComputePass {
  constructor(enc) { this.enc = enc; this.pipeline = null; this.groups = []; }
  setPipeline(p) { this.pipeline = p; }
  setBindGroup(i, g) { this.groups[i] = g; }
  dispatchWorkgroups(x, y = 1, z = 1) { this.enc.cmds.push({ kind: 'dispatch', pipeline: this.pipeline, group: this.groups[0], x, y, z }); }
  end() {}
}
class RenderPass { setPipeline() {} setBindGroup() {} draw() {} end() {} }
class CommandEncoder {
  constructor() { this.cmds = []; }
  beginComputePass() { return new ComputePass(this); }
  beginRenderPass() { this.cmds.push({ kind: 'blit' }); return new RenderPass(); }
  finish() { return { cmds: this.cmds }; }
}

class Device {
  constructor(addon, ordinal, accel) {
    this.addon = addon; this.handle = addon.create(ordinal); this.accel = accel;
    this.listeners = {}; this.uploaded = null; this.canvas = null;
    this.queue = { submit: (lists) => this._submit(lists) };
  }
  addEventListener(type, fn) { (this.listeners[type] = this.listeners[type] || []).push(fn); }
  createBuffer(desc) { return new Buffer(desc); }
  createTexture(desc) { return new Texture(desc); }
  createSampler() { return {}; }
  createShaderModule(desc) { return { code: desc.code }; }
  createBindGroupLayout(desc) { return { entries: desc.entries }; }
  createBindGroup(desc) { return { layout: desc.layout, entries: desc.entries }; }
  createPipelineLayout(desc) { return desc; }
  createRenderPipeline(desc) { return { kind: 'render', desc }; }
  createComputePipeline(desc) { return { kind: 'compute', desc }; }
  createCommandEncoder() { return new CommandEncoder(); }
  destroy() { if (this.handle) { this.addon.destroy(this.handle); this.handle = null; } }

  _entry(group, binding) { const e = group.entries.find((x) => x.binding === binding); return e && (e.resource.buffer || e.resource.texture || e.resource); }

  _submit(lists) {
    try {
      for (const list of lists) {
        for (const c of list.cmds) {
          if (c.kind !== 'dispatch') continue;
          const n = c.group.entries.length;
          if (n === 1) continue; // UpdateVariables.wgsl: sample++ -- folded into trace()
          if (n !== 9) throw new Error(`compute pass with ${n} bindings: not the path tracer's bind group`);
          const bufs = [4, 5, 6, 7, 8].map((b) => this._entry(c.group, b));
          const key = bufs.map((b) => b.version).join(',') + ':' + bufs.map((b) => b.size).join(',');
          if (this.uploaded !== key) { // main.js:147-393 happened since the last dispatch
            const [camera, cie, spectra, lights, primitives] = bufs.map((b) => b.data);
            this.addon.uploadScene(this.handle, primitives, lights, spectra, cie, camera);
            this.addon.buildAccel(this.handle, this.accel);
            this.uploaded = key;
            const cam = new Float32Array(camera);
            const w = Math.trunc(cam[11]), h = Math.trunc(cam[12]);
            if (c.x !== Math.ceil(w / 8) || c.y !== Math.ceil(h / 8)) throw new Error('dispatch size does not cover the camera viewport in 8x8 workgroups');
          }
          this.addon.trace(this.handle, 1); // UpdateVariables + ComputeShader for one frame
          this.framebuffer = this._entry(c.group, 0);
        }
      }
      if (this.framebuffer && this.framebuffer.pixels) { // b0, the rgba8 storage texture
        this.framebuffer.pixels.set(this.addon.readRgba8(this.handle));
        if (this.canvas) this.canvas.pixels = this.framebuffer.pixels;
      }
    } catch (err) { // main.js:11-14: errors surface through "uncapturederror"
      const ls = this.listeners.uncapturederror || [];
      if (!ls.length) throw err;
      ls.forEach((fn) => fn({ error: err }));
    }
  }
}

// Install navigator.gpu, document.getElementById('canvas'), the GPU* constant tables and
// requestAnimationFrame (bounded: `frames` callbacks, then `onDone(device, canvas)`).
function install(target, options = {}) {
  const addon = require(path.join(__dirname, '..', 'addon', 'crt_napi.node'));
  const accel = options.accel === 'none' ? 0 : 1;
  let device = null;
  const canvas = { width: 0, height: 0, pixels: null, getContext: () => ({ configure() {}, getCurrentTexture: () => ({ createView: () => ({}) }) }) };
  target.GPUBufferUsage = GPUBufferUsage; target.GPUTextureUsage = GPUTextureUsage; target.GPUShaderStage = GPUShaderStage;
  target.navigator = { gpu: { requestAdapter: async () => ({ requestDevice: async () => { device = new Device(addon, options.device || 0, accel); device.canvas = canvas; return device; } }) } };
  target.document = { getElementById: () => canvas };
  let left = options.frames === undefined ? 1 : options.frames;
  target.requestAnimationFrame = (fn) => {
    if (left-- > 0) setImmediate(fn);
    else if (options.onDone) setImmediate(() => options.onDone(device, canvas));
  };
  return { get device() { return device; }, canvas, addon };
}

module.exports = { install, GPUBufferUsage, GPUTextureUsage, GPUShaderStage };
