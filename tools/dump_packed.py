"""Packed host buffers of a synthetic scene for the Node hosts: python tools/dump_packed.py <scene> <W> <H> <prefix>
writes <prefix>.{primitives,lights,camera,spectra,cie}.bin (the layouts of include/crt.h / src/main.js:147-393)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from computeraytracer_amd import scenes_synth
scene, W, H, pre = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ps = scenes_synth.SCENES[scene](W, H)
np.ascontiguousarray(ps.primitives).tofile(pre + '.primitives.bin')
np.ascontiguousarray(ps.lights).tofile(pre + '.lights.bin')
np.ascontiguousarray(ps.camera, np.float32).tofile(pre + '.camera.bin')
np.ascontiguousarray(ps.spectra, np.float32).tofile(pre + '.spectra.bin')
np.ascontiguousarray(ps.cie, np.float32).tofile(pre + '.cie.bin')
print(len(ps.primitives), 'primitives')
