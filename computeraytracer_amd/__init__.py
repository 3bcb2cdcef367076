"""computeraytracer_amd -- MI355X (gfx950) drop-in for the path-trace compute
pass of Meryx/ComputeRayTracer (ComputeShader.wgsl + UpdateVariables.wgsl as
dispatched by src/main.js).  The compute path is libcrt.so (hand-written HIP,
C ABI in include/crt.h); this package is the thin Python host side."""
from .scene import PackedScene, cornell, load_scene, pack_scene  # noqa: F401

__all__ = ["PackedScene", "cornell", "load_scene", "pack_scene", "Renderer"]


def __getattr__(name):
    if name == "Renderer":          # lazy: importing the package must not need a GPU
        from .renderer import Renderer
        return Renderer
    raise AttributeError(name)
