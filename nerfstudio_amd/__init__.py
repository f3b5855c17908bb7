"""nerfstudio_amd — MI355X (gfx950) native volumetric-rendering core behind nerfstudio's Field / Encoding / Sampler /
Renderer plugin API (the nerfacto hot path).

The arithmetic lives in `libnsamd.so` (hand-written HIP, C ABI in include/nsamd.h); this package is the host-side
mirror of the reference's Python interface for that path: same class names, argument meaning and error behaviour
as `/root/reference/nerfstudio/{cameras/rays,field_components,fields,model_components}`.
"""
__version__ = "0.1.0"
