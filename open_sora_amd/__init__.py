"""MI355X-native (gfx950) Open-Sora denoise path: hand-written HIP kernels behind the reference's
module API.  Importing the compute modules requires the in-tree libosk_hip.so (no CPU fallback):
    from open_sora_amd import mmdit      # MMDiTModel / Flux / block processors
    from open_sora_amd import sampling   # schedule, pack/unpack, CFG Euler sampler (host logic)
"""
__version__ = "0.1.0"
