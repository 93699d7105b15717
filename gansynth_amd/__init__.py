"""gansynth_amd -- MI355X (gfx950) native hot path of skmhrk1209/GANSynth.

Python host mirroring the reference call surface (ops.py, networks.py, spectral_ops.py,
models.py) on top of libgansynth_hip.so (hand-written HIP kernels behind a C ABI,
include/gansynth_hip.h).  There is no CPU or torch fallback: importing the kernel layer
without the built library raises.
"""
__version__ = "0.1.0"
