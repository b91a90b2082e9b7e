"""flexynesis_amd -- MI355X (gfx950) native training engine for the flexynesis multi-omic
encoder/fusion hot path.  Hand-written HIP kernels behind a C ABI (include/fxhip.h), Python host code
mirroring the reference's model-class / dataset interface.  There is no CPU fallback: importing the
kernels without libfxhip.so, or calling them on non-GPU tensors, raises."""
__version__ = "0.1.0"
