"""tdvc-mi355x: MI355X-native (gfx950) implementation of TD-VC-GAN's G+D train-step path.

The HIP library (csrc/ -> libtdvc_hip.so) is loaded lazily on the first operator call and
the load fails loudly when it is missing: there is no CPU or ATen fallback in the product path.
"""
__version__ = '0.1.0'
