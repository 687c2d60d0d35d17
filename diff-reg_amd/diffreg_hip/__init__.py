"""diffreg_hip -- MI355X-native reverse-diffusion matching engine (host side).

`lib` (ctypes binding of libdiffreg_hip.so) is imported lazily so that CPU-only tooling
(synthetic generator, config handling) works on boxes without the built extension.
"""
__all__ = ["synth"]
__version__ = "0.1.0"
