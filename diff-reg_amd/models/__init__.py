"""Drop-in overlay of the reference's `models` package for the reverse-diffusion matching path.

Put this directory's parent AHEAD of a Diff-Reg-3dmatch / Diff-Reg-4dmatch checkout on sys.path:
`models.pipeline`, `models.matching`, `models.procrustes`, `models.transformer(o)` and
`models.position_encoding` then resolve here (HIP kernels through libdiffreg_hip.so), everything else
(`models.backbone`, `models.blocks`, `models.loss`, ...) still resolves in the reference tree.
"""
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)
