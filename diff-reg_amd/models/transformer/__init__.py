"""`models.transformer`: Diff-Reg-4dmatch keeps GeometryAttentionLayer / RepositioningTransformer in a
module of this name (4D/models/transformer.py), while Diff-Reg-3dmatch has an unrelated (dead, RoITr)
package `models/transformer/` that models.backbone still imports (3D/models/backbone.py:4).  This package
serves both: it re-exports the accelerated classes and lets sub-modules resolve in the reference tree."""
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)

from models.transformero import GeometryAttentionLayer, RepositioningTransformer  # noqa: E402,F401
