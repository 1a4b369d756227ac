"""configs/models/backbone/sparse_resnet18.yaml -> `_target_: det3d.models.backbones.sparse_resnet.SparseResNet` (masked-dense stand-in)."""
from pillarnext_amd.models import SparseResNet  # noqa: F401
