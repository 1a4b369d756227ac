"""configs/models/reader/mvf_encoder.yaml -> `_target_: det3d.models.readers.mvf_encoder.MVFFeatureNet`."""
from pillarnext_amd.mvf_encoder import CylinderNet, MVFFeatureNet, PillarVoxelNet, PointNet, SingleView  # noqa: F401
