"""configs/models/reader/voxel_encoder.yaml -> `_target_: det3d.models.readers.voxel_encoder.VoxelFeatureNet`."""
from pillarnext_amd.voxel_encoder import DynamicVoxelEncoder, VoxelFeatureNet, VoxelNet  # noqa: F401
