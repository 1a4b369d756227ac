"""configs/models/reader/pillar_encoder.yaml -> `_target_: det3d.models.readers.pillar_encoder.PillarFeatureNet`."""
from pillarnext_amd.reader import PFNLayer, PillarFeatureNet, PillarNet  # noqa: F401
