"""configs/models/detectors/pillarnet18_aspp.yaml -> `_target_: det3d.models.detectors.single_stage.SingleStageDetector`."""
from pillarnext_amd.models import SingleStageDetector  # noqa: F401
