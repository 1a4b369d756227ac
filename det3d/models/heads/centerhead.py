"""configs/models/head/centerhead.yaml -> `_target_: det3d.models.heads.centerhead.CenterHead`."""
from pillarnext_amd.models import CenterHead, SepHead  # noqa: F401
