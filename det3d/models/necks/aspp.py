"""configs/models/neck/aspp.yaml -> `_target_: det3d.models.necks.aspp.ASPPNeck`."""
from pillarnext_amd.models import ASPPNeck  # noqa: F401
