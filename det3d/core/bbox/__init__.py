from . import box_torch_ops  # noqa: F401
