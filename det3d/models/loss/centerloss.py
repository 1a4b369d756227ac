"""Losses used by CenterHead.loss."""
from pillarnext_amd.losses import FastFocalLoss, IouLoss, IouRegLoss, RegLoss  # noqa: F401
