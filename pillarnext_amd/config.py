"""The slice of Hydra this hot path needs: `${a.b[1]}` interpolation and `_target_` instantiation of a resolved config tree
(hydra.utils.instantiate is what tools/train.py:53 / tools/test.py:49 call on `cfg.model`).  hydra-core / omegaconf are not in this
image; when they are importable, use them -- the semantics below are theirs for the keys the PillarNeXt configs use:
`_target_`, `_recursive_` (false: nested configs are passed on as plain dicts), `_partial_`, keyword overrides."""
import functools
import importlib
import re

import yaml

_INTERP = re.compile(r"\$\{([^}]+)\}")


def load(path):
    with open(path) as f:
        return resolve(yaml.safe_load(f))


def _lookup(root, expr):
    node = root
    for part in re.findall(r"[^.\[\]]+|\[\d+\]", expr):
        node = node[int(part[1:-1])] if part.startswith("[") else node[part]
    return node


def resolve(cfg, root=None):
    """Replace `${path}` strings (whole-value interpolations, the only form the reference's YAMLs use) by the node they name."""
    root = cfg if root is None else root
    if isinstance(cfg, dict):
        return {k: resolve(v, root) for k, v in cfg.items()}
    if isinstance(cfg, list):
        return [resolve(v, root) for v in cfg]
    if isinstance(cfg, str):
        m = _INTERP.fullmatch(cfg.strip())
        if m:
            return resolve(_lookup(root, m.group(1)), root)
    return cfg


def _locate(target):
    mod, name = target.rsplit(".", 1)
    return getattr(importlib.import_module(mod), name)


def instantiate(cfg, **overrides):
    """hydra.utils.instantiate for a resolved dict: build cfg['_target_'](**kwargs), instantiating nested `_target_` nodes first
    unless `_recursive_: false`."""
    if not isinstance(cfg, dict) or "_target_" not in cfg:
        return cfg
    cfg = dict(cfg)
    cfg.update(overrides)
    fn = _locate(cfg.pop("_target_"))
    recursive = cfg.pop("_recursive_", True)
    partial = cfg.pop("_partial_", False)
    kwargs = {k: (_instantiate_tree(v) if recursive else v) for k, v in cfg.items()}
    return functools.partial(fn, **kwargs) if partial else fn(**kwargs)


def _instantiate_tree(v):
    if isinstance(v, dict):
        return instantiate(v) if "_target_" in v else {k: _instantiate_tree(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_instantiate_tree(x) for x in v]
    return v
