"""Minimal Hydra-compatible configuration layer.

The reference drives everything through Hydra (``@hydra.main`` + ``hydra.utils.instantiate``,
train.py:135, generate.py:197); neither hydra nor omegaconf is installed here, so this module
provides the three things the hot path needs, with the same key names:

  * ``instantiate(cfg, *args, **kwargs)`` -- the ``_target_`` plugin boundary
    (``_recursive_`` / ``_convert_`` accepted and ignored: values are plain Python objects already).
    Targets under ``swift.`` (the reference's package) resolve to this package, so a saved
    ``.hydra/config.yaml`` of a reference run works unchanged;
  * ``compose(config_dir, overrides)`` -- defaults-list composition of the yaml tree under
    ``swift_amd/configs`` (group selection ``a=b`` / ``a/b=c``, ``override /x: y`` entries,
    ``# @package _global_`` files, dotted value overrides ``a.b.c=v``, ``${oc.env:X}`` and ``${key}``);
  * ``Cfg`` -- a dict with attribute access (``cfg.model.dim``), the only OmegaConf feature used.
"""
from __future__ import annotations

import importlib
import os
import re
from typing import Any, Optional

import yaml

TARGET_ALIASES = (("swift.", "swift_amd."),)


class Cfg(dict):
    """dict with attribute access; nested dicts are wrapped on the way in."""

    def __init__(self, *a, **k):
        super().__init__()
        for key, v in dict(*a, **k).items():
            self[key] = v

    @staticmethod
    def wrap(v):
        if isinstance(v, dict) and not isinstance(v, Cfg):
            return Cfg(v)
        if isinstance(v, (list, tuple)):
            return [Cfg.wrap(x) for x in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, Cfg.wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    def to_plain(self):
        def conv(v):
            if isinstance(v, dict):
                return {k: conv(x) for k, x in v.items()}
            if isinstance(v, list):
                return [conv(x) for x in v]
            return v
        return conv(self)


def resolve_target(path: str):
    for old, new in TARGET_ALIASES:
        if path.startswith(old):
            path = new + path[len(old):]
            break
    mod, name = path.rsplit(".", 1)
    return getattr(importlib.import_module(mod), name)


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


def instantiate(cfg, *args, **kwargs):
    """``hydra.utils.instantiate`` for the ``_target_`` convention (precond.py:123-131, train.py:212-220)."""
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    recursive = kwargs.pop("_recursive_", True)
    kwargs.pop("_convert_", None)
    params = {}
    for k, v in {**cfg, **kwargs}.items():
        if recursive and isinstance(v, dict) and "_target_" in v:
            v = instantiate(v)
        elif k != "model_config":
            v = _plain(v)
        params[k] = v
    return resolve_target(target)(*args, **params)


# ----------------------------------------------------------------------------- composition

_PKG_RE = re.compile(r"^#\s*@package\s+(\S+)", re.M)


def _load_yaml(path):
    with open(path) as f:
        text = f.read()
    pkg = _PKG_RE.search(text)
    return (_yaml_load(text) or {}), (pkg.group(1) if pkg else None)


def _merge(dst: dict, src: dict):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _place(root: dict, pkg_path: str, content: dict):
    node = root
    for p in pkg_path.strip("/").split("/"):
        if p:
            node = node.setdefault(p, {})
    return _merge(node, content)


def _compose_file(config_dir, group, name, root, choices, pkg_path=None):
    """Merge ``group/name.yaml`` into ``root``.

    Packages follow Hydra: a file lands at its group path unless a ``# @package`` header says otherwise, and
    every defaults entry of a file is packaged *relative to that file's package* -- so ``/loss: crps`` inside
    ``finetune/multistep.yaml`` ends up at ``finetune.loss`` (reference train.py:75-77 then hoists those keys).
    """
    path = os.path.join(config_dir, group.strip("/"), f"{name}.yaml")
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    body, header = _load_yaml(path)
    if header == "_global_":
        here = ""
    elif header:
        here = header.replace(".", "/")
    else:
        here = group.strip("/") if pkg_path is None else pkg_path
    defaults = body.pop("defaults", [])
    self_done = False
    for d in defaults:
        if d == "_self_":
            _place(root, here, body)
            self_done = True
            continue
        if isinstance(d, str):
            d = {d: None}
        (k, v), = d.items()
        if k.startswith("override "):
            continue  # recorded by _collect_overrides before composition
        if v is None and "/" not in k and not os.path.isdir(os.path.join(config_dir, group.strip("/"), k)):
            _compose_file(config_dir, group, k, root, choices, pkg_path=here)  # plain file of the same group
            continue
        sub = (k if k.startswith("/") else (group.strip("/") + "/" + k if group.strip("/") else k)).strip("/")
        child_pkg = (here + "/" if here else "") + k.strip("/")
        choice = choices.get(_choice_key(sub, child_pkg), v)
        if choice is None or choice == "null":
            continue
        _compose_file(config_dir, sub, choice, root, choices, pkg_path=child_pkg)
    if not self_done:
        _place(root, here, body)


def _choice_key(group: str, pkg: str) -> str:
    """Hydra keys a defaults entry by group AND destination package (``get_override_key``): ``override /optimizer: muon`` or
    a command-line ``optimizer=x`` reaches the entry packaged at ``optimizer`` only; the ``/optimizer: adamw`` that
    ``finetune/multistep.yaml`` packages at ``finetune.optimizer`` keeps its own default unless addressed as
    ``optimizer@finetune.optimizer=x``."""
    group, pkg = group.strip("/"), pkg.strip("/")
    return group if pkg == group else f"{group}@{pkg.replace('/', '.')}"


def _collect_overrides(config_dir, group, name, choices):
    path = os.path.join(config_dir, group.strip("/"), f"{name}.yaml")
    if not os.path.exists(path):
        return
    body, _ = _load_yaml(path)
    for d in body.get("defaults", []):
        if isinstance(d, dict):
            (k, v), = d.items()
            if k.startswith("override "):
                choices.setdefault(k[len("override "):].strip().strip("/"), v)
            elif v is not None:
                sub = (k if k.startswith("/") else (group.strip("/") + "/" + k if group.strip("/") else k)).strip("/")
                _collect_overrides(config_dir, sub, choices.get(sub, v), choices)


class _Loader(yaml.SafeLoader):
    """SafeLoader with the YAML 1.2 float grammar: PyYAML (YAML 1.1) reads ``1e-4`` / ``3e-4`` as strings, Hydra reads floats."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"^[-+]?(?:(?:[0-9][0-9_]*)?\.[0-9_]+(?:[eE][-+]?[0-9]+)?|[0-9][0-9_]*\.?(?:[eE][-+]?[0-9]+)|\.(?:inf|Inf|INF)|"
               r"[-+]\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$"),
    list("-+0123456789."))


def _yaml_load(text):
    return yaml.load(text, Loader=_Loader)


def _parse_value(s: str):
    if len(s) > 1 and s.isdigit() and s[0] == "0":  # run ids like 000 / 001 stay strings (resume=000)
        return s
    if re.fullmatch(r"\d{8}_\d{6}", s):  # a run directory named by its start time (HYDRA_RUN_ID unset); YAML would read an integer
        return s
    try:
        return _yaml_load(s)
    except yaml.YAMLError:
        return s


def _interp(root: dict):
    pat = re.compile(r"\$\{([^}]+)\}")

    def lookup(key):
        if key.startswith("oc.env:"):
            name, _, default = key[len("oc.env:"):].partition(",")
            return os.environ.get(name, default)
        node = root
        for p in key.split("."):
            node = node[p]
        return node

    def walk(v):
        if isinstance(v, dict):
            return {k: walk(x) for k, x in v.items()}
        if isinstance(v, list):
            return [walk(x) for x in v]
        if isinstance(v, str) and "${" in v:
            m = pat.fullmatch(v)
            if m:
                return lookup(m.group(1))
            return pat.sub(lambda mm: str(lookup(mm.group(1))), v)
        return v

    return walk(root)


def compose(config_dir: str, config_name: str = "train", overrides=()) -> Cfg:
    """Compose ``config_name`` with Hydra-style command-line overrides."""
    choices, values = {}, []
    for o in overrides:
        k, _, v = o.partition("=")
        k = k.lstrip("+")
        if os.path.isdir(os.path.join(config_dir, k.partition("@")[0])):
            choices[k] = v
        else:
            values.append((k, v))
    # overrides declared inside selected files (e.g. "override /optimizer: muon") apply unless the CLI chose
    _collect_overrides(config_dir, "", config_name, choices)
    for g, n in list(choices.items()):
        if n not in (None, "null"):
            _collect_overrides(config_dir, g, n, choices)
    root: dict = {}
    _compose_file(config_dir, "", config_name, root, choices)
    for k, v in values:
        node = root
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = _parse_value(v)
    root.pop("hydra", None) if False else None
    return Cfg(_interp(root))


def load_saved(path: str) -> Cfg:
    """Load a composed ``.hydra/config.yaml`` (what generate.py:161 and train.py:57 read back)."""
    with open(path) as f:
        return Cfg(_yaml_load(f.read()))


def to_yaml(cfg) -> str:
    return yaml.safe_dump(_plain(cfg), sort_keys=False)
