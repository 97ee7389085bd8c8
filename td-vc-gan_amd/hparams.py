"""Reads the reference's multi-document YAML configs unchanged (util/hparams.py:17-68 semantics:
all documents merged into one nested dict with attribute access)."""
import yaml


class Dotdict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__

    def __init__(self, dct=None):
        super().__init__()
        for k, v in (dct or {}).items():
            self[k] = Dotdict(v) if hasattr(v, 'keys') else v


class HParam(Dotdict):
    def __init__(self, file):
        merged = {}
        with open(file) as f:
            for doc in yaml.safe_load_all(f):
                merged.update(doc or {})
        super().__init__(merged)
