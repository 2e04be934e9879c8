"""Hyper-parameters: audio.yaml <- train.yaml <- <model>.yaml merged into one HParams object.

Mirrors neural_speech/hparams/__init__.py:8-26 (load / get_hparams / debug_string) and the
part of tf.contrib.training.HParams the reference uses: attribute get/set (train.py:45),
.values(), .parse("a=1,b=[2,3]") typed by the existing entry (train.py:163)."""
import os
import re

import yaml

yaml_path = os.path.dirname(os.path.abspath(__file__))
_hparams = None


class HParams(object):
    def __init__(self, **kw):
        object.__setattr__(self, "_d", dict(kw))

    def __getattr__(self, k):
        d = object.__getattribute__(self, "_d")
        if k in d:
            return d[k]
        raise AttributeError(k)

    def __setattr__(self, k, v):
        self._d[k] = v

    def __contains__(self, k):
        return k in self._d

    def __getitem__(self, k):
        return self._d[k]

    def get(self, k, default=None):
        return self._d.get(k, default)

    def values(self):
        return dict(self._d)

    def parse(self, s):
        """'a=1,b=[2,3],c=x' -> typed by the current value; unknown names raise (as TF does)."""
        if not s:
            return self
        for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_]*)\s*=\s*(\[[^\]]*\]|[^,]*)", s):
            k, raw = m.group(1), m.group(2).strip()
            if k not in self._d:
                raise ValueError("Unknown hyperparameter: %s" % k)
            self._d[k] = _cast(raw, self._d[k])
        return self


def _cast(raw, like):
    if isinstance(like, bool):
        return raw.lower() in ("1", "true", "yes")
    if isinstance(like, int):
        return int(raw)
    if isinstance(like, float):
        return float(raw)
    if isinstance(like, list):
        items = [x.strip() for x in raw.strip("[]").split(",") if x.strip()]
        proto = like[0] if like else 0
        return [_cast(x, proto) for x in items]
    return raw.strip("\"'")


def debug_string(hp):
    values = hp.values()
    return "Hyperparameters:\n" + "\n".join("  %s: %s" % (n, values[n]) for n in sorted(values))


def load(model_type):
    global _hparams
    cfg = {}
    for name in ("audio", "train", model_type):
        with open(os.path.join(yaml_path, name + ".yaml")) as f:
            cfg.update(yaml.safe_load(f))
    _hparams = HParams(**cfg)
    return _hparams


def get_hparams():
    return _hparams


def set_hparams(hp):
    global _hparams
    _hparams = hp
    return hp
