"""Model factory (neural_speech/models/__init__.py:7-17)."""


def create_model(name, hparams, **kw):
    if name == "taco2":
        from .tacotron2 import Tacotron2
        return Tacotron2(hparams, **kw)
    if name == "taco1":
        from .tacotron import Tacotron
        return Tacotron(hparams, **kw)
    raise Exception("Unknown model: " + name)
