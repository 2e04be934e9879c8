"""Model factory (neural_speech/models/__init__.py:7-17)."""


def create_model(name, hparams, **kw):
    if name == "taco2":
        from .tacotron2 import Tacotron2
        return Tacotron2(hparams, **kw)
    if name == "taco1":
        from .tacotron import Tacotron
        return Tacotron(hparams, **kw)
    if name == "wavenet":          # models/__init__.py:13-14: the full model (biases, scalar input, conditioning)
        from .wavenet import WaveNetModel
        return WaveNetModel(hparams, **kw)
    if name == "simple_wavenet":   # :15-16
        from .wavenet import SimpleWaveNet
        return SimpleWaveNet(hparams, **kw)
    raise Exception("Unknown model: " + name)
