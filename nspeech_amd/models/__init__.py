"""Model factory (neural_speech/models/__init__.py:7-17)."""


def create_model(name, hparams, **kw):
    if name == "taco2":
        from .tacotron2 import Tacotron2
        return Tacotron2(hparams, **kw)
    if name == "taco1":
        from .tacotron import Tacotron
        return Tacotron(hparams, **kw)
    if name in ("simple_wavenet", "wavenet"):
        # models/__init__.py:13-16 maps both names; the full WaveNetModel's extra options (scalar input, biases,
        # conditioning) are off in the shipped wavenet.yaml, where the two graphs coincide
        from .wavenet import SimpleWaveNet
        return SimpleWaveNet(hparams, **kw)
    raise Exception("Unknown model: " + name)
