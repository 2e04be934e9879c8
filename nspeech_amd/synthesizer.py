"""Inference wrapper with the reference's surface (neural_speech/synthesizer.py:9-54):
Synthesizer(hparams).load(checkpoint_path, model_name); .synthesize(text, speaker_id) ->
(wav, mel[T,80], lin[T,1025]).  The waveform is Griffin-Lim of the linear output, then
inv_preemphasis and find_endpoint, exactly the reference's order (synthesizer.py:30,52-53)."""
import numpy as np
import torch

from . import hparams as hparams_mod
from .models import create_model
from .utils import audio
from .utils.text import text_to_sequence


class Synthesizer(object):
    def __init__(self, hparams, dtype="mixed", device="cuda:0"):
        self.hparams = hparams
        self.dtype = dtype
        self.device = device
        self.model = None

    def load(self, checkpoint_path, model_name="taco2"):
        print("Constructing model: %s" % model_name)
        hparams_mod.set_hparams(self.hparams)
        self.model = create_model(model_name, self.hparams, device=self.device, dtype=self.dtype)
        if checkpoint_path is not None:
            print("Loading checkpoint: %s" % checkpoint_path)
            from .utils import tf_bundle
            if tf_bundle.is_bundle(checkpoint_path):       # a TensorFlow checkpoint prefix (model.ckpt-N.index + data)
                tf_bundle.load_into_model(self.model, checkpoint_path)
            else:
                self.model.load_state_dict(torch.load(checkpoint_path, map_location="cpu", weights_only=True))
        return self

    def synthesize(self, text, speaker_id=0):
        cleaner_names = [x.strip() for x in self.hparams.cleaners.split(",")]
        seq = text_to_sequence(text, cleaner_names)
        inputs = np.asarray([seq], dtype=np.int32)
        lengths = np.asarray([len(seq)], dtype=np.int32)
        m = self.model
        m.initialize(inputs, lengths, np.asarray([speaker_id], dtype=np.int32))
        wav = audio.inv_spectrogram_tensorflow(m.linear_outputs[0].contiguous())
        mel = m.mel_outputs[0].float().cpu().numpy()
        lin = m.linear_outputs[0].float().cpu().numpy()
        m.check_status()        # after the host copies (stream synchronised): a persistent BiLSTM kernel that gave up
                                # on an exchange leaves invalid outputs behind - never vocode those silently
        wav = audio.inv_preemphasis(wav.cpu().numpy())
        wav = wav[:audio.find_endpoint(wav)]
        return wav, mel, lin
