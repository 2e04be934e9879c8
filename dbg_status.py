import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from nspeech_amd import hparams as hparams_mod
from nspeech_amd.synthesizer import Synthesizer
from nspeech_amd.utils import audio
from nspeech_amd.utils.text import text_to_sequence
hp = hparams_mod.load("taco2")
hp.max_iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
use_graph = (sys.argv[2] != "nograph") if len(sys.argv) > 2 else True
hparams_mod.set_hparams(hp)
text = "Turn left on {HH AW1 S S T AH0 N} Street, then {R AY1 T} at the {L AY1 T}."
synth = Synthesizer(hp, dtype="mixed").load(None, "taco2")
m = synth.model
m.use_graph = use_graph
seq = text_to_sequence(text, ["english_cleaners"])
inputs = np.asarray([seq], dtype=np.int32); lengths = np.asarray([len(seq)], dtype=np.int32)

def who(ptr):
    for kk, b in m._bufs.items():
        if torch.is_tensor(b) and b.data_ptr() <= ptr < b.data_ptr() + b.numel() * b.element_size():
            return "%s+%d" % (kk, ptr - b.data_ptr())
    for kk, b in getattr(m, "tsh", {}).items():
        if torch.is_tensor(b) and b.data_ptr() <= ptr < b.data_ptr() + b.numel() * b.element_size():
            return "tsh:%s+%d" % (kk, ptr - b.data_ptr())
    return "?"

for it in range(4):
    m.initialize(inputs, lengths, np.asarray([0], np.int32))
    torch.cuda.synchronize()
    w = m._status_words[("expl", "fwd")]
    q = w[:32].view(torch.int64).tolist()
    print(it, "graph" if use_graph else "nograph", "work ptr %x" % w.data_ptr(), [("%x" % v, who(v)) for v in q[:8]])
    sys.stdout.flush()
