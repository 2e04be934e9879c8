import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from util import make_batch
from nspeech_amd import hparams as H
from nspeech_amd.models import create_model
hp = H.load("taco2")
N, Ti, To = 32, 160, 1000
inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=17)
m = create_model("taco2", hp, device="cuda:0", dtype="mixed", seed=7)
names = ("dec_al", "dec_q", "dec_keys_t", "d_energy", "d_wcl", "dec_al_t", "keys")
snaps = []
for overlap in (False, True, True):
    m.overlap_wgrads = overlap
    m.initialize(inputs, lengths, None, mel, lin)
    torch.cuda.synchronize()
    fwd = {k: m._bufs[k].clone() for k in names if k in m._bufs}
    m.backward()
    torch.cuda.synchronize()
    bwd = {k: m._bufs[k].clone() for k in names if k in m._bufs}
    wcl = m.tsh["wcl"].clone()
    snaps.append((fwd, bwd, wcl))
for i in (1, 2):
    f0, b0, w0 = snaps[0]
    f1, b1, w1 = snaps[i]
    print("run", i, "forward buffers differing:", [(k, float((f0[k].float() - f1[k].float()).abs().max())) for k in f0 if not torch.equal(f0[k], f1[k])])
    print("       after backward differing:", [(k, float((b0[k].float() - b1[k].float()).abs().max())) for k in b0 if not torch.equal(b0[k], b1[k])])
    print("       changed by its own backward:", [(k, float((f1[k].float() - b1[k].float()).abs().max())) for k in f1 if not torch.equal(f1[k], b1[k])])
    print("       wcl equal", torch.equal(w0, w1))
