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
snaps = []
names = ("d_wcl", "d_energy", "d_keys", "d_q", "d_ga", "d_f1", "d_p2", "d_hc", "d_values", "d_g1b", "d_g2b", "d_h1", "d_h2", "d_mel")
for overlap in (False, False, True, True):
    m.overlap_wgrads = overlap
    m.initialize(inputs, lengths, None, mel, lin)
    m.backward()
    torch.cuda.synchronize()
    snaps.append((m.flat_g.clone(), {k: m._bufs[k].clone() for k in names if k in m._bufs}))
for i in range(1, 4):
    g0, b0 = snaps[0]
    g1, b1 = snaps[i]
    bad = []
    for name, (off, shape) in m.layout.entries.items():
        n = int(np.prod(shape))
        if not torch.equal(g0[off:off + n], g1[off:off + n]):
            bad.append((name, float((g0[off:off + n] - g1[off:off + n]).abs().max()), float(g0[off:off+n].abs().max())))
    print("run", i, "grad tensors differing:", bad)
    print("   buffers differing:", [(k, float((b0[k].float() - b1[k].float()).abs().max())) for k in b0 if not torch.equal(b0[k], b1[k])])

# ---- where does d_wcl's difference come from: the parked partial sums or the finish?
from nspeech_amd import ops
parts = []
for overlap in (False, True, True):
    m.overlap_wgrads = overlap
    m.initialize(inputs, lengths, None, mel, lin)
    m.backward()
    torch.cuda.synchronize()
    part = list(ops._POST_PART.values())[0].clone()
    parts.append((part, m._bufs["d_wcl"].clone()))
A = hp.attention_dim
for i in (1, 2):
    pa, wa = parts[0]
    pb, wb = parts[i]
    nb = 32 * 3
    va = pa[:nb * 9 * A].view(nb, 9, A)
    vb = pb[:nb * 9 * A].view(nb, 9, A)
    d = (va - vb).abs()
    print("partials differ:", bool(d.max() > 0), "max", float(d.max()), "blocks with differences", torch.nonzero(d.amax(dim=(1, 2)) > 0).flatten().tolist()[:20],
          "k with differences", torch.nonzero(d.amax(dim=(0, 2)) > 0).flatten().tolist())
    host = va[:, 1:8].double().sum(0).float()
    print("finish vs host sum of the same partials: max", float((wa[:7 * A].view(7, A) - host).abs().max()), "; d_wcl a vs b", float((wa - wb).abs().max()))
