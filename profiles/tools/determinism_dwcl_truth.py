import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from util import make_batch
from nspeech_amd import hparams as H, ops
from nspeech_amd.models import create_model
hp = H.load("taco2")
N, Ti, To = 32, 160, 1000
A = hp.attention_dim
inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=17)
m = create_model("taco2", hp, device="cuda:0", dtype="mixed", seed=7)
S = To // hp.outputs_per_step
Tia = Ti
def standalone():
    B = m._bufs
    dk = torch.zeros(N * A * Tia, device="cuda:0"); dv = torch.zeros(A, device="cuda:0"); dw = torch.zeros(7, A, device="cuda:0")
    vv = m.flat_p[m._o("decoder/attention/attention_v"):m._o("decoder/attention/attention_v") + A].clone()
    ops.attention_post_bwd(N, S, Ti, Tia, A, 7, m.input_lengths, B["dec_keys_t"], B["dec_q"], B["dec_al"], B["d_energy"], m.tsh["wcl"], vv, dk, dv, dw)
    torch.cuda.synchronize()
    return dw.clone()
def truth():
    B = m._bufs
    kt = B["dec_keys_t"][:N * A * Tia].view(N, A, Tia).double().cpu()
    q = B["dec_q"][:N * (S + 1) * A].view(N, S + 1, A).double().cpu()
    al = B["dec_al"][:N * (S + 1) * Tia].view(N, S + 1, Tia).double().cpu()
    de = B["d_energy"][:N * (S + 1) * Tia].view(N, S + 1, Tia).double().cpu()
    wcl = m.tsh["wcl"][:7 * A].view(7, A).double().cpu()
    v = m.flat_p[m._o("decoder/attention/attention_v"):m._o("decoder/attention/attention_v") + A].double().cpu()
    L = m.input_lengths.cpu().numpy()
    dw = torch.zeros(7, A, dtype=torch.float64)
    for n in range(N):
        ap = torch.zeros(S + 1, Ti + 6, dtype=torch.float64)
        ap[:, 3:3 + Ti] = al[n, :, :Ti]
        win = torch.stack([ap[:, k:k + Ti] for k in range(7)], 2)       # [S+1, Ti, 7]  align_{s}[t + k - 3]
        for s in range(1, S + 1):
            loc = win[s - 1] @ wcl                                        # [Ti, A]
            x = kt[n, :, :Ti].t() + q[n, s][None, :] + loc
            th = torch.tanh(x)
            d = de[n, s, :Ti].clone(); d[L[n]:] = 0
            dpre = d[:, None] * v[None, :] * (1 - th * th)
            dw += win[s - 1].t() @ dpre
    return dw
for overlap in (False, True, True):
    m.overlap_wgrads = overlap
    m.initialize(inputs, lengths, None, mel, lin)
    m.backward()
    torch.cuda.synchronize()
    inmodel = m._bufs["d_wcl"][:7 * A].view(7, A).clone()
    sa = standalone()
    tr = truth()
    e_im = (inmodel.double().cpu() - tr).abs()
    e_sa = (sa.double().cpu() - tr).abs()
    print("overlap", overlap, "| |truth| max %.3e" % tr.abs().max().item(), "| in-model err max %.3e per tap" % e_im.max().item(), e_im.max(1).values.numpy(),
          "| standalone err max %.3e per tap" % e_sa.max().item(), e_sa.max(1).values.numpy())
