"""ns_attention_post_bwd (the sums of the attention backward pass that no recurrence needs: dkeys, dv, dWcl;
attention.py:53-60 differentiated) against a float64 torch restatement, over shapes that cover a single staged block,
several blocks with a ragged last one, ragged lengths, more than one 64-position chunk, a filter width other than 7 and
unit counts that leave waves of a workgroup idle; repeated calls must give the same bits for dkeys (plain stores)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(N, S, Ti, Tia, A, kw, lengths, keys_t, q, align, de, wcl, v):
    half = (kw - 1) // 2
    dkeys = torch.zeros(N, A, Tia, dtype=torch.float64)
    dv = torch.zeros(A, dtype=torch.float64)
    dw = torch.zeros(kw, A, dtype=torch.float64)
    for n in range(N):
        L = int(lengths[n])
        for s in range(1, S + 1):
            ap = torch.zeros(Ti + kw, dtype=torch.float64)          # align_{s-1}[t + k - half] at index t + k
            ap[half:half + Ti] = align[n, s - 1, :Ti]
            loc = torch.stack([ap[k:k + Ti] for k in range(kw)], 1) @ wcl          # [Ti, A]
            x = keys_t[n, :, :Ti].t() + q[n, s][None, :] + loc
            th = torch.tanh(x)
            d = de[n, s, :Ti].clone()
            d[L:] = 0
            dpre = d[:, None] * v[None, :] * (1 - th * th)
            dpre[L:] = 0
            dkeys[n, :, :Ti] += dpre.t()
            dv += (d[:, None] * th)[:L].sum(0)
            for k in range(kw):
                dw[k] += (ap[k:k + Ti, None] * dpre).sum(0)
    return dkeys, dv, dw


@pytest.mark.parametrize("shape", [(33, 2, 9, 64, 7), (3, 8, 70, 64, 7), (2, 21, 130, 40, 7), (4, 9, 33, 24, 5),
                                   (2, 17, 64, 256, 7), (1, 1, 5, 8, 3)])
def test_attention_post_bwd_matches_float64(dev, shape):
    from nspeech_amd import ops
    N, S, Ti, A, kw = shape
    Tia = (Ti + 7) // 8 * 8
    g = torch.Generator().manual_seed(sum(shape))
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    lengths = torch.randint(max(1, Ti // 2), Ti + 1, (N,), generator=g).to(torch.int32)
    lengths[0] = Ti
    keys_t, q = rnd(N, A, Tia), rnd(N, S + 1, A)
    align = torch.softmax(rnd(N, S + 1, Tia) * 2, -1)
    de, wcl, v = rnd(N, S + 1, Tia) * 0.1, rnd(kw, A) * 0.5, rnd(A)
    want = _ref(N, S, Ti, Tia, A, kw, lengths, keys_t, q, align, de, wcl, v)
    f = lambda t: t.to("cuda:0", torch.float32).contiguous()
    args = [f(keys_t), f(q), f(align), f(de), f(wcl), f(v)]
    ln = lengths.to("cuda:0")
    outs = []
    for _ in range(3):
        dk = torch.full((N, A, Tia), float("nan"), device="cuda:0")
        dv = torch.zeros(A, device="cuda:0")
        dw = torch.zeros(kw, A, device="cuda:0")
        ops.attention_post_bwd(N, S, Ti, Tia, A, kw, ln, *args, dk, dv, dw)
        torch.cuda.synchronize()
        outs.append((dk.cpu(), dv.cpu(), dw.cpu()))
    dk, dv, dw = outs[0]
    # positions past Ti inside the padded row are written as zeros
    assert torch.isfinite(dk).all()
    for got, ref in ((dk, want[0]), (dv, want[1]), (dw, want[2])):
        err = (got.double() - ref).abs().max().item()
        assert err <= 2e-5 * ref.abs().max().item() + 1e-7, (shape, err, ref.abs().max().item())
    for o in outs[1:]:
        assert torch.equal(o[0], dk)
        assert (o[1] - dv).abs().max() <= 1e-5 * dv.abs().max() and (o[2] - dw).abs().max() <= 1e-5 * dw.abs().max()
