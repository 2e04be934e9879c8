"""The packed-fp32 operand-select hazard of MI355X (profiles/r04_determinism.txt item 4, profiles/r04_pk_opsel_probe.txt)
guarded ON THE GPU: tests/test_isa_guard_cpu.py keeps the src1 low-lane select out of the library's assembly, these two tests
check what that regex cannot see - that the workaround still holds on the box and toolchain the suite runs on.

1. attn_post_kernel's dWcl (the sums of the location-filter gradient over all decoder steps, attention.py:41-43 backward)
   at the benchmark shape, in the model's own backward pass, i.e. BESIDE the queued weight-gradient products on the
   second stream - the situation in which the odd filter taps once came out wrong by 0.15 % - against a float64
   restatement of the same sums from the same history buffers.
2. The probe itself (profiles/tools/pk_opsel_probe.hip, compiled here with hipcc): every packed-fp32 operand form the
   library may still contain - everything except op_sel's src1 select for the low lane - evaluates exactly, 3e9 times
   per form, while MFMA waves of another kernel share the CUs."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest
import torch

from util import make_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dwcl_float64(m, N, S, Ti, Tia, A):
    B = m._bufs
    ov = m._o("decoder/attention/attention_v")
    kt = B["dec_keys_t"][:N * A * Tia].view(N, A, Tia).double().cpu()
    q = B["dec_q"][:N * (S + 1) * A].view(N, S + 1, A).double().cpu()
    al = B["dec_al"][:N * (S + 1) * Tia].view(N, S + 1, Tia).double().cpu()
    de = B["d_energy"][:N * (S + 1) * Tia].view(N, S + 1, Tia).double().cpu()
    wcl = m.tsh["wcl"][:7 * A].view(7, A).double().cpu()
    v = m.flat_p[ov:ov + A].double().cpu()
    L = m.input_lengths.cpu().numpy()
    dw = torch.zeros(7, A, dtype=torch.float64)
    for n in range(N):
        ap = torch.zeros(S + 1, Ti + 6, dtype=torch.float64)
        ap[:, 3:3 + Ti] = al[n, :, :Ti]
        win = torch.stack([ap[:S, k:k + Ti] for k in range(7)], 2)                  # [S, Ti, 7]: align[s-1][t + k - 3], s = 1 .. S
        x = kt[n, :, :Ti].t()[None] + q[n, 1:, None, :] + win @ wcl                 # [S, Ti, A]
        d = de[n, 1:, :Ti].clone()
        d[:, L[n]:] = 0
        dpre = d[:, :, None] * v[None, None, :] * (1 - torch.tanh(x) ** 2)
        dw += torch.einsum("stk,sta->ka", win, dpre)
    return dw


def test_dwcl_beside_the_queued_weight_gradients_matches_float64(dev):
    from nspeech_amd import hparams as H
    from nspeech_amd.models import create_model
    hp = H.load("taco2")
    N, Ti, To = 32, 160, 1000
    A, S = hp.attention_dim, To // hp.outputs_per_step
    inputs, lengths, mel, lin = make_batch(hp, N, Ti, To, seed=17)
    m = create_model("taco2", hp, device="cuda:0", dtype="mixed", seed=7)
    assert m.overlap_wgrads                    # the weight-gradient products DO run on the second stream beside the post-pass
    worst = 0.0
    for _ in range(3):                         # the misread was intermittent (~1.3 per million evaluations): three passes
        m.initialize(inputs, lengths, None, mel, lin)
        m.backward()
        torch.cuda.synchronize()
        m.check_status()
        got = m._bufs["d_wcl"][:7 * A].view(7, A).double().cpu()
        ref = _dwcl_float64(m, N, S, Ti, Ti, A)
        err = (got - ref).abs().max(1).values / ref.abs().max()
        worst = max(worst, float(err.max()))
        # fp32 sums of 200 x 32 x 160 terms: 1e-6 measured; the hazard showed as 1.5e-3 in the odd taps only
        assert float(err.max()) < 2e-5, err.numpy()
    print("dWcl beside the queued weight gradients: worst tap error %.2e of the largest entry" % worst)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not on this box")
def test_safe_packed_forms_stay_exact_beside_mfma_waves(dev, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "pk_opsel_probe")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(ROOT, "profiles", "tools", "pk_opsel_probe.hip")])
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout
    sections = re.split(r"^victim ", out, flags=re.M)[1:]
    assert len(sections) == 3, out
    unsafe = re.compile(r"op_sel:\[0,1")       # the low lane takes the HIGH half of src1: the one form the library must not contain
    seen_safe = 0
    for sec in sections:
        head, *rows = sec.strip().split("\n")
        for row in rows:
            form, count = row.rsplit(None, 1)
            if unsafe.search(form):
                continue                       # (still misreads beside MFMA waves on the boxes measured; not asserted either way)
            seen_safe += 1
            assert int(count) == 0, "%s: %s wrong results %s" % (head.split(":")[0], count, form.strip())
    assert seen_safe == 3 * 11, seen_safe
    # the library's kernels contain none of the unsafe forms: tests/test_isa_guard_cpu.py (CPU suite, same build flags)
