"""ISA guard: no kernel of libnspeech_hip.so may contain a packed-fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 /
v_pk_add_f32) whose op_sel selects the HIGH half of src1 for the low lane.  On MI355X that operand form returns a wrong
result about once per million executions while MFMA waves of another kernel share the CU - never alone, never beside a
VALU-only kernel, never for src0 / src2 selects or any op_sel_hi form (profiles/tools/pk_opsel_probe.hip; found through
attn_post_kernel, whose location-filter gradient came out 0.15 % wrong beside the queued weight-gradient products,
profiles/r04_determinism.txt item 4).  The compiler picks operand selects freely when it folds a vector shuffle into a packed
instruction, so the check runs on the device assembly of every source, compiled with the flags of the real build (no GPU
needed: hipcc cross-compiles)."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def unsafe_packed_ops(asm_text):
    """[(kernel symbol, instruction)] of packed-fp32 instructions with op_sel bit 1 (src1) set."""
    bad, cur = [], None
    for line in asm_text.split("\n"):
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not line.startswith("."):
            cur = m.group(1)
        if re.search(r"\bv_pk_(fma|mul|add)_f32\b", line):
            sel = re.search(r"op_sel:\[(\d),(\d)", line)
            if sel and sel.group(2) == "1":
                bad.append((cur, line.strip()))
    return bad


def test_the_scanner_sees_the_form():
    text = "k1:\n  v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[0,1,0]\n  v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]\n" \
           "k2:\n  v_pk_mul_f32 v[6:7], v[2:3], v[4:5] op_sel:[1,1] op_sel_hi:[0,1]\n  v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel_hi:[1,0,1]\n"
    assert [k for k, _ in unsafe_packed_ops(text)] == ["k1", "k2"]


def test_no_kernel_selects_the_high_half_of_src1():
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, NS_ASM_DIR=d)
        out = subprocess.run(["bash", os.path.join(ROOT, "nspeech_amd", "csrc", "build.sh")], env=env, capture_output=True, text=True,
                             timeout=1500)
        assert out.returncode == 0, out.stderr[-2000:]
        files = sorted(f for f in os.listdir(d) if f.endswith(".s"))
        assert len(files) >= 10, files
        bad, packed = {}, 0
        for f in files:
            text = open(os.path.join(d, f)).read()
            packed += len(re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", text))
            hits = unsafe_packed_ops(text)
            if hits:
                bad[f] = hits[:5] + ([("...", "%d in all" % len(hits))] if len(hits) > 5 else [])
        assert packed > 1000          # the scan saw the library's packed arithmetic (it is not looking at empty files)
        assert not bad, bad
