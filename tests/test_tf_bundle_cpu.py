"""TensorFlow checkpoint container: CRC-32C and varint known answers, table / bundle round trips over
several blocks, corruption detection, and the mapping of TF-style variable names onto the Tacotron-2 layout.
(Unverified against files written by TensorFlow itself: none is available here.)"""
import os

import numpy as np
import pytest

from nspeech_amd.utils import tf_bundle as B


def test_crc32c_and_varint_known_answers():
    assert B.crc32c(b"123456789") == 0xE3069283                       # the CRC-32C check value
    assert B.crc32c(b"") == 0 and B.crc32c(b"\x00" * 32) == 0x8A9136AA  # RFC 3720 B.4: 32 zero bytes
    assert B.crc32c(bytes(range(32))) == 0x46DD794E                   # RFC 3720 B.4: 0x00..0x1f
    for c in (0, 1, 0xE3069283, 0xFFFFFFFF):
        assert B.unmask_crc(B.mask_crc(c)) == c and (c == 0 or B.mask_crc(c) != c)
    for v in (0, 1, 127, 128, 300, 2 ** 32 + 5, 2 ** 63 - 1):
        enc = B._put_varint(v)
        assert B._get_varint(enc, 0) == (v, len(enc))
    assert B._put_varint(300) == b"\xac\x02"


def test_table_round_trip_prefix_compression_and_blocks(tmp_path):
    items = [(("model/inference/layer_%03d/kernel" % i).encode(), os.urandom(5 + i % 7)) for i in range(150)]
    items.append((b"", b"header"))
    path = str(tmp_path / "t.index")
    B.write_table(path, items, entries_per_block=16)
    got = B.read_table(path)
    assert got == sorted(items)
    raw = bytearray(open(path, "rb").read())
    assert int.from_bytes(raw[-8:], "little") == B.MAGIC and len(raw) > 48
    raw[20] ^= 0x40                                                    # flip a bit inside the first data block
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        B.read_table(path)


def test_bundle_round_trip_dtypes_shapes_and_checksums(tmp_path):
    rs = np.random.RandomState(0)
    t = {"model/inference/embedding/embedding": rs.randn(149, 8).astype(np.float32),
         "model/inference/decoder/x/bias": rs.randn(7).astype(np.float32),
         "global_step": np.asarray(1234, np.int64),
         "beta1_power": np.asarray(0.5, np.float32),
         "model/inference/enc/kernel": rs.randn(5, 3, 4).astype(np.float32),
         "ints": np.arange(6, dtype=np.int32).reshape(2, 3)}
    prefix = str(tmp_path / "model.ckpt-1234")
    B.save_tf_checkpoint(prefix, t, entries_per_block=2)
    assert B.is_bundle(prefix) and os.path.exists(prefix + ".data-00000-of-00001")
    got = B.load_tf_checkpoint(prefix)
    assert set(got) == set(t)
    for k in t:
        assert got[k].dtype == t[k].dtype and got[k].shape == t[k].shape and np.array_equal(got[k], t[k]), k
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[10] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError):
        B.load_tf_checkpoint(prefix)
    B.load_tf_checkpoint(prefix, check_crc=False)                       # the flipped bit is then accepted


def test_name_mapping_onto_the_tacotron2_layout(tmp_path):
    from nspeech_amd import hparams as H
    from nspeech_amd.models import params as P
    hp = H.load("taco2")
    for k, v in dict(num_mels=8, num_freq=17, embedding_dim=8, encoder_conv_channels=8, encoder_lstm_units=4,
                     attention_dim=8, decoder_lstm_units=8, postnet_conv_channels=8, expand_conv_channels=8,
                     expand_lstm_units=4).items():
        setattr(hp, k, v)
    lay, st = P.taco2_layout(hp, 149)
    pv, sv = P.init_values(lay, st, 3)
    # a checkpoint spelt the way TF 1.x wrappers scope the decoder-loop variables, with optimizer slots beside them
    tf_names = {}
    for name in list(pv) + list(sv):
        tf_names[name] = B.name_candidates(name)[1] if len(B.name_candidates(name)) > 2 and name.startswith("decoder/") \
            else "model/inference/" + name
    assert tf_names["decoder/lstm_1/kernel"].endswith("multi_rnn_cell/cell_1/lstm_cell/kernel")
    tensors = {tf_names[k]: v for k, v in {**pv, **sv}.items()}
    tensors["global_step"] = np.asarray(77, np.int64)
    tensors["model/inference/embedding/embedding/Adam"] = np.zeros_like(pv["embedding/embedding"])
    tensors["beta2_power"] = np.asarray(0.9, np.float32)
    tensors["model/inference/something/else"] = np.zeros(3, np.float32)
    prefix = str(tmp_path / "model.ckpt-77")
    B.save_tf_checkpoint(prefix, tensors)
    params, stats, rep = B.map_checkpoint(B.load_tf_checkpoint(prefix), lay, st)
    assert rep["missing"] == [] and rep["global_step"] == 77 and rep["unused"] == ["model/inference/something/else"]
    assert all(np.array_equal(params[k], pv[k]) for k in pv) and all(np.array_equal(stats[k], sv[k]) for k in sv)
    # an explicit name map wins; a wrong shape is refused; a missing variable is reported
    alt = dict(tensors)
    alt["custom/emb"] = alt.pop("model/inference/embedding/embedding")
    p2, _, rep2 = B.map_checkpoint(alt, lay, st, name_map={"embedding/embedding": "custom/emb"})
    assert rep2["missing"] == [] and np.array_equal(p2["embedding/embedding"], pv["embedding/embedding"])
    _, _, rep3 = B.map_checkpoint(alt, lay, st)
    assert rep3["missing"] == ["embedding/embedding"]
    bad = dict(tensors)
    bad["model/inference/embedding/embedding"] = np.zeros((3, 3), np.float32)
    with pytest.raises(ValueError):
        B.map_checkpoint(bad, lay, st)
    # Adam slots (tf.train.Saver writes `<variable>/Adam`, `/Adam_1` beside every trainable, train.py:60): taken only as
    # a complete set, under this build's names; one slot above (embedding only) is not a set
    assert rep["adam_slots"] is None
    full = dict(tensors)
    rng = np.random.default_rng(5)
    for name in pv:
        full[tf_names[name] + "/Adam"] = rng.standard_normal(pv[name].shape).astype(np.float32)
        full[tf_names[name] + "/Adam_1"] = rng.random(pv[name].shape).astype(np.float32)
    full["beta1_power"] = np.asarray(0.9 ** 77, np.float32)
    prefix2 = str(tmp_path / "model.ckpt-78")
    B.save_tf_checkpoint(prefix2, full)
    _, _, rep4 = B.map_checkpoint(B.load_tf_checkpoint(prefix2), lay, st)
    m, v = rep4["adam_slots"]
    assert rep4["unused"] == ["model/inference/something/else"] and set(m) == set(pv) == set(v)
    assert all(np.array_equal(m[k], full[tf_names[k] + "/Adam"]) and np.array_equal(v[k], full[tf_names[k] + "/Adam_1"]) for k in pv)
    wrong = dict(full)
    wrong[tf_names["embedding/embedding"] + "/Adam_1"] = np.zeros((2, 2), np.float32)
    with pytest.raises(ValueError):
        B.map_checkpoint(wrong, lay, st)
    # the spelling the REFERENCE's graph gives its slots: the optimizer is built inside
    # variable_scope('model') / variable_scope('optimizer') (train.py:49, tacotron2.py:146) and TF 1.x nests a slot's
    # name under the open scope: model/optimizer/<variable op name>/Adam[_1], model/optimizer/beta{1,2}_power
    ref = dict(tensors)
    ref.pop("model/inference/embedding/embedding/Adam")
    ref.pop("beta2_power")
    for name in pv:
        ref["model/optimizer/" + tf_names[name] + "/Adam"] = full[tf_names[name] + "/Adam"]
        ref["model/optimizer/" + tf_names[name] + "/Adam_1"] = full[tf_names[name] + "/Adam_1"]
    ref["model/optimizer/beta1_power"] = np.asarray(0.9 ** 78, np.float32)
    ref["model/optimizer/beta2_power"] = np.asarray(0.999 ** 78, np.float32)
    prefix3 = str(tmp_path / "model.ckpt-79")
    B.save_tf_checkpoint(prefix3, ref)
    _, _, rep5 = B.map_checkpoint(B.load_tf_checkpoint(prefix3), lay, st)
    assert rep5["adam_slots"] is not None and rep5["adam_slots_reason"] is None, rep5["adam_slots_reason"]
    m5, v5 = rep5["adam_slots"]
    assert all(np.array_equal(m5[k], m[k]) and np.array_equal(v5[k], v[k]) for k in pv)
    assert rep5["unused"] == ["model/inference/something/else"]
    # an incomplete set says why it was not taken
    part = dict(ref)
    part.pop("model/optimizer/" + tf_names["dense/bias"] + "/Adam_1")
    _, _, rep6 = B.map_checkpoint(part, lay, st)
    assert rep6["adam_slots"] is None and rep6["adam_slots_present"] and "dense/bias" in rep6["adam_slots_reason"]
    assert rep["adam_slots_reason"] and rep["adam_slots_present"]


def test_export_writes_the_reference_optimizer_names(tmp_path):
    """export_model(with_adam_slots=True) -> model/optimizer/model/inference/<var>/Adam[_1] and beta powers =
    beta^(global_step + 1), what a TF Saver of the reference's graph would look up."""
    class _M(object):
        global_step = 4

        class _hparams(object):
            adam = {"beta1": 0.9, "beta2": 0.999}

        def numpy_params(self):
            return {"a/kernel": np.ones((2, 3), np.float32)}

        def numpy_stats(self):
            return {"a/moving_mean": np.zeros(3, np.float32)}

        def numpy_adam_slots(self):
            return {"a/kernel": np.full((2, 3), 2.0, np.float32)}, {"a/kernel": np.full((2, 3), 3.0, np.float32)}

    prefix = str(tmp_path / "model.ckpt-4")
    B.export_model(_M(), prefix, with_adam_slots=True)
    t = B.load_tf_checkpoint(prefix)
    assert set(t) == {"global_step", "model/inference/a/kernel", "model/inference/a/moving_mean",
                      "model/optimizer/model/inference/a/kernel/Adam", "model/optimizer/model/inference/a/kernel/Adam_1",
                      "model/optimizer/beta1_power", "model/optimizer/beta2_power"}
    assert abs(float(t["model/optimizer/beta1_power"]) - 0.9 ** 5) < 1e-7
    assert abs(float(t["model/optimizer/beta2_power"]) - 0.999 ** 5) < 1e-7
    assert float(t["model/optimizer/model/inference/a/kernel/Adam_1"][0, 0]) == 3.0


def test_reader_against_hand_assembled_bundle(tmp_path):
    """A bundle assembled byte by byte from the published LevelDB table format and tensor_bundle.proto by
    tests/golden/make_bundle_known_answer.py, which shares no code with nspeech_amd.utils.tf_bundle (its CRC-32C is the
    bitwise form, its varints / protobuf fields / block entries are written out literally): the reader must accept
    the footer magic, both block trailers and the prefix-compressed keys, and return the three tensors.  (Still not
    a file written by TensorFlow: none exists here - SURVEY 8c.)"""
    import json
    import os

    import numpy as np

    from nspeech_amd.utils import tf_bundle as B
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bundle_known_answer.json")))
    index, data = bytes.fromhex(gold["index_hex"]), bytes.fromhex(gold["data_hex"])
    assert index[-8:] == bytes.fromhex("57fb808b247547db")                    # kTableMagicNumber, little-endian
    n = gold["data_block_bytes"]
    trailer = index[n:n + 5]
    assert trailer.hex() == gold["data_block_trailer_hex"] and trailer[0] == 0              # uncompressed
    assert B.unmask_crc(int.from_bytes(trailer[1:], "little")) == B.crc32c(index[:n] + b"\x00")
    prefix = str(tmp_path / "model.ckpt-7")
    open(prefix + ".index", "wb").write(index)
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    assert B.is_bundle(prefix)
    got = B.load_tf_checkpoint(prefix, check_crc=True)
    assert sorted(got) == sorted(gold["tensors"])
    for name, t in gold["tensors"].items():
        assert got[name].dtype == np.dtype(t["dtype"]) and list(got[name].shape) == t["shape"]
        assert got[name].ravel().tolist() == t["values"]
    # a flipped payload bit is caught by the per-tensor checksum, a flipped index bit by the block trailer
    bad = bytearray(data)
    bad[13] ^= 0x10
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(bad))
    with __import__("pytest").raises(ValueError):
        B.load_tf_checkpoint(prefix, check_crc=True)
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    badi = bytearray(index)
    badi[20] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(badi))
    with __import__("pytest").raises(ValueError):
        B.load_tf_checkpoint(prefix)
    # and this build's writer produces the same table bytes for the same content
    tensors = {k: np.asarray(t["values"], dtype=t["dtype"]).reshape(t["shape"]) for k, t in gold["tensors"].items()}
    p2 = str(tmp_path / "again")
    B.save_tf_checkpoint(p2, tensors)
    assert open(p2 + ".data-00000-of-00001", "rb").read() == data
    mine = B.read_table(p2 + ".index")
    assert [k for k, _ in mine] == [b"", b"a/bias", b"a/kernel", b"b"]
    open(prefix + ".index", "wb").write(index)
    assert [v for _, v in mine][1:] == [v for _, v in B.read_table(prefix + ".index")][1:]     # identical entry protos
