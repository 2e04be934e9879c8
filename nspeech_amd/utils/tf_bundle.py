"""TensorFlow checkpoint (TensorBundle, "V2") container: reader, writer and the name mapping onto this build's
parameter layout (the reference saves and restores through tf.train.Saver, train.py:60,67-71,96-97 and
synthesizer.py:37-38).

UNVERIFIED AGAINST REAL FILES: TensorFlow is not installable here and the reference ships no checkpoint, so the
container is restated from the published format and checked by round trips and known answers only:
  * `<prefix>.index` is a LevelDB-format sorted table (tensorflow/core/lib/io/table*): data blocks of
    prefix-compressed (key, value) entries with a restart array, an index block, an (empty) metaindex block and a
    48-byte footer ending in the magic 0xdb4775248b80fb57; every block is followed by a 1-byte compression type
    (0 = none; snappy blocks are refused) and a masked CRC-32C.
  * key "" holds a BundleHeaderProto (num_shards, endianness, version); every other key is a variable name whose
    value is a BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (masked, fixed32)}.
  * `<prefix>.data-SSSSS-of-NNNNN` holds the raw little-endian tensor bytes.
The variable names inside tf.contrib.seq2seq.dynamic_decode are TensorFlow-internal ([3P]); `name_candidates`
lists the spellings tried for each variable of this build and `load_into_model` accepts an explicit name map, reports
what it could not match and refuses silently partial loads."""
import os
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
_DT = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 19: np.float16, 14: "bfloat16"}
_DT_INV = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9,
           np.dtype(np.float16): 19}


# ------------------------------------------------------------------ CRC-32C (Castagnoli), masked as LevelDB does
def _crc_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_T = _crc_table()


def crc32c(data, crc=0):
    """Byte-at-a-time table CRC (about 10 MB/s in CPython: a 140 MB Tacotron-2 bundle takes ~15 s to verify)."""
    c = (crc ^ 0xFFFFFFFF) & 0xFFFFFFFF
    tab = _T
    for b in bytes(data):
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def unmask_crc(m):
    r = (m - 0xa282ead8) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------ varints / protobuf wire format
def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf, pos):
    shift = v = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _pb_fields(buf):
    """Yields (field number, wire type, value) of one protobuf message; length-delimited values as bytes."""
    pos = 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _pb_varint(field, v):
    return _put_varint(field << 3) + _put_varint(v)


def _pb_bytes(field, b):
    return _put_varint((field << 3) | 2) + _put_varint(len(b)) + b


def _entry_proto(dtype_enum, shape, shard, offset, size, crc_masked):
    dims = b"".join(_pb_bytes(2, _pb_varint(1, d)) for d in shape)          # TensorShapeProto.dim = 2, Dim.size = 1
    msg = _pb_varint(1, dtype_enum) + _pb_bytes(2, dims)
    if shard:
        msg += _pb_varint(3, shard)
    if offset:
        msg += _pb_varint(4, offset)
    msg += _pb_varint(5, size) + _put_varint((6 << 3) | 5) + struct.pack("<I", crc_masked)
    return msg


def _parse_entry(buf):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for f, wt, v in _pb_fields(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            for f2, _, v2 in _pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1:
                            size = v3
                    e["shape"].append(size)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = v
        elif f == 7:
            e["sliced"] = True
    return e


# ------------------------------------------------------------------ sorted table (LevelDB format)
def _read_block(buf, offset, size):
    raw = buf[offset:offset + size]
    ctype = buf[offset + size]
    stored = struct.unpack_from("<I", buf, offset + size + 1)[0]
    if unmask_crc(stored) != crc32c(bytes(raw) + bytes([ctype])):
        raise ValueError("table block at %d: checksum mismatch" % offset)
    if ctype != 0:
        raise ValueError("table block at %d is compressed (type %d): only uncompressed bundles are supported" % (offset, ctype))
    n_restarts = struct.unpack_from("<I", raw, len(raw) - 4)[0]
    end = len(raw) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _get_varint(raw, pos)
        non_shared, pos = _get_varint(raw, pos)
        vlen, pos = _get_varint(raw, pos)
        key = key[:shared] + bytes(raw[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(raw[pos:pos + vlen])))
        pos += vlen
    return out


def _build_block(items, restart_interval=16):
    body, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        if i % restart_interval == 0:
            restarts.append(len(body))
            shared = 0
        else:
            shared = 0
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        body += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    body += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    return bytes(body)


def _append_block(out, block):
    handle = (len(out), len(block))
    out += block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00")))
    return handle


def _handle(h):
    return _put_varint(h[0]) + _put_varint(h[1])


def read_table(path):
    buf = memoryview(open(path, "rb").read())
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError("%s is not a TensorBundle index (bad magic)" % path)
    foot = buf[len(buf) - 48:]
    pos = 0
    _, pos = _get_varint(foot, pos)          # metaindex handle
    _, pos = _get_varint(foot, pos)
    ioff, pos = _get_varint(foot, pos)
    isize, pos = _get_varint(foot, pos)
    items = []
    for _, hv in _read_block(buf, ioff, isize):
        off, p = _get_varint(hv, 0)
        size, _ = _get_varint(hv, p)
        items.extend(_read_block(buf, off, size))
    return items


def write_table(path, items, entries_per_block=64):
    items = sorted(items)
    out, index = bytearray(), []
    for i in range(0, len(items), entries_per_block):
        chunk = items[i:i + entries_per_block]
        h = _append_block(out, _build_block(chunk))
        index.append((chunk[-1][0], _handle(h)))          # separator: the block's last key
    meta = _append_block(out, _build_block([]))
    idx = _append_block(out, _build_block(index, restart_interval=1))
    foot = _handle(meta) + _handle(idx)
    out += foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", MAGIC)
    with open(path, "wb") as f:
        f.write(bytes(out))


# ------------------------------------------------------------------ bundles
def _bf16_to_f32(raw):
    u = np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16
    return u.view(np.float32)


def load_tf_checkpoint(prefix, check_crc=True):
    """name -> numpy array for every variable of the bundle `<prefix>.index` + data shards."""
    items = read_table(prefix + ".index")
    if not items or items[0][0] != b"":
        raise ValueError("bundle header missing")
    num_shards, endian = 1, 0
    for f, _, v in _pb_fields(items[0][1]):
        if f == 1:
            num_shards = v
        elif f == 2:
            endian = v
    if endian != 0:
        raise ValueError("big-endian bundles are not supported")
    shards = {}
    out = {}
    for key, val in items[1:]:
        e = _parse_entry(val)
        if e["sliced"]:
            raise ValueError("%s: partitioned (sliced) variables are not supported" % key.decode())
        if e["dtype"] not in _DT:
            raise ValueError("%s: unsupported dtype enum %d" % (key.decode(), e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), dtype=np.uint8, mode="r")
        raw = bytes(shards[sid][e["offset"]:e["offset"] + e["size"]])
        if len(raw) != e["size"]:
            raise ValueError("%s: data shard is truncated" % key.decode())
        if check_crc and e["crc32c"] is not None and unmask_crc(e["crc32c"]) != crc32c(raw):
            raise ValueError("%s: tensor checksum mismatch" % key.decode())
        dt = _DT[e["dtype"]]
        arr = _bf16_to_f32(raw) if dt == "bfloat16" else np.frombuffer(raw, dtype=np.dtype(dt).newbyteorder("<"))
        out[key.decode()] = arr.reshape(e["shape"]).copy()
    return out


def save_tf_checkpoint(prefix, tensors, entries_per_block=64):
    """Writes `<prefix>.index` and `<prefix>.data-00000-of-00001` (one shard, uncompressed, little-endian)."""
    header = _pb_varint(1, 1) + _pb_bytes(3, _pb_varint(1, 1))          # num_shards = 1, version.producer = 1
    items = [(b"", header)]
    off = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in sorted(tensors):
            a = np.asarray(tensors[name])            # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in _DT_INV:
                a = a.astype(np.float32)
            raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
            f.write(raw)
            items.append((name.encode(), _entry_proto(_DT_INV[a.dtype], a.shape, 0, off, len(raw), mask_crc(crc32c(raw)))))
            off += len(raw)
    write_table(prefix + ".index", items, entries_per_block)


# ------------------------------------------------------------------ names
OPTIMIZER_SCOPE = "model/optimizer/"      # train.py:49 variable_scope('model') + tacotron2.py:146 variable_scope('optimizer')


def name_candidates(name):
    """Spellings under which a TF1 graph of the reference may have saved this build's variable `name` (relative to
    'model/inference/').  The first is this build's own convention; the others follow the scopes TF 1.x wrappers add
    inside dynamic_decode (multi_rnn_cell/cell_N, output_projection_wrapper, attention_wrapper, lstm_cell ->
    lstm_block_cell) - unverifiable here, hence the explicit name map in load_into_model."""
    base = "model/inference/"
    c = [base + name]
    dec = "decoder/output_projection_wrapper/multi_rnn_cell/"
    table = {
        "decoder/attention_lstm/": dec + "cell_0/attention_wrapper/lstm_cell/",
        "decoder/decoder_prenet/": dec + "cell_0/attention_wrapper/decoder_prenet/",
        "decoder/dense/": dec + "cell_0/attention_wrapper/dense/",
        "decoder/attention/query_layer/": dec + "cell_0/attention_wrapper/location_sensitive_attention/query_layer/",
        "decoder/attention/location_conv/": dec + "cell_0/attention_wrapper/location_sensitive_attention/location_conv/",
        "decoder/attention/location_layer/": dec + "cell_0/attention_wrapper/location_sensitive_attention/location_layer/",
        "decoder/attention/attention_v": dec + "cell_0/attention_wrapper/location_sensitive_attention/attention_v",
        "decoder/lstm_1/": dec + "cell_1/lstm_cell/",
        "decoder/lstm_2/": dec + "cell_2/lstm_cell/",
        "decoder/output_projection/": "decoder/output_projection_wrapper/",
    }
    for k, v in table.items():
        if name.startswith(k):
            c.append(base + v + name[len(k):])
            c.append(base + (v + name[len(k):]).replace("lstm_cell/", "lstm_block_cell/"))
    c.append(base + name.replace("lstm_cell/", "lstm_block_cell/"))
    seen, out = set(), []
    for x in c:
        if x not in seen:
            seen.add(x)
            out.append(x)
    return out


def map_checkpoint(tensors, layout, stat_layout, name_map=None):
    """-> (params dict, stats dict, report).  `name_map`: this build's name -> checkpoint name overrides."""
    name_map = name_map or {}
    found, missing, used, hits = {}, [], set(), {}
    for lay in (layout, stat_layout):
        for name, (_, shape) in lay.entries.items():
            cands = [name_map[name]] if name in name_map else name_candidates(name)
            hit = next((c for c in cands if c in tensors), None)
            if hit is None:
                missing.append(name)
                continue
            a = np.asarray(tensors[hit])
            if tuple(a.shape) != tuple(shape):
                raise ValueError("%s: checkpoint tensor %s has shape %s, the model needs %s" % (name, hit, a.shape, shape))
            found[name] = a.astype(np.float32)
            used.add(hit)
            hits[name] = hit
    extra = sorted(k for k in tensors if k not in used and not k.endswith("/Adam") and not k.endswith("/Adam_1")
                   and not k.endswith("_power") and k != "global_step")
    report = dict(missing=missing, unused=extra, global_step=int(tensors["global_step"]) if "global_step" in tensors else None)
    params = {k: found[k] for k in layout.entries if k in found}
    stats = {k: found[k] for k in stat_layout.entries if k in found}
    # tf.train.AdamOptimizer keeps its moments as slot variables named after each trainable, `<variable>/Adam` (m) and
    # `<variable>/Adam_1` (v) (the reference saves them with the model, train.py:60 - tf.train.Saver over all variables).
    # TF 1.x creates a slot inside `variable_scope(None, primary.op.name + "/" + slot_name)`, i.e. NESTED under whatever
    # variable scope is open when the optimizer builds its slots: the reference opens 'model' (train.py:49) and then
    # 'optimizer' (tacotron2.py:146), so its checkpoints should hold `model/optimizer/model/inference/<var>/Adam[_1]`
    # and `model/optimizer/beta{1,2}_power` [3P: TensorFlow is not installed here, unverified].  Both spellings are
    # taken: the bare `<variable>/Adam`, else any single tensor whose name ENDS with `/<variable>/Adam`.
    # The bias-correction powers are functions of the update count, which global_step carries.
    # The slots are taken only as a complete set: report["adam_slots"] = (m, v) dicts by this build's names, or None with
    # report["adam_slots_reason"] saying why.
    def slot(hit, suffix):
        if hit + suffix in tensors:
            return hit + suffix
        tail = "/" + hit + suffix
        c = [k for k in tensors if k.endswith(tail)]
        return c[0] if len(c) == 1 else None

    m, v = {}, {}
    reason = None
    has_any = any(k.endswith("/Adam") or k.endswith("/Adam_1") for k in tensors)
    for name in layout.entries:
        hit = hits.get(name)
        km, kv = (slot(hit, "/Adam"), slot(hit, "/Adam_1")) if hit is not None else (None, None)
        if km is None or kv is None:
            reason = ("no Adam slot tensors in the checkpoint" if not has_any else
                      "no (unique) Adam slots for %s (looked for %s/Adam[_1] bare and under any scope prefix)" % (name, hit))
            break
        am, av = np.asarray(tensors[km]), np.asarray(tensors[kv])
        if tuple(am.shape) != tuple(params[name].shape) or tuple(av.shape) != tuple(params[name].shape):
            raise ValueError("%s: Adam slots of %s have shapes %s / %s, the variable %s" % (name, hit, am.shape, av.shape, params[name].shape))
        m[name], v[name] = am.astype(np.float32), av.astype(np.float32)
    if reason is None and missing:
        reason = "variables missing from the checkpoint"
    report["adam_slots"] = (m, v) if reason is None and len(m) == len(layout.entries) else None
    report["adam_slots_reason"] = reason
    report["adam_slots_present"] = has_any
    return params, stats, report


def load_into_model(model, prefix, name_map=None):
    """Loads a TF bundle into a Tacotron model of this build; raises if any variable could not be matched."""
    params, stats, report = map_checkpoint(load_tf_checkpoint(prefix), model.layout, model.stat_layout, name_map)
    if report["missing"]:
        raise KeyError("TF checkpoint %s lacks %d variables (first: %s); pass a name map. Unused checkpoint tensors: %s"
                       % (prefix, len(report["missing"]), report["missing"][:3], report["unused"][:5]))
    model.load_numpy(params, stats)
    if report["global_step"] is not None:
        model.global_step = report["global_step"]
    if report.get("adam_slots") is not None and hasattr(model, "load_adam_slots"):
        model.load_adam_slots(*report["adam_slots"])       # resume with the reference's optimizer state (train.py:67-71)
    elif report.get("adam_slots_present"):
        import sys
        sys.stderr.write("[tf_bundle] %s holds Adam slot tensors that were NOT taken (%s): the moments start from zero\n"
                         % (prefix, report.get("adam_slots_reason")))
    return report


def export_model(model, prefix, with_adam_slots=False):
    """Writes the model's trainables, BatchNorm moving statistics and global_step as a TF bundle under this build's
    names ('model/inference/...'); with_adam_slots: also the optimizer state the way the reference's graph names it
    (see map_checkpoint): the moments as `model/optimizer/model/inference/<variable>/Adam`, `/Adam_1` and
    `model/optimizer/beta{1,2}_power` = beta^(updates + 1) - TF initialises the powers to beta and multiplies once per
    update (adam.py: _create_slots / _finish) [3P, unverified here]."""
    t = {"model/inference/" + k: v for k, v in model.numpy_params().items()}
    t.update({"model/inference/" + k: v for k, v in model.numpy_stats().items()})
    t["global_step"] = np.asarray(model.global_step, np.int64)
    if with_adam_slots and hasattr(model, "numpy_adam_slots"):
        m, v = model.numpy_adam_slots()
        t.update({OPTIMIZER_SCOPE + "model/inference/" + k + "/Adam": a for k, a in m.items()})
        t.update({OPTIMIZER_SCOPE + "model/inference/" + k + "/Adam_1": a for k, a in v.items()})
        hp = getattr(model, "_hparams", None)
        adam = getattr(hp, "adam", None) or {}
        n = int(model.global_step) + 1
        t[OPTIMIZER_SCOPE + "beta1_power"] = np.asarray(float(adam.get("beta1", 0.9)) ** n, np.float32)
        t[OPTIMIZER_SCOPE + "beta2_power"] = np.asarray(float(adam.get("beta2", 0.999)) ** n, np.float32)
    save_tf_checkpoint(prefix, t)


def is_bundle(path):
    return os.path.isfile(path + ".index")
