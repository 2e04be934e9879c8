#!/usr/bin/env python3
"""Hand-assembled TensorBundle ("V2" checkpoint) of three tensors, written WITHOUT nspeech_amd.utils.tf_bundle: every
byte is laid down here from the published formats, so that the reader is checked against something other than its own
writer (TensorFlow itself cannot run in this container).

  LevelDB table format (leveldb/doc/table_format.md, used by tensorflow/core/lib/io/table*):
    block   := entry* restart_offset(uint32 LE)* num_restarts(uint32 LE)
    entry   := varint(shared key bytes) varint(unshared key bytes) varint(value bytes) key_delta value
    on disk := block | compression type (1 byte, 0 = none) | masked CRC-32C of (block | type) (uint32 LE)
    mask(c) := rotate_right(c, 15) + 0xa282ead8                                     (leveldb/util/crc32c.h)
    file    := data blocks | metaindex block | index block | footer
    footer  := metaindex handle | index handle | zero padding to 40 bytes | magic 0xdb4775248b80fb57 (uint64 LE)
    handle  := varint64 offset, varint64 size (size without the 5-byte trailer)
    index block entry: key >= last key of the data block (here its short successor), value = the block's handle
  tensorflow/core/protobuf/tensor_bundle.proto:
    key ""   -> BundleHeaderProto { num_shards = 1 (field 1); endianness = LITTLE (field 2, default); version = 3 }
    key name -> BundleEntryProto  { dtype = 1; shape = 2 (TensorShapeProto.dim = 2 { size = 1 }); shard_id = 3;
                                    offset = 4; size = 5; crc32c = 6 (fixed32, masked CRC-32C of the tensor bytes) }
    <prefix>.data-00000-of-00001 holds the raw little-endian tensor bytes back to back.

Run: python tests/golden/make_bundle_known_answer.py   (rewrites bundle_known_answer.json next to it)."""
import json
import os
import struct


def crc32c_bitwise(data):
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) one bit at a time - deliberately not the table form."""
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 & -(c & 1))
    return c ^ 0xFFFFFFFF


assert crc32c_bitwise(b"123456789") == 0xE3069283          # the standard check value
assert crc32c_bitwise(bytes(32)) == 0x8A9136AA               # RFC 3720 B.4: 32 bytes of zeros


def masked(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def with_trailer(block):
    return block + b"\x00" + struct.pack("<I", masked(crc32c_bitwise(block + b"\x00")))


# ---- the tensors and the data shard
a_bias = struct.pack("<3f", 0.5, -1.25, 3.0)                                  # a/bias   float32 [3]
a_kernel = struct.pack("<6f", 1.0, 2.0, 3.0, 4.0, 5.0, 6.0)                  # a/kernel float32 [2, 3]
b_steps = struct.pack("<4i", 7, -1, 0, 2 ** 31 - 1)                            # b        int32   [4]
data = a_bias + a_kernel + b_steps


def shape_proto(dims):
    body = b"".join(b"\x12" + varint(2) + b"\x08" + varint(d) for d in dims)    # dim (field 2) { size (field 1) }
    return b"\x12" + varint(len(body)) + body                                   # shape = field 2 of the entry


def entry_proto(dtype, dims, offset, raw):
    out = b"\x08" + varint(dtype) + shape_proto(dims)
    if offset:
        out += b"\x20" + varint(offset)                                         # offset (field 4); 0 is the default
    out += b"\x28" + varint(len(raw))                                           # size (field 5)
    out += b"\x35" + struct.pack("<I", masked(crc32c_bitwise(raw)))             # crc32c (field 6, fixed32)
    return out


header = b"\x08\x01" + b"\x1a\x02\x08\x01"            # num_shards = 1; version { producer = 1 }
e_bias = entry_proto(1, [3], 0, a_bias)               # DT_FLOAT = 1
e_kernel = entry_proto(1, [2, 3], 12, a_kernel)
e_steps = entry_proto(3, [4], 36, b_steps)            # DT_INT32 = 3

# ---- the one data block: keys in order "", "a/bias", "a/kernel" (shares "a/" with its predecessor), "b"
blk = bytearray()
blk += varint(0) + varint(0) + varint(len(header)) + b"" + header
blk += varint(0) + varint(6) + varint(len(e_bias)) + b"a/bias" + e_bias
blk += varint(2) + varint(6) + varint(len(e_kernel)) + b"kernel" + e_kernel
blk += varint(0) + varint(1) + varint(len(e_steps)) + b"b" + e_steps
blk += struct.pack("<I", 0) + struct.pack("<I", 1)    # one restart point at offset 0
data_block = bytes(blk)
meta_block = struct.pack("<I", 0) + struct.pack("<I", 1)                      # empty metaindex
index = bytearray()
index_value = varint(0) + varint(len(data_block))
index += varint(0) + varint(1) + varint(len(index_value)) + b"c" + index_value      # "c" = short successor of "b"
index += struct.pack("<I", 0) + struct.pack("<I", 1)
index_block = bytes(index)

out = bytearray()
out += with_trailer(data_block)
meta_handle = varint(len(out)) + varint(len(meta_block))
out += with_trailer(meta_block)
index_handle = varint(len(out)) + varint(len(index_block))
out += with_trailer(index_block)
footer = meta_handle + index_handle
footer += bytes(40 - len(footer)) + bytes.fromhex("57fb808b247547db")
out += footer

here = os.path.dirname(os.path.abspath(__file__))
json.dump({"index_hex": bytes(out).hex(), "data_hex": data.hex(),
           "data_block_bytes": len(data_block),
           "data_block_trailer_hex": bytes(out[len(data_block):len(data_block) + 5]).hex(),
           "tensors": {"a/bias": {"dtype": "float32", "shape": [3], "values": [0.5, -1.25, 3.0]},
                       "a/kernel": {"dtype": "float32", "shape": [2, 3], "values": [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]},
                       "b": {"dtype": "int32", "shape": [4], "values": [7, -1, 0, 2 ** 31 - 1]}}},
          open(os.path.join(here, "bundle_known_answer.json"), "w"), indent=1)
print("wrote bundle_known_answer.json: index %d bytes, data %d bytes" % (len(out), len(data)))
