# coding: utf-8
"""TensorFlow V2 checkpoint ("tensor bundle") reader and writer in plain Python/numpy
(SURVEY.md §8(f)-2): the on-disk format the reference's ``tf.train.Saver`` produces
(utils/saver.py:22-24,75; restore by variable name utils/saver.py:131-170), so published
``model-N.{index,data-00000-of-00001}`` checkpoints can be imported by reference variable name and
checkpoints written here can be restored by the reference.

Format (TensorFlow ``tensor_bundle`` + its LevelDB-derived ``table``; restated from the published
format, TensorFlow itself is not available in this image -- PARITY UNPINNED: no TF-written file
exists here to read back; covered by write->read round trips and structural checks):
  * ``<prefix>.index``: an immutable sorted string table.  Blocks of prefix-compressed entries
    ``varint shared | varint non_shared | varint value_len | key suffix | value`` followed by the
    uint32 restart offsets and their count; every block is trailed by a compression byte
    (0 = none, 1 = snappy) and a masked CRC32C of block+type.  The file ends with a 48-byte footer:
    metaindex handle, index handle (varint offset,size), zero padding, magic 0xdb4775248b80fb57.
    Key "" holds ``BundleHeaderProto`` (num_shards=1, little endian, version.producer=1); every
    other key is a tensor name holding ``BundleEntryProto`` {dtype=1, shape=2, shard_id=3,
    offset=4, size=5, crc32c=6 (masked CRC32C of the tensor bytes)}.
  * ``<prefix>.data-00000-of-00001``: the raw little-endian tensor bytes, in key order.
"""

import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8

# DataType enum values of tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"),
           6: np.dtype("i1"), 9: np.dtype("<i8"), 10: np.dtype("?"), 17: np.dtype("<u2"), 19: np.dtype("<f2"),
           22: np.dtype("<u4"), 23: np.dtype("<u8")}
_DT_BFLOAT16 = 14
_ENUM_OF = {v: k for k, v in _DTYPES.items()}


# ---- CRC32C (Castagnoli).  The C-ABI library's zk_crc32c does the bulk work when it is built;
# the table-driven fallback keeps this module usable without it (index blocks are tiny).
def _make_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_TABLE = _make_table()


def _crc32c_py(data, crc=0):
    crc ^= 0xffffffff
    tab = _TABLE
    for b in bytes(data):
        crc = tab[(crc ^ b) & 0xff] ^ (crc >> 8)
    return crc ^ 0xffffffff


def crc32c(data):
    """CRC32C of a bytes-like / contiguous numpy array."""
    arr = np.ascontiguousarray(data) if isinstance(data, np.ndarray) else None
    nbytes = arr.nbytes if arr is not None else len(data)
    if nbytes >= 4096:
        try:
            from zero_amd import hip
            import ctypes
            fn = hip.lib().raw("zk_crc32c")
            if arr is None:
                arr = np.frombuffer(bytes(data), dtype=np.uint8)
            return int(fn(ctypes.c_void_p(arr.ctypes.data), ctypes.c_size_t(nbytes), ctypes.c_uint32(0)))
        except Exception:          # library not built: plain-Python path below
            pass
    return _crc32c_py(arr.tobytes() if arr is not None else data)


def mask_crc(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + _MASK_DELTA) & 0xffffffff


def unmask_crc(masked):
    rot = (masked - _MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- protobuf wire helpers (only what the two bundle messages need)
def _put_varint(out, v):
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)


def _get_varint(buf, pos):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7f) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _fields(buf):
    """(field number, wire type, value) triples of one message."""
    pos, end = 0, len(buf)
    while pos < end:
        tag, pos = _get_varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _get_varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            val, pos = buf[pos:pos + n], pos + n
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, val


def _encode_entry(dtype_enum, shape, offset, size, masked_crc):
    out = bytearray()
    out += b"\x08"
    _put_varint(out, dtype_enum)
    shp = bytearray()
    for d in shape:
        dim = bytearray(b"\x08")
        _put_varint(dim, int(d))
        shp += b"\x12"
        _put_varint(shp, len(dim))
        shp += dim
    out += b"\x12"
    _put_varint(out, len(shp))
    out += shp
    if offset:
        out += b"\x20"
        _put_varint(out, offset)
    if size:
        out += b"\x28"
        _put_varint(out, size)
    out += b"\x35" + struct.pack("<I", masked_crc)
    return bytes(out)


def _decode_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
    for num, wt, val in _fields(buf):
        if num == 1:
            e["dtype"] = val
        elif num == 2:
            for n2, _, dimbuf in _fields(val):
                if n2 == 2:
                    size = 0
                    for n3, _, v3 in _fields(dimbuf):
                        if n3 == 1:
                            size = v3 - (1 << 64) if v3 >> 63 else v3
                    e["shape"].append(size)
        elif num == 3:
            e["shard_id"] = val
        elif num == 4:
            e["offset"] = val
        elif num == 5:
            e["size"] = val
        elif num == 6:
            e["crc32c"] = struct.unpack("<I", val)[0]
        elif num == 7:
            e["sliced"] = True
    return e


# ---- table (sorted string table) --------------------------------------------------------
def _build_block(items, restart_interval=16):
    out, restarts, last, count = bytearray(), [], b"", 0
    for key, value in items:
        if count % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            for a, b in zip(last, key):
                if a != b:
                    break
                shared += 1
        _put_varint(out, shared)
        _put_varint(out, len(key) - shared)
        _put_varint(out, len(value))
        out += key[shared:] + value
        last, count = key, count + 1
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _parse_block(buf):
    nrestart = struct.unpack("<I", buf[-4:])[0]
    end = len(buf) - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(buf, pos)
        non_shared, pos = _get_varint(buf, pos)
        vlen, pos = _get_varint(buf, pos)
        key = key[:shared] + bytes(buf[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(buf[pos:pos + vlen])
        pos += vlen


def _handle(offset, size):
    out = bytearray()
    _put_varint(out, offset)
    _put_varint(out, size)
    return bytes(out)


def _write_table(path, items, block_bytes=4096):
    """items: sorted [(key bytes, value bytes)]."""
    with open(path, "wb") as f:
        pos = 0

        def emit(block):
            nonlocal pos
            trailer = b"\x00"
            f.write(block + trailer + struct.pack("<I", mask_crc(_crc32c_py(block + trailer))))
            h = (pos, len(block))
            pos += len(block) + 5
            return h

        index, pending, pending_bytes = [], [], 0
        for key, value in items:
            pending.append((key, value))
            pending_bytes += len(key) + len(value) + 3
            if pending_bytes >= block_bytes:
                index.append((pending[-1][0], _handle(*emit(_build_block(pending)))))
                pending, pending_bytes = [], 0
        if pending:
            index.append((pending[-1][0], _handle(*emit(_build_block(pending)))))
        meta = emit(_build_block([]))
        idx = emit(_build_block(index, restart_interval=1))
        footer = _handle(*meta) + _handle(*idx)
        footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
        f.write(footer)


def _read_block(raw, offset, size, verify=True):
    block, ctype = raw[offset:offset + size], raw[offset + size]
    if verify:
        stored = struct.unpack("<I", raw[offset + size + 1:offset + size + 5])[0]
        if unmask_crc(stored) != _crc32c_py(raw[offset:offset + size + 1]):
            raise ValueError("checkpoint index: block checksum mismatch at offset %d" % offset)
    if ctype != 0:
        raise ValueError("checkpoint index block is compressed (type %d); only uncompressed tables are "
                         "supported (tf.train.Saver writes them uncompressed)" % ctype)
    return block


def _read_table(path):
    raw = open(path, "rb").read()
    if len(raw) < 48 or struct.unpack("<Q", raw[-8:])[0] != _MAGIC:
        raise ValueError("%s is not a tensor-bundle index (bad magic)" % path)
    footer = raw[-48:]
    _, p = _get_varint(footer, 0)
    _, p = _get_varint(footer, p)
    ioff, p = _get_varint(footer, p)
    isize, p = _get_varint(footer, p)
    for _, hv in _parse_block(_read_block(raw, ioff, isize)):
        off, q = _get_varint(hv, 0)
        size, _ = _get_varint(hv, q)
        for key, value in _parse_block(_read_block(raw, off, size)):
            yield key, value


# ---- public API ------------------------------------------------------------------------
def _bf16_to_f32(raw):
    return (raw.astype(np.uint32) << 16).view(np.float32)


def list_variables(prefix):
    """[(name, shape, dtype enum)] of a checkpoint."""
    out = []
    for key, value in _read_table(prefix + ".index"):
        if key == b"":
            continue
        e = _decode_entry(value)
        out.append((key.decode("utf-8"), tuple(e["shape"]), e["dtype"]))
    return out


def load_checkpoint(prefix, names=None, verify=False):
    """{variable name: numpy array} of ``<prefix>.index`` + data shards.  ``names``: restrict to
    these variables.  bfloat16 tensors are widened to float32."""
    entries, num_shards = {}, 1
    for key, value in _read_table(prefix + ".index"):
        if key == b"":
            for num, _, val in _fields(value):
                if num == 1:
                    num_shards = val
                elif num == 2 and val != 0:
                    raise ValueError("big-endian checkpoints are not supported")
            continue
        name = key.decode("utf-8")
        if names is None or name in names:
            entries[name] = _decode_entry(value)
    shards, out = {}, {}
    for name, e in entries.items():
        if e["sliced"]:
            raise ValueError("%s is a partitioned variable; not supported" % name)
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), dtype=np.uint8, mode="r")
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        if verify and e["crc32c"] is not None and unmask_crc(e["crc32c"]) != crc32c(np.asarray(raw)):
            raise ValueError("checkpoint tensor %s: checksum mismatch" % name)
        if e["dtype"] == _DT_BFLOAT16:
            arr = _bf16_to_f32(np.frombuffer(raw.tobytes(), dtype="<u2"))
        elif e["dtype"] in _DTYPES:
            arr = np.frombuffer(raw.tobytes(), dtype=_DTYPES[e["dtype"]])
        else:
            raise ValueError("checkpoint tensor %s has unsupported dtype enum %d" % (name, e["dtype"]))
        out[name] = arr.reshape(e["shape"]).copy()
    return out


def save_checkpoint(prefix, tensors):
    """Write {name: numpy array} as a single-shard V2 checkpoint at ``prefix``."""
    items = [(b"", b"\x08\x01\x1a\x02\x08\x01")]        # num_shards=1, version{producer=1}
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as data:
        for name in sorted(tensors, key=lambda n: n.encode("utf-8")):
            arr = np.require(tensors[name], requirements="C")      # keeps 0-d scalars 0-d
            if arr.dtype.newbyteorder("<") not in _ENUM_OF and arr.dtype not in _ENUM_OF:
                raise ValueError("unsupported dtype %s for %s" % (arr.dtype, name))
            arr = arr.astype(arr.dtype.newbyteorder("<"), copy=False)
            enum = _ENUM_OF[np.dtype(arr.dtype)]
            data.write(arr.tobytes())
            items.append((name.encode("utf-8"),
                          _encode_entry(enum, arr.shape, offset, arr.nbytes, mask_crc(crc32c(arr)))))
            offset += arr.nbytes
    _write_table(prefix + ".index", items)
