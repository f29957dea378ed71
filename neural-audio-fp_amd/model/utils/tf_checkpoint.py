"""Read a checkpoint written by the REFERENCE (tf.train.Checkpoint(optimizer=..., model=m_fp), saved by
tf.train.CheckpointManager: model/utils/experiment_helper.py:100-136; restored at model/generate.py:26-52) straight from
its files, without TensorFlow:

    <LOG_ROOT_DIR>/checkpoint/<NAME>/ckpt-<N>.index                  TensorBundle index: an SSTable (LevelDB table format)
    <LOG_ROOT_DIR>/checkpoint/<NAME>/ckpt-<N>.data-00000-of-00001    the tensor bytes

UNTESTED AGAINST A TF-WRITTEN FILE: TensorFlow is absent from the build image and the reference ships no checkpoint.
What is restated here is the PUBLISHED on-disk format (tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc},
tensorflow/core/lib/io/{format,block,table}.cc, tensorflow/core/protobuf/tensor_bundle.proto), and every safeguard the
format offers is enforced so that a misreading cannot pass silently:
  * the table footer's magic number, every block's masked CRC-32C (index blocks) and every tensor's masked CRC-32C
    (BundleEntryProto.crc32c) are verified -- CRC-32C (Castagnoli) itself is pinned to the RFC 3720 test vectors;
  * dtype must be DT_FLOAT, shapes must equal the encoder's tensor shapes, all 576 keras variables of the model must be
    found exactly once, and the parameter total must be the model's (16,939,008 for the 1-s input).
`tests/test_tf_checkpoint.py` round-trips a bundle written by a writer restated from the same sources (test code).

Object-graph keys (tf.train.Checkpoint names a variable by the shortest attribute path from the root):
    model/front_conv/layer_with_weights-<b>/{conv2d_1x3,conv2d_3x1}/{kernel,bias}/.ATTRIBUTES/VARIABLE_VALUE
    model/front_conv/layer_with_weights-<b>/{BN_1x3,BN_3x1}/{gamma,beta}/.ATTRIBUTES/VARIABLE_VALUE
    model/div_enc/split_fc_layers/<q>/layer_with_weights-{0,1}/{kernel,bias}/.ATTRIBUTES/VARIABLE_VALUE
(attribute names of model/fp/nnfp.py:48-79, 120-139, 210-222); the equivalent path through ConvLayer.forward
(`forward/layer_with_weights-{0..3}`) is accepted too.  `optimizer/...` and bookkeeping keys are ignored.
"""
import ctypes
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
MASK_DELTA = 0xa282ead8
DT_FLOAT = 1
SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'


# ---- CRC-32C -----------------------------------------------------------------------------------------------------
def crc32c(data, crc=0):
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) through the library's host routine."""
    from ... import _lib
    lib = _lib.load()
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    return int(lib.nafp_crc32c_host(buf.ctypes.data_as(ctypes.c_void_p), buf.size, ctypes.c_uint32(crc)))


def mask_crc(crc):
    """crc32c::Mask (tensorflow/core/lib/hash/crc32c.h): rotate right by 15 bits, add a constant."""
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + MASK_DELTA) & 0xffffffff


# ---- varints / minimal protobuf -------------------------------------------------------------------------------------
def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]; pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise ValueError('varint too long')


def _fields(buf):
    """(field number, wire type, value) of one protobuf message; value = int (varint / fixed) or bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]; pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + ln]); pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]; pos += 4
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        yield fno, wt, v


def parse_bundle_entry(buf):
    """BundleEntryProto (tensor_bundle.proto): dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5,
    crc32c = 6 (fixed32), slices = 7."""
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': 0, 'slices': 0}
    for fno, wt, v in _fields(buf):
        if fno == 1:
            e['dtype'] = v
        elif fno == 2:                                   # TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }
            for f2, _, v2 in _fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            size = v3
                    e['shape'].append(size)
        elif fno == 3:
            e['shard_id'] = v
        elif fno == 4:
            e['offset'] = v
        elif fno == 5:
            e['size'] = v
        elif fno == 6:
            e['crc32c'] = v
        elif fno == 7:
            e['slices'] += 1
    return e


def parse_bundle_header(buf):
    """BundleHeaderProto: num_shards = 1, endianness = 2 (0 = little), version = 3."""
    h = {'num_shards': 1, 'endianness': 0}
    for fno, _, v in _fields(buf):
        if fno == 1:
            h['num_shards'] = v
        elif fno == 2:
            h['endianness'] = v
    return h


# ---- SSTable (LevelDB table format as tensorflow/core/lib/io writes it) ----------------------------------------------
def _read_block(data, offset, size):
    """Block contents at a BlockHandle; the 5-byte trailer (compression type, masked crc32c of contents + type) is verified."""
    raw = data[offset:offset + size + 5]
    if len(raw) != size + 5:
        raise ValueError('index file truncated')
    ctype = raw[size]
    want = struct.unpack_from('<I', raw, size + 1)[0]
    if mask_crc(crc32c(raw[:size + 1])) != want:
        raise ValueError(f'block at {offset}: CRC-32C mismatch')
    if ctype != 0:
        raise NotImplementedError('compressed (snappy) table block; BundleWriter writes uncompressed blocks')
    return raw[:size]


def _block_entries(block):
    """(key, value) pairs of one block: prefix-compressed entries followed by the restart array."""
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b''
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path):
    """{key: value} of an SSTable file."""
    data = open(path, 'rb').read()
    if len(data) < 48:
        raise ValueError(f'{path}: too short for a table footer')
    footer = data[-48:]
    if struct.unpack_from('<Q', footer, 40)[0] != TABLE_MAGIC:
        raise ValueError(f'{path}: not an SSTable (bad magic number)')
    _, p = _varint(footer, 0); _, p = _varint(footer, p)            # metaindex handle (unused)
    idx_off, p = _varint(footer, p); idx_size, p = _varint(footer, p)
    out = {}
    for _, handle in _block_entries(_read_block(data, idx_off, idx_size)):
        off, q = _varint(handle, 0); size, q = _varint(handle, q)
        for k, v in _block_entries(_read_block(data, off, size)):
            out[k] = v
    return out


# ---- the bundle ----------------------------------------------------------------------------------------------------------
def read_bundle(prefix, want=None):
    """{tensor name: float32 array} of the bundle `prefix` (.index + .data-XXXXX-of-YYYYY); every tensor's CRC is
    verified.  `want`: optional predicate on the name (skip the optimizer slots without reading them)."""
    table = read_table(prefix + '.index')
    if b'' not in table:
        raise ValueError(f'{prefix}.index: no bundle header entry')
    header = parse_bundle_header(table[b''])
    if header['endianness'] != 0:
        raise NotImplementedError('big-endian bundle')
    shards = {}
    out = {}
    for key, val in table.items():
        if key == b'':
            continue
        name = key.decode()
        if want is not None and not want(name):
            continue
        e = parse_bundle_entry(val)
        if e['slices']:
            raise NotImplementedError(f'{name}: partitioned (sliced) variable')
        if e['dtype'] != DT_FLOAT:
            continue                                   # save counters, object graph (DT_STRING / DT_INT64): not weights
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = np.memmap('%s.data-%05d-of-%05d' % (prefix, sid, header['num_shards']), dtype=np.uint8, mode='r')
        raw = shards[sid][e['offset']:e['offset'] + e['size']]
        n = int(np.prod(e['shape'])) if e['shape'] else 1
        if raw.size != e['size'] or e['size'] != 4 * n:
            raise ValueError(f'{name}: {e["size"]} bytes for shape {e["shape"]} (data file truncated?)')
        if mask_crc(crc32c(np.ascontiguousarray(raw))) != e['crc32c']:
            raise ValueError(f'{name}: CRC-32C mismatch (corrupt data file)')
        out[name] = np.frombuffer(np.ascontiguousarray(raw).tobytes(), dtype='<f4').reshape(e['shape'])
    return out


_CONV = re.compile(r'^model/front_conv/layer_with_weights-(\d+)/(?:(conv2d_1x3|conv2d_3x1|BN_1x3|BN_3x1)|forward/layer_with_weights-([0-3]))/'
                   r'(kernel|bias|gamma|beta|moving_mean|moving_variance)' + re.escape(SUFFIX) + '$')      # (moving_*: MODEL.BN = 'batch_norm')
_DIV = re.compile(r'^model/div_enc/split_fc_layers/(\d+)/layer_with_weights-([01])/(kernel|bias)' + re.escape(SUFFIX) + '$')
_FWD = {0: 'conv2d_1x3', 1: 'BN_1x3', 2: 'conv2d_3x1', 3: 'BN_3x1'}      # ConvLayer.forward's layers with weights (nnfp.py:69-75)


def state_dict_from_tf_checkpoint(prefix, names, shapes, emb_sz):
    """The encoder's state dict (keys `names`, shapes `shapes`: nnfp.tensor_names() / library order) from the reference's
    checkpoint `prefix` = .../ckpt-<N>.  Raises if any variable is missing, duplicated or mis-shaped."""
    # optimizer slots hang off the variables they belong to (`<variable>/.OPTIMIZER_SLOT/optimizer/{m,v}/...`): not weights
    tensors = read_bundle(prefix, want=lambda n: n.startswith('model/') and n.endswith(SUFFIX) and '/.OPTIMIZER_SLOT/' not in n)
    got, div = {}, {}
    unrecognised = []
    for name, arr in tensors.items():
        m = _CONV.match(name)
        if m:
            blk, attr, fwd, kind = int(m.group(1)), m.group(2), m.group(3), m.group(4)
            attr = attr or _FWD[int(fwd)]
            if attr.startswith('conv') != (kind in ('kernel', 'bias')):
                raise ValueError(f'{name}: `{kind}` does not belong to `{attr}`')
            key = f'front_conv.{blk}.{attr}.{kind}'
            if key in got:
                raise ValueError(f'{key}: found twice in the checkpoint')
            got[key] = arr
            continue
        m = _DIV.match(name)
        if m:
            k2 = (int(m.group(1)), 'fc1' if m.group(2) == '0' else 'fc2', m.group(3))
            if k2 in div:
                raise ValueError(f'{name}: found twice in the checkpoint')
            div[k2] = arr
            continue
        unrecognised.append(name)
    missing_div = []
    for fc in ('fc1', 'fc2'):
        for kind in ('kernel', 'bias'):
            parts = [div.get((q, fc, kind)) for q in range(emb_sz)]
            if any(p is None for p in parts):
                missing_div.append(f'div_enc.{fc}.{kind} ({sum(p is None for p in parts)} of {emb_sz} slices)')
            else:
                got[f'div_enc.{fc}.{kind}'] = np.stack(parts)
    missing = [n for n in names if n not in got]
    if unrecognised or missing:
        # the object-graph key names are the one part of this reader no TensorFlow-written file has confirmed: when they do not line
        # up, say so with BOTH sides on the table instead of stopping at the first missing variable
        def _some(xs):
            return ', '.join(xs[:6]) + (f', ... ({len(xs)} in all)' if len(xs) > 6 else '')
        raise KeyError(f'{prefix}: the checkpoint and the encoder do not name the same variables.\n'
                       f'  encoder variables with no tensor in the checkpoint ({len(missing)}): {_some(missing) or "none"}'
                       + (f'\n  incomplete divide-and-encode stacks: {_some(missing_div)}' if missing_div else '') +
                       f'\n  model/... tensors of the checkpoint this reader has no variable for ({len(unrecognised)}): {_some(sorted(unrecognised)) or "none"}\n'
                       f'  (expected keys: model/front_conv/layer_with_weights-<b>/{{conv2d_1x3,BN_1x3,conv2d_3x1,BN_3x1}}/<kind>{SUFFIX}, '
                       f'model/div_enc/split_fc_layers/<q>/layer_with_weights-{{0,1}}/{{kernel,bias}}{SUFFIX})')
    if len(div) != 4 * emb_sz:
        raise ValueError('more divide-and-encode slices in the checkpoint than EMB_SZ')
    sd = {}
    for n, shp in zip(names, shapes):
        if n not in got:
            raise KeyError(f'{n}: not in the checkpoint')
        a = got.pop(n)
        if tuple(a.shape) != tuple(shp):
            raise ValueError(f'{n}: shape {tuple(a.shape)} in the checkpoint, {tuple(shp)} expected')
        sd[n] = np.ascontiguousarray(a, dtype=np.float32)
    if got:
        raise ValueError(f'variables the encoder does not have: {sorted(got)[:4]}')
    return sd


def has_tf_checkpoint(prefix):
    return os.path.exists(prefix + '.index')
