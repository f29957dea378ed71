"""CPU: the protobuf layer of the TensorFlow-checkpoint reader against a parser this repository did not write.

`model/utils/tf_checkpoint.py` decodes BundleHeaderProto / BundleEntryProto with its own 30-line wire-format reader, and until now
only ever read messages produced by the writer of tests/test_tf_checkpoint.py -- same author on both sides.  Here the messages are
built and parsed by `google.protobuf` (7.x, in the image) from descriptors that restate the PUBLISHED schema
(tensorflow/core/protobuf/tensor_bundle.proto, tensorflow/core/framework/{tensor_shape,tensor_slice,versions}.proto -- field numbers,
types, nesting), and the two decoders must agree in both directions:

  * messages serialised by google.protobuf (random shapes incl. scalars and zero-size dims, 64-bit offsets, every field set, the
    optional fields in any order protobuf chooses) -> `parse_bundle_entry` / `parse_bundle_header` return the same values;
  * the bytes the test writer emits -> google.protobuf parses them to the same values (so the writer the round-trip test relies on
    speaks the schema too).

The row stays "partial" until a TF-written file exists: the object-graph KEY NAMES are what a real file would pin."""
import importlib.util
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _messages():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name='nafp_tensor_bundle_schema.proto', package='tensorflow_schema', syntax='proto3')

    def msg(parent, name):
        m = parent.message_type.add() if hasattr(parent, 'message_type') else parent.nested_type.add()
        m.name = name
        return m

    def field(m, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None):
        f = m.field.add(name=name, number=number, type=ftype, label=label)
        if type_name:
            f.type_name = type_name
        return f

    # tensor_shape.proto: message TensorShapeProto { message Dim { int64 size = 1; string name = 2; } repeated Dim dim = 2; bool unknown_rank = 3; }
    shape = msg(fd, 'TensorShapeProto')
    dim = msg(shape, 'Dim')
    field(dim, 'size', 1, F.TYPE_INT64); field(dim, 'name', 2, F.TYPE_STRING)
    field(shape, 'dim', 2, F.TYPE_MESSAGE, F.LABEL_REPEATED, '.tensorflow_schema.TensorShapeProto.Dim')
    field(shape, 'unknown_rank', 3, F.TYPE_BOOL)
    # tensor_slice.proto: message TensorSliceProto { message Extent { int64 start = 1; oneof has_length { int64 length = 2; } } repeated Extent extent = 1; }
    sl = msg(fd, 'TensorSliceProto')
    ext = msg(sl, 'Extent')
    field(ext, 'start', 1, F.TYPE_INT64); field(ext, 'length', 2, F.TYPE_INT64)
    field(sl, 'extent', 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, '.tensorflow_schema.TensorSliceProto.Extent')
    # versions.proto: message VersionDef { int32 producer = 1; int32 min_consumer = 2; repeated int32 bad_consumers = 3; }
    ver = msg(fd, 'VersionDef')
    field(ver, 'producer', 1, F.TYPE_INT32); field(ver, 'min_consumer', 2, F.TYPE_INT32)
    field(ver, 'bad_consumers', 3, F.TYPE_INT32, F.LABEL_REPEATED)
    # tensor_bundle.proto: BundleHeaderProto { int32 num_shards = 1; Endianness endianness = 2 (enum: LITTLE = 0, BIG = 1); VersionDef version = 3; }
    hd = msg(fd, 'BundleHeaderProto')
    field(hd, 'num_shards', 1, F.TYPE_INT32); field(hd, 'endianness', 2, F.TYPE_INT32)      # (an enum travels as a varint)
    field(hd, 'version', 3, F.TYPE_MESSAGE, type_name='.tensorflow_schema.VersionDef')
    # BundleEntryProto { DataType dtype = 1; TensorShapeProto shape = 2; int32 shard_id = 3; int64 offset = 4; int64 size = 5; fixed32 crc32c = 6;
    #                    repeated TensorSliceProto slices = 7; }
    en = msg(fd, 'BundleEntryProto')
    field(en, 'dtype', 1, F.TYPE_INT32)                                                      # (DataType enum: DT_FLOAT = 1)
    field(en, 'shape', 2, F.TYPE_MESSAGE, type_name='.tensorflow_schema.TensorShapeProto')
    field(en, 'shard_id', 3, F.TYPE_INT32); field(en, 'offset', 4, F.TYPE_INT64); field(en, 'size', 5, F.TYPE_INT64)
    field(en, 'crc32c', 6, F.TYPE_FIXED32)
    field(en, 'slices', 7, F.TYPE_MESSAGE, F.LABEL_REPEATED, '.tensorflow_schema.TensorSliceProto')
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName('tensorflow_schema.' + n))      # noqa: E731
    return get('BundleHeaderProto'), get('BundleEntryProto')


@pytest.fixture(scope='module')
def tfc():
    import neural_audio_fp_amd
    from neural_audio_fp_amd.model.utils import tf_checkpoint
    return tf_checkpoint


def test_reader_agrees_with_google_protobuf_on_protobuf_written_messages(tfc):
    Header, Entry = _messages()
    rng = np.random.default_rng(0)
    for trial in range(200):
        e = Entry()
        e.dtype = int(rng.choice([1, 1, 1, 7, 9]))                        # DT_FLOAT mostly; DT_STRING / DT_INT64 (object graph, save counter)
        rank = int(rng.integers(0, 5))
        dims = [int(rng.choice([0, 1, 3, 128, 1024, 70000])) for _ in range(rank)]
        for d in dims:
            e.shape.dim.add().size = d
        e.shard_id = int(rng.integers(0, 3))
        e.offset = int(rng.choice([0, 1, 2 ** 31 + 5, 2 ** 40 + 17]))
        e.size = int(rng.choice([0, 4, 4 * 16939008, 2 ** 33]))
        e.crc32c = int(rng.integers(0, 2 ** 32))
        n_sl = int(rng.choice([0, 0, 0, 2]))
        for _ in range(n_sl):
            s = e.slices.add()
            x = s.extent.add(); x.start = 3; x.length = 9
        got = tfc.parse_bundle_entry(e.SerializeToString())
        assert got['dtype'] == e.dtype and got['shape'] == dims and got['shard_id'] == e.shard_id, (trial, got)
        assert got['offset'] == e.offset and got['size'] == e.size and got['crc32c'] == e.crc32c and got['slices'] == n_sl, (trial, got)
    for num_shards, endian, producer in ((1, 0, 1), (4, 0, 1), (1, 1, 1), (2, 0, 0)):
        h = Header()
        h.num_shards = num_shards; h.endianness = endian; h.version.producer = producer
        got = tfc.parse_bundle_header(h.SerializeToString())
        assert got['num_shards'] == num_shards and got['endianness'] == endian


def test_google_protobuf_parses_what_the_test_writer_emits(tfc):
    Header, Entry = _messages()
    spec = importlib.util.spec_from_file_location('_tfck_writer', os.path.join(ROOT, 'tests', 'test_tf_checkpoint.py'))
    w = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(w)
    for shape, offset, size, crc in (((1, 3, 1, 128), 0, 1536, 0xdeadbeef), ((128,), 1536, 512, 1), ((), 7, 4, 2 ** 32 - 1), ((256, 16, 128), 2 ** 33, 2097152, 5)):
        raw = w._entry_proto(shape, offset, size, crc)
        e = Entry()
        e.ParseFromString(raw)
        assert e.dtype == 1 and [d.size for d in e.shape.dim] == list(shape) and e.offset == offset and e.size == size and e.crc32c == crc
        assert e.shard_id == 0 and len(e.slices) == 0
        assert tfc.parse_bundle_entry(raw) == {'dtype': 1, 'shape': list(shape), 'shard_id': 0, 'offset': offset, 'size': size, 'crc32c': crc, 'slices': 0}
    h = Header()
    h.ParseFromString(b'\x08\x01\x1a\x02\x08\x01')                       # the header bytes the test writer stores under the empty key
    assert h.num_shards == 1 and h.endianness == 0 and h.version.producer == 1
    assert struct.pack('<I', 1) == b'\x01\x00\x00\x00'                     # (fixed32 is little-endian on the wire)
