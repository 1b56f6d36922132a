"""The record files in front of the training path, without TensorFlow: the
TFRecord container, the tf.train.Example message with the three byte features
the reference writes (utils/tfrecord_voc_utils.py:55-59), the PASCAL VOC
annotation reader (utils/tfrecord_voc_utils.py:33-62) and the batch generator
(utils/tfrecord_voc_utils.py:95-120).

Formats (both public, restated here):
  * TFRecord: per record  uint64 length | uint32 masked_crc32c(length) | data |
    uint32 masked_crc32c(data), little endian; masked = rotr(crc, 15) + 0xa282ead8.
  * tf.train.Example: Example{1: Features{1: map<string, Feature>}},
    Feature{1: BytesList{1: repeated bytes} | 2: FloatList{1: packed float} |
    3: Int64List{1: packed varint}}.
The reference's float_feature / int64_feature helpers pass their lists as
bytes_list and cannot work (utils/tfrecord_voc_utils.py:15-30); only
bytes_feature is used, with 'image' = the encoded JPEG file, 'shape' =
int32[3] (h, w, depth) bytes, 'ground_truth' = float32[n, 5] bytes of
(ymin, ymax, xmin, xmax, class_id).
"""
import glob
import io
import math
import os
import struct
import sys
import warnings
import xml.etree.ElementTree as ElementTree

import numpy as np

from .wire import classname_to_ids

# ----------------------------------------------------------------- crc32c

_CRC_POLY = 0x82F63B78          # Castagnoli, reflected


def _make_tables():
    t0 = []
    for n in range(256):
        c = n
        for _ in range(8):
            c = (c >> 1) ^ _CRC_POLY if c & 1 else c >> 1
        t0.append(c)
    tables = [t0]
    for k in range(1, 8):
        prev = tables[k - 1]
        tables.append([(prev[n] >> 8) ^ t0[prev[n] & 0xFF] for n in range(256)])
    return tables


_T = _make_tables()


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli) of a bytes-like object, eight bytes per step."""
    t0, t1, t2, t3, t4, t5, t6, t7 = _T
    crc ^= 0xFFFFFFFF
    view = memoryview(data).cast('B')
    n8 = len(view) // 8
    if n8:
        for lo, hi in struct.iter_unpack('<II', view[:n8 * 8]):
            lo ^= crc
            crc = (t7[lo & 0xFF] ^ t6[(lo >> 8) & 0xFF] ^ t5[(lo >> 16) & 0xFF] ^ t4[lo >> 24] ^
                   t3[hi & 0xFF] ^ t2[(hi >> 8) & 0xFF] ^ t1[(hi >> 16) & 0xFF] ^ t0[hi >> 24])
    for b in view[n8 * 8:]:
        crc = t0[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data):
    crc = crc32c(data)
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ----------------------------------------------------------------- TFRecord container

class TFRecordWriter:
    """tf.python_io.TFRecordWriter stand-in (context manager, write(bytes))."""

    def __init__(self, path):
        self._f = open(path, 'wb')

    def write(self, record):
        header = struct.pack('<Q', len(record))
        self._f.write(header)
        self._f.write(struct.pack('<I', masked_crc32c(header)))
        self._f.write(record)
        self._f.write(struct.pack('<I', masked_crc32c(record)))

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_tfrecords(paths, verify=False):
    """Yields the records of one file or a list of files, in order.  verify=True
    checks both checksums of every record (slow in Python; the length checksum is
    always checked: it is what tells a TFRecord file from anything else)."""
    if isinstance(paths, (str, bytes, os.PathLike)):
        paths = [paths]
    for path in paths:
        with open(path, 'rb') as f:
            while True:
                header = f.read(8)
                if not header:
                    break
                if len(header) != 8:
                    raise ValueError('%s: truncated record header' % path)
                (length_crc,) = struct.unpack('<I', f.read(4))
                if masked_crc32c(header) != length_crc:
                    raise ValueError('%s: corrupt record length' % path)
                (length,) = struct.unpack('<Q', header)
                data = f.read(length)
                tail = f.read(4)
                if len(data) != length or len(tail) != 4:
                    raise ValueError('%s: truncated record' % path)
                if verify and masked_crc32c(data) != struct.unpack('<I', tail)[0]:
                    raise ValueError('%s: corrupt record data' % path)
                yield data


# ----------------------------------------------------------------- tf.train.Example

def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(number, payload):
    """A length-delimited field."""
    return _varint((number << 3) | 2) + _varint(len(payload)) + payload


def _read_varint(buf, pos):
    shift = value = 0
    while True:
        b = buf[pos]
        pos += 1
        value |= (b & 0x7F) << shift
        if not b & 0x80:
            return value, pos
        shift += 7


def _fields(buf):
    """(field number, wire type, value) of every field of a message."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _read_varint(buf, pos)
        number, wire = key >> 3, key & 7
        if wire == 0:
            value, pos = _read_varint(buf, pos)
        elif wire == 1:
            value, pos = buf[pos:pos + 8], pos + 8
        elif wire == 2:
            length, pos = _read_varint(buf, pos)
            value, pos = buf[pos:pos + length], pos + length
        elif wire == 5:
            value, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wire)
        yield number, wire, value


def encode_example(features):
    """features: {name: bytes | list of bytes | float array | int array} ->
    serialized tf.train.Example.  Names are written in sorted order."""
    entries = b''
    for name in sorted(features):
        value = features[name]
        if isinstance(value, (bytes, bytearray)):
            value = [bytes(value)]
        if isinstance(value, (list, tuple)) and all(isinstance(v, (bytes, bytearray)) for v in value):
            feature = _field(1, b''.join(_field(1, bytes(v)) for v in value))
        else:
            arr = np.asarray(value)
            if arr.dtype.kind == 'f':
                feature = _field(2, _field(1, arr.astype('<f4').tobytes()))
            elif arr.dtype.kind in 'iu':
                feature = _field(3, _field(1, b''.join(_varint(int(v) & 0xFFFFFFFFFFFFFFFF) for v in arr.reshape(-1))))
            else:
                raise TypeError('feature %r: unsupported value type' % name)
        entries += _field(1, _field(1, name.encode('utf-8')) + _field(2, feature))
    return _field(1, entries)


def decode_example(record):
    """Serialized tf.train.Example -> {name: list of bytes | float32 array | int64 array}."""
    out = {}
    for number, _, features in _fields(bytes(record)):
        if number != 1:
            continue
        for n2, _, entry in _fields(features):
            if n2 != 1:
                continue
            name, feature = None, b''
            for n3, _, value in _fields(entry):
                if n3 == 1:
                    name = value.decode('utf-8')
                elif n3 == 2:
                    feature = value
            for kind, _, payload in _fields(feature):
                if kind == 1:
                    out[name] = [v for n4, _, v in _fields(payload) if n4 == 1]
                elif kind == 2:
                    chunks = [np.frombuffer(v, '<f4') for n4, _, v in _fields(payload) if n4 == 1]
                    out[name] = np.concatenate(chunks) if chunks else np.zeros(0, np.float32)
                elif kind == 3:
                    vals = []
                    for n4, wire, v in _fields(payload):
                        if n4 != 1:
                            continue
                        if wire == 0:
                            vals.append(v)
                        else:
                            pos = 0
                            while pos < len(v):
                                x, pos = _read_varint(v, pos)
                                vals.append(x)
                    out[name] = np.asarray([x - (1 << 64) if x >> 63 else x for x in vals], dtype=np.int64)
    return out


# ----------------------------------------------------------------- PASCAL VOC -> records

def xml_to_example(xmlpath, imgpath, name_to_id=classname_to_ids):
    """utils/tfrecord_voc_utils.py:33-62: one annotation file and its image ->
    serialized Example.  ground_truth rows are (ymin, ymax, xmin, xmax, class_id)."""
    root = ElementTree.parse(xmlpath).getroot()
    imgname = os.path.join(imgpath, root.find('filename').text)
    with open(imgname, 'rb') as f:
        image = f.read()
    size = root.find('size')
    shape = np.asarray([int(size.find('height').text), int(size.find('width').text), int(size.find('depth').text)],
                       np.int32)
    objects = list(root.iter('object'))                       # '//object': at any depth
    ground_truth = np.zeros([len(objects), 5], np.float32)
    for i, obj in enumerate(objects):
        box = obj.find('bndbox')
        ground_truth[i, :] = np.asarray([float(box.find('ymin').text), float(box.find('ymax').text),
                                         float(box.find('xmin').text), float(box.find('xmax').text),
                                         name_to_id[obj.find('name').text]], np.float32)
    return encode_example({'image': image, 'shape': shape.tobytes(), 'ground_truth': ground_truth.tobytes()})


def dataset2tfrecord(xml_dir, img_dir, output_dir, name, total_shards=5):
    """utils/tfrecord_voc_utils.py:65-92, shard arithmetic included
    (num_per_shard = int(ceil(len) / total_shards): the division truncates, so up to
    total_shards - 1 trailing annotations are left out, as in the reference)."""
    if not os.path.exists(output_dir):
        os.makedirs(output_dir)
        print(output_dir, 'does not exist, create it done')
    elif len(os.listdir(output_dir)) == 0:
        print(output_dir, 'already exist, need not create new')
    else:
        warnings.warn(output_dir + ' is not empty!', UserWarning)
    outputfiles = []
    xmllist = sorted(glob.glob(os.path.join(xml_dir, '*.xml')))
    num_per_shard = int(math.ceil(len(xmllist)) / float(total_shards))
    for shard_id in range(total_shards):
        outputname = os.path.join(output_dir, '%s_%05d-of-%05d.tfrecord' % (name, shard_id + 1, total_shards))
        outputfiles.append(outputname)
        with TFRecordWriter(outputname) as writer:
            for i in range(shard_id * num_per_shard, min((shard_id + 1) * num_per_shard, len(xmllist))):
                sys.stdout.write('\r>> Converting image %d/%d shard %d/%d' % (i + 1, len(xmllist), shard_id + 1, total_shards))
                sys.stdout.flush()
                writer.write(xml_to_example(xmllist[i], img_dir))
            sys.stdout.write('\n')
            sys.stdout.flush()
    return outputfiles


# ----------------------------------------------------------------- records -> batches

def parse_record(record):
    """utils/tfrecord_voc_utils.py:95-106: (image float32 [h, w, 3], ground_truth
    float32 [n, 5] of (ymin, ymax, xmin, xmax, class_id), shape int32[3])."""
    from PIL import Image
    features = decode_example(record)
    shape = np.frombuffer(features['shape'][0], np.int32).reshape(3)
    ground_truth = np.frombuffer(features['ground_truth'][0], np.float32).reshape(-1, 5)
    with Image.open(io.BytesIO(features['image'][0])) as im:
        image = np.asarray(im.convert('RGB'), dtype=np.float32)
    return image.reshape(shape), ground_truth, shape


def parse_function(record, config, rng=None):
    """Record -> (image, ground_truth [pad_truth_to, 5]) through the augmentor."""
    from .augment import image_augmentor
    image, ground_truth, shape = parse_record(record)
    return image_augmentor(image=image, input_shape=shape, ground_truth=ground_truth, rng=rng, **config)


def get_generator(tfrecords, batch_size, buffer_size, image_preprocess_config, seed=0):
    """utils/tfrecord_voc_utils.py:115-120: map(parse + augment) -> shuffle(buffer) ->
    batch(drop_remainder) -> repeat, as (initializer, iterator): calling the
    initializer restarts the stream, next(iterator) yields (images [B, ...] float32,
    ground_truth [B, pad_truth_to, 5] float32) NumPy arrays - the pair the
    YOLOv2 / YOLOv3 classes take as data_provider['train_generator']."""
    state = {'gen': None, 'epoch': 0}

    def stream():
        rng = np.random.default_rng([seed, state['epoch']])
        while True:                                           # repeat()
            buffer, images, truths = [], [], []

            def emit(sample):
                images.append(np.asarray(sample[0], np.float32))
                truths.append(np.asarray(sample[1], np.float32))
                if len(images) == batch_size:
                    batch = (np.stack(images), np.stack(truths))
                    del images[:], truths[:]
                    return batch
                return None
            for record in read_tfrecords(tfrecords):
                buffer.append(parse_function(record, image_preprocess_config, rng))
                if len(buffer) > max(1, buffer_size) - 1:
                    batch = emit(buffer.pop(int(rng.integers(len(buffer)))))
                    if batch is not None:
                        yield batch
            while buffer:
                batch = emit(buffer.pop(int(rng.integers(len(buffer)))))
                if batch is not None:
                    yield batch
            # drop_remainder: a partial last batch of the pass is discarded

    def initializer():
        state['epoch'] += 1
        state['gen'] = stream()

    class _Iterator:
        def __iter__(self):
            return self

        def __next__(self):
            if state['gen'] is None:
                initializer()
            return next(state['gen'])

    return initializer, _Iterator()
