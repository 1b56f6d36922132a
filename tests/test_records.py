"""The record files and the augmentor in front of the training path (records.py,
augment.py).  Known answers: the CRC-32C check value, hand-assembled protobuf bytes,
hand-computed box arithmetic of utils/image_augmentor.py."""
import os
import struct

import numpy as np
import pytest
import torch

import yolo_tf_amd as y
from yolo_tf_amd import augment, records


def test_crc32c_known_answers():
    assert records.crc32c(b'') == 0
    assert records.crc32c(b'123456789') == 0xE3069283           # the CRC catalogue's check value
    assert records.crc32c(bytes(32)) == 0x8A9136AA              # RFC 3720 B.4: 32 zero bytes
    assert records.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43     # RFC 3720 B.4: 32 bytes of ones
    assert records.crc32c(bytes(range(32))) == 0x46DD794E       # RFC 3720 B.4: ascending bytes
    data = os.urandom(1000)
    assert records.crc32c(data[7:]) == records.crc32c(data[500:], records.crc32c(data[7:500]))


def test_tfrecord_framing(tmp_path):
    path = str(tmp_path / 'a.tfrecord')
    payloads = [b'123456789', b'', os.urandom(300)]
    with records.TFRecordWriter(path) as w:
        for p in payloads:
            w.write(p)
    raw = open(path, 'rb').read()
    # first record by hand: length, masked crc of the length bytes, data, masked crc of the data
    crc = 0xE3069283
    masked = ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF
    assert raw[:8] == struct.pack('<Q', 9) and raw[12:21] == b'123456789'
    assert raw[21:25] == struct.pack('<I', masked)
    assert list(records.read_tfrecords(path, verify=True)) == payloads
    broken = bytearray(raw)
    broken[14] ^= 1
    open(path, 'wb').write(bytes(broken))
    with pytest.raises(ValueError):
        list(records.read_tfrecords(path, verify=True))
    broken = bytearray(raw)
    broken[0] ^= 1
    open(path, 'wb').write(bytes(broken))
    with pytest.raises(ValueError):
        list(records.read_tfrecords(path))


def test_example_wire_format():
    # Example{1: Features{1: entry{1: 'a', 2: Feature{1: BytesList{1: 'xy'}}}}}
    want = bytes([0x0A, 0x0D, 0x0A, 0x0B, 0x0A, 0x01, 0x61, 0x12, 0x06, 0x0A, 0x04, 0x0A, 0x02, 0x78, 0x79])
    assert records.encode_example({'a': b'xy'}) == want
    assert records.decode_example(want) == {'a': [b'xy']}
    ex = records.encode_example({'img': [b'\x00\x01', b''], 'f': np.float32([1.5, -2.0]), 'i': np.int64([3, -1, 1 << 40])})
    got = records.decode_example(ex)
    assert got['img'] == [b'\x00\x01', b'']
    assert got['f'].tolist() == [1.5, -2.0] and got['i'].tolist() == [3, -1, 1 << 40]


VOC_XML = '''<annotation><folder>VOC</folder><filename>%s</filename>
<size><width>%d</width><height>%d</height><depth>3</depth></size>
<object><name>dog</name><bndbox><xmin>10</xmin><ymin>20</ymin><xmax>50</xmax><ymax>40</ymax></bndbox>
  <part><name>head</name></part></object>
<object><name>person</name><bndbox><xmin>5</xmin><ymin>6</ymin><xmax>30</xmax><ymax>60</ymax></bndbox></object>
</annotation>'''


def _make_voc(tmp_path, n, h=64, w=96):
    from PIL import Image
    xml_dir, img_dir = tmp_path / 'xml', tmp_path / 'img'
    xml_dir.mkdir(); img_dir.mkdir()
    rng = np.random.default_rng(0)
    for i in range(n):
        name = 'im%03d.jpg' % i
        Image.fromarray(rng.integers(0, 255, size=(h, w, 3), dtype=np.uint8)).save(str(img_dir / name), quality=95)
        (xml_dir / ('im%03d.xml' % i)).write_text(VOC_XML % (name, w, h))
    return str(xml_dir), str(img_dir)


def test_voc_to_records_and_back(tmp_path):
    xml_dir, img_dir = _make_voc(tmp_path, 7)
    files = records.dataset2tfrecord(xml_dir, img_dir, str(tmp_path / 'out'), 'voc', total_shards=2)
    assert [os.path.basename(f) for f in files] == ['voc_00001-of-00002.tfrecord', 'voc_00002-of-00002.tfrecord']
    recs = list(records.read_tfrecords(files, verify=True))
    assert len(recs) == 6                                      # int(ceil(7) / 2.) = 3 per shard: the 7th is left out
    image, gt, shape = records.parse_record(recs[0])
    assert shape.tolist() == [64, 96, 3] and image.shape == (64, 96, 3) and image.dtype == np.float32
    dog, person = y.wire.classname_to_ids['dog'], y.wire.classname_to_ids['person']
    assert gt.tolist() == [[20., 40., 10., 50., dog], [6., 60., 5., 30., person]]      # ymin ymax xmin xmax class


def test_generator_batches_shuffle_and_restart(tmp_path):
    xml_dir, img_dir = _make_voc(tmp_path, 10)
    files = records.dataset2tfrecord(xml_dir, img_dir, str(tmp_path / 'out'), 'voc', total_shards=1)
    cfg = dict(data_format='channels_last', output_shape=[32, 32], pad_truth_to=6)
    init, it = records.get_generator(files, batch_size=4, buffer_size=5, image_preprocess_config=cfg, seed=3)
    init()
    batches = [next(it) for _ in range(5)]                    # 10 samples -> 2 batches per pass, then it repeats
    for images, gt in batches:
        assert images.shape == (4, 32, 32, 3) and images.dtype == np.float32
        assert gt.shape == (4, 6, 5) and gt.dtype == np.float32
        assert (gt[:, 2:, :] == -1).all() and (gt[:, :2, 0] > 0).all()
    init()
    again = next(it)
    assert again[0].shape == (4, 32, 32, 3)


def test_resize_without_aspect_ratio_scales_boxes():
    image = np.arange(4 * 6 * 3, dtype=np.float32).reshape(4, 6, 3)
    gt = np.float32([[1., 3., 2., 5., 7.]])                    # ymin ymax xmin xmax class
    out, block = augment.image_augmentor(image, [4, 6, 3], 'channels_last', [8, 9], ground_truth=gt, pad_truth_to=3)
    assert out.shape == (8, 9, 3)
    # align_corners=True keeps the corner pixels
    assert np.allclose(out[0, 0], image[0, 0]) and np.allclose(out[-1, -1], image[-1, -1])
    ry, rx = np.float32(8 / 4), np.float32(9 / 6)
    ymin, ymax, xmin, xmax = 1 * ry, 3 * ry, 2 * rx, 5 * rx
    assert np.allclose(block[0], [(ymin + ymax) / 2, (xmin + xmax) / 2, ymax - ymin, xmax - xmin, 7.])
    assert (block[1:] == -1).all()


def test_keep_aspect_ratio_pads_bottom_right_and_center_crop():
    image = np.ones((10, 20, 3), dtype=np.float32)
    gt = np.float32([[2., 6., 4., 12., 1.]])
    out, block = augment.image_augmentor(image, [10, 20, 3], 'channels_last', [30, 30], zoom_size=[40, 40],
                                         crop_method='center', keep_aspect_ratios=True, constant_values=0.,
                                         ground_truth=gt, pad_truth_to=2)
    # ratio = 40 / 20 = 2: resized to 20 x 40, padded to 40 x 40 below, centre crop offset 5
    assert out.shape == (30, 30, 3)
    assert np.allclose(out[:15], 1., atol=1e-6) and (out[15:] == 0).all()
    # ymin = 2 * 2 - 5 = -1 is clamped to 0 before centre and height are taken (image_augmentor.py:216-227)
    assert np.allclose(block[0], [(0 + 7) / 2, (3 + 19) / 2, 7., 16., 1.])


def test_flips_map_coordinates_as_the_reference_does():
    image = np.arange(6 * 8 * 3, dtype=np.float32).reshape(6, 8, 3)
    gt = np.float32([[1., 2., 1., 4., 0.]])
    out, block = augment.image_augmentor(image, [6, 8, 3], 'channels_last', [6, 8], flip_prob=[1., 1.],
                                         ground_truth=gt, pad_truth_to=2)
    assert np.array_equal(out, image[::-1, ::-1])
    # ymax, ymin = 6 - 1 - 1, 6 - 2 - 1; xmax, xmin = 8 - 1 - 1, 8 - 4 - 1
    assert np.allclose(block[0], [(4 + 3) / 2, (6 + 3) / 2, 1., 3., 0.])
    same, block2 = augment.image_augmentor(image, [6, 8, 3], 'channels_last', [6, 8], flip_prob=[0., 0.],
                                           ground_truth=gt, pad_truth_to=2)
    assert np.array_equal(same, image) and np.allclose(block2[0], [1.5, 2.5, 1., 3., 0.])


def test_identity_jitter_rotation_and_channels_first():
    rng = np.random.default_rng(5)
    image = rng.uniform(0, 255, size=(3, 12, 10)).astype(np.float32)
    out = augment.image_augmentor(image, [12, 10, 3], 'channels_first', [12, 10], rotate=[1., 0., 0.])
    assert out.shape == (3, 12, 10) and np.allclose(out, image, atol=1e-3)
    hwc = image.transpose(1, 2, 0)
    assert np.allclose(augment.adjust_hue(hwc, 0.), hwc, atol=1e-3)
    shifted = augment.adjust_hue(hwc, 1. / 3.)                 # a third of a turn: R -> G -> B
    assert np.allclose(shifted[..., 1], hwc[..., 0], atol=1e-2) and np.allclose(shifted[..., 2], hwc[..., 1], atol=1e-2)
    quarter = augment.rotate_image(hwc[:10], np.pi / 2)        # square image, a quarter turn counter-clockwise
    assert np.allclose(quarter, np.rot90(hwc[:10], 1), atol=1e-2)


def test_all_boxes_dropped_leaves_an_all_pad_block():
    image = np.zeros((8, 8, 3), np.float32)
    gt = np.float32([[-6., -2., 1., 3., 4.]])                  # centre above the image
    _, block = augment.image_augmentor(image, [8, 8, 3], 'channels_last', [8, 8], ground_truth=gt, pad_truth_to=3)
    assert (block == -1).all()


def test_argument_checks():
    image = np.zeros((4, 4, 3), np.float32)
    with pytest.raises(Exception):
        augment.image_augmentor(image, [4, 4, 3], 'nhwc', [4, 4])
    with pytest.raises(Exception):
        augment.image_augmentor(image, [4, 4, 3], 'channels_last', [4, 4], fill_mode='LANCZOS')
    with pytest.raises(Exception):
        augment.image_augmentor(image, [4, 4, 3], 'channels_last', [4, 4], zoom_size=[8, 8], crop_method='corner')
    with pytest.raises(Exception):
        augment.image_augmentor(image, [4, 4, 3], 'channels_last', [4, 4], rotate=[0.5, 3., 1.])


def test_draw_detections_marks_the_box_outline():
    image = np.zeros((40, 60, 3), np.float32)
    canvas = y.wire.draw_detections(image, [0.9], [[10., 20., 30., 50.]], [11])       # y1 x1 y2 x2
    px = np.asarray(canvas)
    assert px.shape == (40, 60, 3)
    assert tuple(px[30, 35]) == (255, 0, 0) and tuple(px[20, 50]) == (255, 0, 0)       # bottom and right edges
    assert tuple(px[25, 35]) == (0, 0, 0) and tuple(px[5, 5]) == (0, 0, 0)             # interior, outside


@pytest.mark.gpu
def test_records_to_training_step_end_to_end(tmp_path):
    torch.manual_seed(0)
    xml_dir, img_dir = _make_voc(tmp_path, 4)
    files = records.dataset2tfrecord(xml_dir, img_dir, str(tmp_path / 'out'), 'voc', total_shards=1)
    aug = dict(data_format='channels_last', output_shape=[96, 96], flip_prob=[0., 0.5], pad_truth_to=5)
    gen = records.get_generator(files, batch_size=2, buffer_size=4, image_preprocess_config=aug, seed=1)
    cfg = dict(mode='train', data_shape=[96, 96, 3], num_classes=20, weight_decay=1e-4, keep_prob=0.5,
               data_format='channels_last', batch_size=2, coord_scale=1, noobj_scale=1, obj_scale=5., class_scale=1.,
               nms_score_threshold=0.5, nms_max_boxes=10, nms_iou_threshold=0.5, rescore_confidence=False,
               priors=[[1., 1.], [2., 1.5]])
    prov = dict(num_train=4, num_val=0, train_generator=gen, val_generator=None)
    net = y.YOLOv2(cfg, prov)
    losses = [float(net.train_one_epoch(1e-3)) for _ in range(3)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_example_codec_round_trips_random_features():
    """encode_example -> decode_example gives the features back: byte lists (empty strings, empty lists), float32
    arrays (non-finite values by bit pattern), int64 arrays up to both ends of the range; through a record file too."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    import tempfile
    names = st.text(min_size=1, max_size=12)
    value = st.one_of(
        st.lists(st.binary(max_size=40), max_size=5),
        st.lists(st.floats(width=32, allow_nan=True, allow_infinity=True), max_size=20).map(lambda v: np.asarray(v, np.float32)),
        st.lists(st.integers(-2 ** 63, 2 ** 63 - 1), max_size=20).map(lambda v: np.asarray(v, np.int64)))

    @settings(max_examples=300, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(features=st.dictionaries(names, value, max_size=6))
    def check(features):
        record = records.encode_example(features)
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, 'one.tfrecord')
            with records.TFRecordWriter(path) as w:
                w.write(record)
                w.write(b'')
            back = list(records.read_tfrecords(path, verify=True))
        assert back == [record, b'']
        got = records.decode_example(back[0])
        assert set(got) == set(features)
        for name, v in features.items():
            if isinstance(v, list):
                assert got[name] == v
            elif v.dtype == np.float32:
                assert got[name].dtype == np.float32 and np.array_equal(got[name].view(np.uint32), v.view(np.uint32))
            else:
                assert got[name].dtype == np.int64 and np.array_equal(got[name], v)
    check()
