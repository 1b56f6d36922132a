"""The data formats on either side of the head path.

Before the loss: the ground-truth wire format the reference's input pipeline
hands to `train_one_epoch` (utils/image_augmentor.py:205-244): per image a
[pad_truth_to, 5] float32 block of (y_centre, x_centre, h, w, class_id) rows in
pixels of the network input, padded with -1 rows; `_get_normlized_gn`
(YOLOv3.py:435-442) finds the first pad row by argmin of column 0, so at least
one pad row must remain.

After NMS: `test_one_image` returns [scores, bbox (y1, x1, y2, x2), class_id]
(YOLOv3.py:365-368); the reference's consumer maps class ids to names
(utils/voc_classname_encoder.py:1-22) and draws the boxes (testYOLOv3.py:81-97).
"""
import numpy as np

# PASCAL VOC class order of utils/voc_classname_encoder.py:1-22
VOC_CLASSNAMES = ('aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow',
                  'diningtable', 'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa',
                  'train', 'tvmonitor')
classname_to_ids = {name: i for i, name in enumerate(VOC_CLASSNAMES)}
ids_to_classname = {i: name for i, name in enumerate(VOC_CLASSNAMES)}


def encode_ground_truth(corner_boxes, class_ids, output_hw, pad_truth_to):
    """Last stage of the reference's augmentor for one image
    (utils/image_augmentor.py:205-244): boxes (ymin, xmin, ymax, xmax) in pixels of
    the output image -> [pad_truth_to, 5] rows (yc, xc, h, w, class), -1 padded.
    A box whose centre is not strictly inside (0, size - 1) is dropped; the others
    are clamped to [0, size - 1] before centre and size are taken.  When no box
    survives the reference falls back to the un-augmented sample; here the caller
    gets an all -1 block and decides (the loss of such an image is NaN in the
    reference as well: division by a GT count of zero)."""
    boxes = np.asarray(corner_boxes, dtype=np.float32).reshape(-1, 4)
    cls = np.asarray(class_ids, dtype=np.float32).reshape(-1)
    if boxes.shape[0] != cls.shape[0]:
        raise ValueError('%d boxes but %d class ids' % (boxes.shape[0], cls.shape[0]))
    h_out, w_out = float(output_hw[0]), float(output_hw[1])
    ymin, xmin, ymax, xmax = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    yc = (ymin + ymax) / np.float32(2.)
    xc = (xmin + xmax) / np.float32(2.)
    keep = (yc > 0.) & (yc < h_out - 1.) & (xc > 0.) & (xc < w_out - 1.)
    ymin, xmin, ymax, xmax, cls = ymin[keep], xmin[keep], ymax[keep], xmax[keep], cls[keep]
    ymin = np.clip(ymin, 0., h_out - 1.); ymax = np.clip(ymax, 0., h_out - 1.)
    xmin = np.clip(xmin, 0., w_out - 1.); xmax = np.clip(xmax, 0., w_out - 1.)
    rows = np.stack([(ymin + ymax) / np.float32(2.), (xmin + xmax) / np.float32(2.), ymax - ymin, xmax - xmin, cls],
                    axis=-1).astype(np.float32)
    if rows.shape[0] >= int(pad_truth_to):
        raise ValueError('%d boxes do not leave a pad row in pad_truth_to=%d' % (rows.shape[0], pad_truth_to))
    out = np.full((int(pad_truth_to), 5), -1.0, dtype=np.float32)
    out[:rows.shape[0]] = rows
    return out


def encode_ground_truth_batch(corner_boxes_list, class_ids_list, output_hw, pad_truth_to):
    """[B, pad_truth_to, 5] for a list of images (what the dataset's padded_batch yields)."""
    return np.stack([encode_ground_truth(b, c, output_hw, pad_truth_to)
                     for b, c in zip(corner_boxes_list, class_ids_list)], axis=0)


def decode_ground_truth(block):
    """Inverse view of one image's block: (corner boxes [G, 4], class ids [G]) of the
    rows before the first pad row (the slice `_get_normlized_gn` takes)."""
    block = np.asarray(block, dtype=np.float32)
    g = int(np.argmin(block, axis=0)[0])
    rows = block[:g]
    half_h, half_w = rows[:, 2] / np.float32(2.), rows[:, 3] / np.float32(2.)
    boxes = np.stack([rows[:, 0] - half_h, rows[:, 1] - half_w, rows[:, 0] + half_h, rows[:, 1] + half_w], axis=-1)
    return boxes.astype(np.float32), rows[:, 4].astype(np.int32)


def draw_detections(image, scores, bbox, class_id, names=ids_to_classname, color=(255, 0, 0)):
    """The reference's (commented-out) plotting snippet, testYOLOv3.py:81-97: a rectangle
    per detection from (x1, y1) to (x2, y2) and the caption 'name score' at its top-left
    corner.  image: HWC array, 0..255.  Returns a PIL image."""
    from PIL import Image, ImageDraw
    canvas = Image.fromarray(np.clip(np.asarray(image), 0, 255).astype(np.uint8))
    draw = ImageDraw.Draw(canvas)
    for rec in detections_to_records(scores, bbox, class_id, names):
        draw.rectangle([rec['x1'], rec['y1'], max(rec['x1'], rec['x2']), max(rec['y1'], rec['y2'])], outline=color)
        draw.text((rec['x1'], rec['y1']), '%s %.3f' % (rec['name'], rec['score']), fill=color)
    return canvas


def detections_to_records(scores, bbox, class_id, names=ids_to_classname):
    """test_one_image's result as a list of dicts (name, score, y1, x1, y2, x2), the
    values the reference's plotting snippet reads (testYOLOv3.py:81-97)."""
    out = []
    for s, b, c in zip(np.asarray(scores), np.asarray(bbox), np.asarray(class_id)):
        name = names.get(int(c), str(int(c))) if hasattr(names, 'get') else names[int(c)]
        out.append(dict(name=name, class_id=int(c), score=float(s),
                        y1=float(b[0]), x1=float(b[1]), y2=float(b[2]), x2=float(b[3])))
    return out
