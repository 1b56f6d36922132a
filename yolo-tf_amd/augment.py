"""image_augmentor of the reference's input pipeline (utils/image_augmentor.py:7-246)
on NumPy arrays: resize (with or without the aspect ratio kept), crop, flips,
colour jitter, small rotations, and the ground-truth rows carried through every
step into the [pad_truth_to, 5] block of (y_centre, x_centre, h, w, class_id)
the loss reads.

Random draws come from a numpy Generator (TensorFlow's random stream cannot be
reproduced); everything else follows the reference step by step, including:
  * ratios are float32 casts of a float64 quotient, the resized side is the
    truncation of float32(side) * ratio (image_augmentor.py:90-99);
  * padding after a ratio-keeping resize goes to the bottom and right only;
  * a flipped coordinate is size - c - 1 (image_augmentor.py:157-166);
  * `if tf.shape(ground_truth_)[0] == 0` (image_augmentor.py:230) compares a
    tensor with an int in graph mode and is never true, so the branch that
    falls back to the un-augmented sample is dead: an image whose boxes were all
    dropped leaves an all -1 block (and a NaN loss, as in the reference).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .wire import encode_ground_truth

_RESIZE_MODES = {'NEAREST_NEIGHBOR': 'nearest', 'BILINEAR': 'bilinear', 'BICUBIC': 'bicubic'}


def resize_image(image, out_h, out_w, fill_mode):
    """tf.image.resize_images(..., align_corners=True) of an HWC float image."""
    h, w = image.shape[0], image.shape[1]
    out_h, out_w = int(out_h), int(out_w)
    if fill_mode == 'NEAREST_NEIGHBOR':
        def index(n_in, n_out):
            scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
            return np.minimum(np.floor(np.arange(n_out) * np.float32(scale) + np.float32(0.5)).astype(np.int64), n_in - 1)
        return image[index(h, out_h)][:, index(w, out_w)]
    x = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).permute(2, 0, 1)[None]
    y = F.interpolate(x, size=(out_h, out_w), mode=_RESIZE_MODES[fill_mode], align_corners=True)
    return y[0].permute(1, 2, 0).contiguous().numpy()


def rotate_image(image, angle):
    """tf.contrib.image.rotate(image, angle, 'BILINEAR'): counter-clockwise about the
    centre, zeros outside.  Output pixel (x, y) samples the input at
    (cos*x - sin*y + x_off, sin*x + cos*y + y_off)."""
    h, w = image.shape[0], image.shape[1]
    c, s = np.float32(np.cos(angle)), np.float32(np.sin(angle))
    x_off = ((w - 1) - (c * (w - 1) - s * (h - 1))) / np.float32(2.)
    y_off = ((h - 1) - (s * (w - 1) + c * (h - 1))) / np.float32(2.)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
    sx = c * xs - s * ys + x_off
    sy = s * xs + c * ys + y_off
    x0, y0 = np.floor(sx), np.floor(sy)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return np.where(ok[..., None], image[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], np.float32(0.))
    top = tap(y0, x0) * (1 - fx) + tap(y0, x0 + 1) * fx
    bottom = tap(y0 + 1, x0) * (1 - fx) + tap(y0 + 1, x0 + 1) * fx
    return (top * (1 - fy) + bottom * fy).astype(np.float32)


def adjust_hue(image, delta):
    """tf.image.adjust_hue on a float RGB image: hue + delta (mod 1) in HSV."""
    r, g, b = image[..., 0], image[..., 1], image[..., 2]
    v = image.max(axis=-1)
    rng = v - image.min(axis=-1)
    safe = np.where(rng > 0, rng, 1)
    hue = np.where(v == r, (g - b) / safe, np.where(v == g, 2. + (b - r) / safe, 4. + (r - g) / safe))
    hue = np.where(rng > 0, (hue / 6.) % 1., 0.)
    sat = np.where(v > 0, rng / np.where(v > 0, v, 1), 0.)
    hue = (hue + delta) % 1.
    k = (np.stack([5., 3., 1.]) + hue[..., None] * 6.) % 6.
    return (v[..., None] - (v * sat)[..., None] * np.clip(np.minimum(k, 4. - k), 0., 1.)).astype(np.float32)


def _check(data_format, fill_mode, zoom_size, output_shape, keep_aspect_ratios, constant_values, crop_method,
           color_jitter_prob, flip_prob, rotate, ground_truth):
    """image_augmentor.py:29-60 (the conditions as the reference evaluates them)."""
    if data_format not in ['channels_first', 'channels_last']:
        raise Exception("data_format must in ['channels_first', 'channels_last']!")
    if fill_mode not in ['CONSTANT', 'NEAREST_NEIGHBOR', 'BILINEAR', 'BICUBIC']:
        raise Exception("fill_mode must in ['CONSTANT', 'NEAREST_NEIGHBOR', 'BILINEAR', 'BICUBIC']!")
    if fill_mode == 'CONSTANT' and zoom_size is not None:
        raise Exception("if fill_mode is 'CONSTANT', zoom_size can't be None!")
    if zoom_size is not None:
        if keep_aspect_ratios and constant_values is None:
            raise Exception('please provide constant_values!')
        if not zoom_size[0] >= output_shape[0] and zoom_size[1] >= output_shape[1]:
            raise Exception("output_shape can't greater that zoom_size!")
        if crop_method not in ['random', 'center']:
            raise Exception("crop_method must in ['random', 'center']!")
    if color_jitter_prob is not None and not 0. <= color_jitter_prob <= 1.:
        raise Exception("color_jitter_prob can't less that 0.0, and can't grater that 1.0")
    if flip_prob is not None:
        if not 0. <= flip_prob[0] <= 1. and 0. <= flip_prob[1] <= 1.:
            raise Exception("flip_prob can't less than 0.0, and can't grater than 1.0")
    if rotate is not None:
        if len(rotate) != 3:
            raise Exception('please provide "rotate" parameter as [rotate_prob, min_angle, max_angle]!')
        if not 0. <= rotate[0] <= 1.:
            raise Exception("rotate prob can't less that 0.0, and can't grater that 1.0")
        if ground_truth is not None:
            if not -5. <= rotate[1] <= 5. and -5. <= rotate[2] <= 5.:
                raise Exception('rotate range must be -5 to 5, otherwise coordinate mapping become imprecise!')
        if not rotate[1] <= rotate[2]:
            raise Exception("rotate[1] can't  grater than rotate[2]")


def image_augmentor(image, input_shape, data_format, output_shape, zoom_size=None,
                    crop_method=None, flip_prob=None, fill_mode='BILINEAR', keep_aspect_ratios=False,
                    constant_values=0., color_jitter_prob=None, rotate=None, ground_truth=None, pad_truth_to=None,
                    rng=None):
    """Arguments as utils/image_augmentor.py:11-28, plus rng (numpy Generator).
    ground_truth rows are (ymin, ymax, xmin, xmax, class_id).  Returns image, or
    (image, ground_truth [pad_truth_to, 5]) when ground_truth is given."""
    _check(data_format, fill_mode, zoom_size, output_shape, keep_aspect_ratios, constant_values, crop_method,
           color_jitter_prob, flip_prob, rotate, ground_truth)
    rng = rng if rng is not None else np.random.default_rng()
    f32 = np.float32
    if fill_mode == 'CONSTANT':
        keep_aspect_ratios = True
    if ground_truth is not None:
        ground_truth = np.asarray(ground_truth, dtype=f32).reshape(-1, 5)
        ymin, ymax = ground_truth[:, 0].copy(), ground_truth[:, 1].copy()
        xmin, xmax = ground_truth[:, 2].copy(), ground_truth[:, 3].copy()
        class_id = ground_truth[:, 4].copy()
    image = np.asarray(image, dtype=f32)
    if data_format == 'channels_first':
        image = image.transpose(1, 2, 0)
    input_h, input_w, input_c = int(input_shape[0]), int(input_shape[1]), int(input_shape[2])
    output_h, output_w = int(output_shape[0]), int(output_shape[1])
    zoom_h, zoom_w = (int(zoom_size[0]), int(zoom_size[1])) if zoom_size is not None else (output_h, output_w)

    if keep_aspect_ratios:
        if fill_mode in _RESIZE_MODES:
            by_height = zoom_h / input_h < zoom_w / input_w
            zoom_ratio = f32(zoom_h / input_h) if by_height else f32(zoom_w / input_w)
            if by_height:
                resize_h, resize_w = zoom_h, int(f32(input_w) * zoom_ratio)
            else:
                resize_h, resize_w = int(f32(input_h) * zoom_ratio), zoom_w
            image = resize_image(image, resize_h, resize_w, fill_mode)
            if ground_truth is not None:
                ymin, ymax, xmin, xmax = ymin * zoom_ratio, ymax * zoom_ratio, xmin * zoom_ratio, xmax * zoom_ratio
        else:
            resize_h, resize_w = input_h, input_w
        image = np.pad(image, [[0, zoom_h - resize_h], [0, zoom_w - resize_w], [0, 0]], mode='constant',
                       constant_values=constant_values)
    else:
        image = resize_image(image, zoom_h, zoom_w, fill_mode)
        if ground_truth is not None:
            ry, rx = f32(zoom_h / input_h), f32(zoom_w / input_w)
            ymin, ymax, xmin, xmax = ymin * ry, ymax * ry, xmin * rx, xmax * rx

    if zoom_size is not None:
        if crop_method == 'random':
            crop_h = int(rng.integers(0, zoom_h - output_h)) if zoom_h > output_h else 0
            crop_w = int(rng.integers(0, zoom_w - output_w)) if zoom_w > output_w else 0
        else:
            crop_h, crop_w = (zoom_h - output_h) // 2, (zoom_w - output_w) // 2
        image = image[crop_h:crop_h + output_h, crop_w:crop_w + output_w, :input_c]
        if ground_truth is not None:
            ymin, ymax, xmin, xmax = ymin - f32(crop_h), ymax - f32(crop_h), xmin - f32(crop_w), xmax - f32(crop_w)

    if flip_prob is not None:
        flip_td, flip_lr = rng.uniform(0., 1.) < flip_prob[0], rng.uniform(0., 1.) < flip_prob[1]
        if flip_td:
            image = image[::-1]
        if flip_lr:
            image = image[:, ::-1]
        if ground_truth is not None:
            if flip_td:
                ymax, ymin = f32(output_h) - ymin - f32(1.), f32(output_h) - ymax - f32(1.)
            if flip_lr:
                xmax, xmin = f32(output_w) - xmin - f32(1.), f32(output_w) - xmax - f32(1.)

    if color_jitter_prob is not None:
        bcs = rng.uniform(0., 1., size=3)
        if bcs[0] < color_jitter_prob:                         # adjust_brightness: + delta
            image = image + f32(rng.uniform(0., 0.3))
        if bcs[1] < color_jitter_prob:                         # adjust_contrast: about the channel means
            mean = image.mean(axis=(0, 1), keepdims=True, dtype=np.float32)
            image = (image - mean) * f32(rng.uniform(0.8, 1.2)) + mean
        if bcs[2] < color_jitter_prob:
            image = adjust_hue(image, f32(rng.uniform(-0.1, 0.1)))

    if rotate is not None:
        angles = f32(rng.uniform(rotate[1], rotate[2])) * f32(3.1415926) / f32(180.)
        image = rotate_image(np.ascontiguousarray(image), angles)
        if ground_truth is not None:
            a = -angles
            ca, sa = f32(np.cos(a)), f32(np.sin(a))
            cx, cy = f32((output_w - 1.) / 2.), f32((output_h - 1.) / 2.)
            off_x = cx * (1 - ca) + cy * sa
            off_y = cy * (1 - ca) - cx * sa
            px = np.stack([x * ca - y * sa + off_x for x, y in ((xmin, ymin), (xmax, ymax), (xmin, ymax), (xmax, ymin))], -1)
            py = np.stack([x * sa + y * ca + off_y for x, y in ((xmin, ymin), (xmax, ymax), (xmin, ymax), (xmax, ymin))], -1)
            xmin, xmax, ymin, ymax = px.min(-1), px.max(-1), py.min(-1), py.max(-1)

    image = np.ascontiguousarray(image, dtype=f32)
    if data_format == 'channels_first':
        image = np.ascontiguousarray(image.transpose(2, 0, 1))
    if ground_truth is None:
        return image
    corners = np.stack([ymin, xmin, ymax, xmax], axis=-1)
    if pad_truth_to is None:
        block = encode_ground_truth(corners, class_id, (output_h, output_w), len(class_id) + 1)
        return image, block[:int(np.argmin(block[:, 0]))] if len(class_id) else block[:0]
    return image, encode_ground_truth(corners, class_id, (output_h, output_w), pad_truth_to)
