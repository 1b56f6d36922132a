"""ctypes binding of libyolohead.so (declarations: include/yolohead.h).

There is no CPU fallback: if the shared library is missing or no HIP device is
present, every compute entry point raises."""
import ctypes
import os

YH_MAX_SCALES = 4
YH_MAX_PRIORS = 16
YH_MAX_CLASSES = 4096
YH_MAX_BOXES_PER_IMAGE = 1 << 20
YH_MAX_NMS_BOXES = 512
YH_STATUS_CANDIDATE_OVERFLOW = 1
YH_STATUS_COUNT_SHIFT = 8
YH_STATIC_SCRATCH_BYTES = 8 << 20

YH_OK = 0
YH_ERR_INVALID_ARG = -1
YH_ERR_NO_DEVICE = -2
YH_ERR_HIP = -3
YH_ERR_WORKSPACE = -4
YH_ERR_LIMIT = -5
YH_ERR_ALIGN = -6

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libyolohead.so')


class YhError(RuntimeError):
    def __init__(self, code, message):
        super().__init__('libyolohead error %d: %s' % (code, message))
        self.code = code


class HeadConfig(ctypes.Structure):
    """struct yh_head_config."""
    _fields_ = [
        ('num_scales', ctypes.c_int32),
        ('num_priors', ctypes.c_int32),
        ('num_classes', ctypes.c_int32),
        ('grid_h', ctypes.c_int32 * YH_MAX_SCALES),
        ('grid_w', ctypes.c_int32 * YH_MAX_SCALES),
        ('box_mult', ctypes.c_float * YH_MAX_SCALES),
        ('gt_rate', ctypes.c_float * YH_MAX_SCALES),
        ('prior_hw', ((ctypes.c_float * 2) * YH_MAX_PRIORS) * YH_MAX_SCALES),
    ]


class Tuning(ctypes.Structure):
    """struct yh_tuning; 0 = the built-in default."""
    _fields_ = [(n, ctypes.c_int32) for n in ('detect_kernel', 'row_steps', 'stream_tile_kib', 'stream_buffers',
                                              'stream_units', 'stream_blocks', 'stream_resident', 'rows_dense_min', 'rows_probe')]


_P = ctypes.c_void_p
_I = ctypes.c_int32
_F = ctypes.c_float
_CFG = ctypes.POINTER(HeadConfig)
_PP = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); every symbol include/yolohead.h declares
SIGNATURES = {
    'yh_version': (ctypes.c_int, []),
    'yh_last_error': (ctypes.c_char_p, []),
    'yh_set_tuning': (ctypes.c_int, [ctypes.POINTER(Tuning)]),
    'yh_get_tuning': (ctypes.c_int, [ctypes.POINTER(Tuning)]),
    'yh_device_count': (ctypes.c_int, [ctypes.POINTER(_I)]),
    'yh_probe_read': (ctypes.c_int, [_P, ctypes.c_size_t, _P]),
    'yh_boxes_per_image': (ctypes.c_int, [_CFG, ctypes.POINTER(ctypes.c_int64)]),
    'yh_decode': (ctypes.c_int, [_CFG, _PP, _I, _P, _P, _P]),
    'yh_detect_workspace_bytes': (ctypes.c_int, [_CFG, _I, _I, _I, ctypes.POINTER(ctypes.c_size_t)]),
    'yh_detect': (ctypes.c_int, [_CFG, _PP, _I, _F, _F, _I, _I, _P, ctypes.c_size_t,
                                 _P, _P, _P, _P, _P, _P, _P, _P]),
    'yh_detect_static': (ctypes.c_int, [_CFG, _PP, _I, _F, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    'yh_detect_timed': (ctypes.c_int, [_CFG, _PP, _I, _F, _F, _I, _I, _P, ctypes.c_size_t,
                                       _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'yh_detect_pipelined': (ctypes.c_int, [_CFG, _PP, _I, _F, _F, _I, _I, _P, _P, ctypes.c_size_t,
                                           _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'yh_detect_candidates': (ctypes.c_int, [_CFG, _PP, _I, _F, _I, _I, _P, ctypes.c_size_t, _P]),
    'yh_detect_select': (ctypes.c_int, [_CFG, _I, _F, _I, _I, _P, ctypes.c_size_t,
                                        _P, _P, _P, _P, _P, _P, _P, _P]),
    'yh_nms_workspace_bytes': (ctypes.c_int, [_I, ctypes.c_int64, _I, _I, _I, ctypes.POINTER(ctypes.c_size_t)]),
    'yh_nms': (ctypes.c_int, [_I, ctypes.c_int64, _I, _P, _P, _F, _F, _I, _I, _P, ctypes.c_size_t,
                              _P, _P, _P, _P, _P, _P, _P, _P]),
    'yh_nms_timed': (ctypes.c_int, [_I, ctypes.c_int64, _I, _P, _P, _F, _F, _I, _I, _P, ctypes.c_size_t,
                                    _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'yh_loss_workspace_bytes': (ctypes.c_int, [_CFG, _I, _I, ctypes.POINTER(ctypes.c_size_t)]),
    'yh_loss_prepare': (ctypes.c_int, [_CFG, _P, _I, _I, _P, ctypes.c_size_t, _P]),
    'yh_loss_forward': (ctypes.c_int, [_CFG, _PP, _P, _I, _I, ctypes.POINTER(_F), _F, _I, _P, ctypes.c_size_t,
                                       _P, _P, _P, _P, _P, _P]),
    'yh_loss_backward': (ctypes.c_int, [_CFG, _PP, _P, _I, _I, ctypes.POINTER(_F), _I, _P, _P, _PP, _P]),
    'yh_loss_with_grad': (ctypes.c_int, [_CFG, _PP, _P, _I, _I, ctypes.POINTER(_F), _F, _I, _P, ctypes.c_size_t,
                                         _P, _P, _P, _P, _P, _PP, _P]),
    'yh_loss': (ctypes.c_int, [_CFG, _PP, _P, _I, _I, ctypes.POINTER(_F), _F, _I, _P, ctypes.c_size_t,
                               _P, _P, _P, _P, _P]),
}

_lib = None
tuning_epoch = 0


def load():
    """Loads libyolohead.so once; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError('%s not found: build it with `make -C yolo-tf_amd/csrc` '
                              '(or __graft_entry__.build()); there is no fallback path' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def set_tuning(**fields):
    """yh_set_tuning: process-wide launch-plan knobs (no arguments = defaults)."""
    global tuning_epoch
    t = Tuning(**{k: int(v) for k, v in fields.items()})
    check(load().yh_set_tuning(ctypes.byref(t)))
    tuning_epoch += 1                     # cached launch plans and workspace sizes are stale now


def get_tuning():
    t = Tuning()
    check(load().yh_get_tuning(ctypes.byref(t)))
    return {n: getattr(t, n) for n, _ in Tuning._fields_}


def check(rc):
    if rc != YH_OK:
        raise YhError(rc, load().yh_last_error().decode('utf-8', 'replace'))


def pointer_array(ptrs):
    arr = (ctypes.c_void_p * len(ptrs))(*ptrs)
    return arr
