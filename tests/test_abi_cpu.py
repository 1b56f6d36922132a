"""CPU: the C-ABI library loads, exports every symbol the header declares,
validates arguments on the host and refuses to compute without a HIP device."""
import ctypes
import os
import re

import numpy as np
import pytest

import yolo_tf_amd as y
from yolo_tf_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'yolohead.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(yh_[a-z_0-9]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _abi.load()
    names = header_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), 'libyolohead.so does not export %s' % n
        assert n in _abi.SIGNATURES, '_abi.py does not bind %s' % n
    assert sorted(_abi.SIGNATURES) == names
    assert lib.yh_version() >= 100


def test_config_struct_layout_matches_header():
    assert ctypes.sizeof(_abi.HeadConfig) == 4 * 3 + 4 * 4 * 4 + 4 * 16 * 2 * 4
    cfg = y.make_config([(13, 13), (26, 26)], [[[1, 2], [3, 4]], [[5, 6], [7, 8]]], [32, 16], [32, 16], 7)
    assert (cfg.num_scales, cfg.num_priors, cfg.num_classes) == (2, 2, 7)
    assert cfg.prior_hw[1][1][0] == 7.0 and cfg.grid_w[1] == 26 and cfg.box_mult[1] == 16.0
    n = ctypes.c_int64(0)
    _abi.check(_abi.load().yh_boxes_per_image(ctypes.byref(cfg), ctypes.byref(n)))
    assert n.value == (13 * 13 + 26 * 26) * 2


def test_host_side_validation_needs_no_gpu():
    lib = _abi.load()
    cfg = y.make_config([(13, 13)], [y.synth.V2_PRIORS], [32], [32], 20)
    nbytes = ctypes.c_size_t(0)
    _abi.check(lib.yh_detect_workspace_bytes(ctypes.byref(cfg), 4, 1000, 10, ctypes.byref(nbytes)))
    assert nbytes.value >= 2 * 4 * 1000 * 8 + 4 * 845 * 16
    _abi.check(lib.yh_loss_workspace_bytes(ctypes.byref(cfg), 4, 60, ctypes.byref(nbytes)))
    assert nbytes.value > 0
    bad = y.make_config([(13, 13)], [y.synth.V2_PRIORS], [32], [32], 20)
    bad.num_classes = 5000
    assert lib.yh_detect_workspace_bytes(ctypes.byref(bad), 4, 1000, 10, ctypes.byref(nbytes)) == _abi.YH_ERR_LIMIT
    assert b'num_classes' in lib.yh_last_error()
    big = y.make_config([(1024, 1024)], [y.synth.V2_PRIORS], [32], [32], 20)
    assert lib.yh_boxes_per_image(ctypes.byref(big), ctypes.byref(ctypes.c_int64(0))) == _abi.YH_ERR_LIMIT
    assert lib.yh_detect_workspace_bytes(ctypes.byref(cfg), 4, 1000, 100000, ctypes.byref(nbytes)) == _abi.YH_ERR_INVALID_ARG
    with pytest.raises(_abi.YhError):
        _abi.check(lib.yh_loss_workspace_bytes(ctypes.byref(cfg), 4, 0, ctypes.byref(nbytes)))
    # the fused loss + gradient call checks its gradient arguments before it looks for a device
    # (the addresses below are never dereferenced on the host)
    scales = (ctypes.c_float * 4)(1., 1., 5., 1.)
    heads = _abi.pointer_array([0x1000])
    call = lambda gw, grads: lib.yh_loss_with_grad(ctypes.byref(cfg), heads, 0x2000, 4, 60, scales, 0.6, 0, 0x3000, 1 << 20,
                                                   0x4000, None, None, None, gw, grads, None)
    assert call(None, _abi.pointer_array([0x5000])) == _abi.YH_ERR_INVALID_ARG
    assert call(0x6000, None) == _abi.YH_ERR_INVALID_ARG
    assert call(0x6000, heads) == _abi.YH_ERR_INVALID_ARG and b'aliases' in lib.yh_last_error()
    assert call(0x6000, _abi.pointer_array([0])) == _abi.YH_ERR_INVALID_ARG


def test_rows_wider_than_a_tile_are_refused_not_divided_by():
    """The workgroup-tile streaming kernel holds whole rows in a 4096-float LDS tile; the ABI admits
    4096 classes.  Shapes that select that kernel with wider rows must come back as YH_ERR_LIMIT from
    the host-side plan (the plan once divided by the tile's row count, which is zero there)."""
    lib = _abi.load()
    nbytes = ctypes.c_size_t(0)
    wide = y.make_config([(2, 2)], [[[1.0, 1.0]]], [32], [32], 4096)
    # dense candidate sets (capacity above the single-launch select stage) take the tile kernel
    assert lib.yh_detect_workspace_bytes(ctypes.byref(wide), 2, 5000, 10, ctypes.byref(nbytes)) == _abi.YH_ERR_LIMIT
    assert b'tile' in lib.yh_last_error()
    # ... the objectness-first kernel takes any row
    _abi.check(lib.yh_detect_workspace_bytes(ctypes.byref(wide), 2, 512, 10, ctypes.byref(nbytes)))
    edge = y.make_config([(2, 2)], [[[1.0, 1.0]]], [32], [32], 4088)         # 4093 floats per row: the widest that fits
    _abi.check(lib.yh_detect_workspace_bytes(ctypes.byref(edge), 2, 5000, 10, ctypes.byref(nbytes)))
    edge.num_classes = 4089
    assert lib.yh_detect_workspace_bytes(ctypes.byref(edge), 2, 5000, 10, ctypes.byref(nbytes)) == _abi.YH_ERR_LIMIT
    # yh_nms streams rows of C confidences through the same kernel
    for C, ok in ((4093, True), (4094, False), (4096, False)):
        rc = lib.yh_nms_workspace_bytes(2, 100, C, 1000, 10, ctypes.byref(nbytes))
        assert (rc == _abi.YH_OK) == ok and (ok or rc == _abi.YH_ERR_LIMIT)
        if not ok:
            # fake, never dereferenced addresses: the size checks come before anything touches a device
            rc = lib.yh_nms(2, 100, C, 0x1000, 0x2000, 0.5, 0.5, 10, 1000, 0x3000, 1 << 30, 0x4000, 0x5000, 0x6000, 0x7000,
                            0x8000, None, 0x9000, None)
            assert rc == _abi.YH_ERR_LIMIT and b'confidences' in lib.yh_last_error()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a HIP device is present')
    n = ctypes.c_int32(7)
    assert _abi.load().yh_device_count(ctypes.byref(n)) == _abi.YH_ERR_NO_DEVICE and n.value == 0
    hd = y.DetectionHead(20, [y.synth.V2_PRIORS], [32.0], [32.0])
    heads = [torch.zeros((1, 13, 13, 125))]
    with pytest.raises(RuntimeError, match='no CPU path'):
        hd.detect(heads, 0.5, 0.5, 10)
    with pytest.raises(RuntimeError, match='no CPU path'):
        hd.decode(heads)
    with pytest.raises(RuntimeError, match='no CPU path'):
        hd.loss(heads, torch.zeros((1, 4, 5)), 1, 1, 5, 1, 0.6, False)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'yolo-tf_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')) or f == 'Makefile':
                text = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in text.lower(), '%s mentions the oracle' % os.path.join(dirpath, f)


def test_inflight_checker_sees_a_register_touched_between_issue_and_wait():
    """tools/check_inflight.py (run by the build on the loss kernels' generated code): a copy of a register
    whose scalar load is still in flight is reported, on the straight path and on a path through a branch;
    clean code is not."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('check_inflight', os.path.join(ROOT, 'tools', 'check_inflight.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    def body(between):
        return ['\t;;#ASMSTART', '\ts_load_dwordx8 s[36:43], s[4:5], 0x0', '\ts_load_dwordx2 s[44:45], s[4:5], 0x20', '\t;;#ASMEND'] + between + [
            '\t;;#ASMSTART', '\ts_waitcnt lgkmcnt(0)', '\t;;#ASMEND', '\tv_min_f32 v1, s36, v2', '\ts_endpgm']
    n, errors = mod.check('clean', body(['\tv_add_f32 v0, v1, v2', '\ts_mov_b32 s50, s51']))
    assert n == 1 and errors == []
    n, errors = mod.check('spill', body(['\tv_writelane_b32 v40, s44, 3']))
    assert n == 1 and len(errors) == 1 and 's[44]' in errors[0].replace('s[44]', 's[44]')
    n, errors = mod.check('branch', body(['\ts_cbranch_scc1 .LBB0_2', '\tv_add_f32 v0, v1, v2', '\ts_branch .LBB0_3', '.LBB0_2:',
                                          '\ts_mov_b64 s[60:61], s[38:39]', '.LBB0_3:']))
    assert len(errors) == 1 and 's_mov_b64' in errors[0]
