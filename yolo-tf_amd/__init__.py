"""MI355X-native detection-head path of YOLO-TF (decode, IoU ignore mask and
target assignment of the loss, per-class NMS) behind the reference's YOLOv2 /
YOLOv3 class API.  Compute runs in hand-written HIP kernels (csrc/) reached
through a ctypes C ABI (include/yolohead.h); there is no CPU fallback."""
from . import _abi, backbone, evaluate, parallel, synth, wire
from .head import DetectionHead, DetectLanes, DetectPipeline, Detections, make_config
from .yolov2 import YOLOv2
from .yolov3 import YOLOv3

__all__ = ['DetectionHead', 'DetectPipeline', 'DetectLanes', 'Detections', 'make_config', 'YOLOv2', 'YOLOv3', 'parallel', 'synth', 'wire', 'backbone', '_abi']
