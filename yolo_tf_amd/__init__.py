"""yolo_tf_amd -- MI355X-native (gfx950) YOLOv2 train + detect hot path behind the plugin surface of
ruiminshen/yolo-tf (config.ini [model]/inference plugins, Builder/Model/Objectives,
utils.postprocess.non_max_suppress, train.py / detect.py).  Compute = hand-written HIP kernels in
csrc/ reached through the C ABI of include/yolo2_hip.h; torch is device memory, streams and
torch.distributed only."""
__version__ = '0.1.0'
