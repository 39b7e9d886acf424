#!/usr/bin/env python
"""detect.py -- YOLOv2 detection on MI355X with the reference CLI (ruiminshen/yolo-tf detect.py:122-131):

    python detect.py image_or_dir -c config.ini config/yolo2/darknet-20.ini -p std -t 0.3 --threshold_iou 0.4

Forward (moving-average BN) + decode + NMS all run on the GPU (the reference's NMS is a Python
triple loop on the host, utils/postprocess.py:39-51).  Restores the latest checkpoint of
utils.get_logdir(config).  Import-safe (`std` / `darknet` preprocessors are reused by camera loops,
reference detect_camera.py:33)."""
import argparse
import configparser
import logging
import os

import numpy as np

from yolo_tf_amd import utils

PREPROCESS = {'std': 0, 'darknet': 1}


def std(image):
    from yolo_tf_amd.utils import preprocess
    return preprocess.per_image_standardization(image)


def darknet(image):
    from yolo_tf_amd.utils import preprocess
    return preprocess.darknet(image)


def read_image(path):
    """PIL load honouring the EXIF orientation (reference detect.py:41-56)."""
    from PIL import Image, ImageOps
    return ImageOps.exif_transpose(Image.open(path)).convert('RGB')


def detect(sess, names, path, args):
    """Returns [(class name, confidence, xmin, ymin, xmax, ymax in pixels), ...] for one image."""
    import torch
    model = sess.model
    width, height = sess.builder.width, sess.builder.height
    _image = read_image(path)
    image_width, image_height = _image.size
    resized = np.asarray(_image.resize((width, height)), np.uint8).astype(np.float32)
    conf, xy_min, xy_max = sess.run(torch.from_numpy(resized[None]).cuda(), PREPROCESS[args.preprocess])   # raises on NaN/Inf
    order = sess.nms(args.threshold, args.threshold_iou)[0].cpu().numpy()
    conf, xy_min, xy_max = conf[0].cpu().numpy(), xy_min[0].cpu().numpy(), xy_max[0].cpu().numpy()
    scale = np.array([image_width / model.cell_width, image_height / model.cell_height], np.float32)
    results = []
    for i in order:                                   # same traversal order as the reference's box list
        index = int(np.argmax(conf[i]))
        if conf[i, index] > args.threshold:
            (x0, y0), (x1, y1) = xy_min[i] * scale, xy_max[i] * scale
            results.append((names[index], float(conf[i, index]), float(x0), float(y0), float(x1), float(y1)))
    return results


def draw(path, results, out):
    import matplotlib
    matplotlib.use('Agg')
    import itertools
    import matplotlib.patches as patches
    import matplotlib.pyplot as plt
    fig = plt.figure()
    ax = fig.gca()
    ax.imshow(np.asarray(read_image(path)))
    colors = itertools.cycle(plt.rcParams['axes.prop_cycle'].by_key()['color'])
    for (name, c, x0, y0, x1, y1), color in zip(results, colors):
        ax.add_patch(patches.Rectangle((x0, y0), x1 - x0, y1 - y0, linewidth=min(c * 10, 3), edgecolor=color, facecolor='none'))
        ax.annotate('%s (%.1f%%)' % (name, c * 100), (x0, y0), color=color)
    ax.set_xticks([])
    ax.set_yticks([])
    fig.savefig(out)
    plt.close(fig)


def main():
    from yolo_tf_amd import checkpoint
    from yolo_tf_amd.session import DetectSession
    model = config.get('config', 'model')
    yolo = __import__('yolo_tf_amd.model.' + model, fromlist=['Builder'])
    utils.ensure_names(config)
    builder = yolo.Builder(args, config)
    builder(None)
    dtype = args.dtype or (config.get('mi355x', 'dtype') if config.has_option('mi355x', 'dtype') else 'bf16')
    sess = DetectSession(builder, 1, dtype=dtype)
    from yolo_tf_amd import tf_checkpoint
    logdir = utils.get_logdir(config)
    model_path = checkpoint.latest_checkpoint(logdir)
    tf_path = None if model_path else tf_checkpoint.latest_checkpoint(logdir)      # a logdir the reference trained (tf.train.latest_checkpoint, detect.py:104)
    if model_path is None and tf_path is None:
        raise FileNotFoundError('no checkpoint in ' + logdir)
    logging.info('load ' + (model_path or tf_path))
    step = checkpoint.restore(model_path, engine=sess.engine) if model_path else tf_checkpoint.restore(tf_path, engine=sess.engine)
    logging.info('global_step=%d' % step)
    path = os.path.expanduser(os.path.expandvars(args.path))
    paths = [path] if os.path.isfile(path) else [os.path.join(d, f) for d, _, fs in os.walk(path) for f in fs
                                                 if os.path.splitext(f)[-1].lower() in args.exts]
    for _path in paths:
        results = detect(sess, builder.names, _path, args)
        print('%s: %d objects detected' % (_path, len(results)))
        for r in results:
            print('  %s %.1f%% (%.1f, %.1f)-(%.1f, %.1f)' % r)
        if args.output:
            os.makedirs(args.output, exist_ok=True)
            draw(_path, results, os.path.join(args.output, os.path.basename(_path) + '.png'))


def make_args():
    parser = argparse.ArgumentParser()
    parser.add_argument('path', help='input image path')
    parser.add_argument('-c', '--config', nargs='+', default=['config.ini'], help='config file')
    parser.add_argument('-p', '--preprocess', default='std', choices=sorted(PREPROCESS), help='the preprocess function')
    parser.add_argument('-t', '--threshold', type=float, default=0.3)
    parser.add_argument('--threshold_iou', type=float, default=0.4, help='IoU threshold')
    parser.add_argument('-e', '--exts', nargs='+', default=['.jpg', '.png'])
    parser.add_argument('--level', default='info', help='logging level')
    parser.add_argument('--output', help='directory for annotated images (the reference opens a matplotlib window instead)')
    parser.add_argument('--dtype', default=None, choices=['bf16', 'f32'])
    return parser.parse_args()


if __name__ == '__main__':
    args = make_args()
    config = configparser.ConfigParser()
    utils.load_config(config, args.config)
    logging.basicConfig()
    if args.level:
        logging.getLogger().setLevel(args.level.upper())
    main()
