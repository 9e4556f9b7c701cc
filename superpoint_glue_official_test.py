#!/usr/bin/env python
"""SuperPoint (official, no BatchNorm, d=256) + SuperGlue registration test — MI355X-native drop-in for the reference
CLI superpoint_glue_official_test.py:15-137: same flags and defaults (the reference's 'supeeglue/...' typo in the
--superpoint_weights default included: its official SuperPoint ignores that flag and loads weights/superpoint_v1.pth
next to the module, superglue/models/superpoint.py:136-137), same directory convention and outputs as
superpoint_glue_test.py.  Per-pair loop: superpoint_glue_test.run_registration.

Extra flags (not in the reference): --synthetic N, --ransac gpu|host.  When a checkpoint is absent (the reference
tree ships Git-LFS pointers) deterministic synthetic weights are used and said so."""
import argparse
import os

import numpy as np
import torch

from image_matching_amd import synth
from image_matching_amd.superglue.models.matching import Matching
from superpoint_glue_test import run_registration, write_synthetic_dataset

torch.set_grad_enabled(False)


def build_parser():
    p = argparse.ArgumentParser(description='SuperPoint + SuperGlue registration test',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--exper_name', type=str, default='superpoint_glue_official', help='path to source image directory')
    p.add_argument('--img_dir', type=str, default='datasets/Camera/', help='path to source image directory')
    p.add_argument('--Result_dir', type=str, default='Results/Camera/', help='Directory where to write matching Results ')
    p.add_argument('--resize_scale', type=float, default=0.125, help='resize scale;height,weight=scale*height,scale*weight')
    p.add_argument('--match_viz', default=True, help='Whether write the match result or not')
    p.add_argument('--show_keypoints', default=True, help='Show the detected keypoints')
    p.add_argument('--descriptor_dim', type=int, default=256, help='The dimension of feature descriptor')
    p.add_argument('--superpoint_weights', type=str, default="supeeglue/models/weights/superpoint_v1.pth", help='SuperPoint official weights')
    p.add_argument('--superglue_weights', type=str, default='superglue/models/weights/superglue_indoor.pth', help='SuperGlue official weights')
    p.add_argument('--sinkhorn_iterations', type=int, default=30, help='Number of Sinkhorn iterations performed by SuperGlue')
    p.add_argument('--match_threshold', type=float, default=0.1, help='SuperGlue match threshold')
    p.add_argument('--keypoint_threshold', type=float, default=0.005, help='SuperPoint keypoint detector confidence threshold')
    p.add_argument('--nms_radius', type=int, default=4, help='SuperPoint Non Maximum Suppression (NMS) radius (Must be positive)')
    p.add_argument('--max_keypoints', type=int, default=-1, help="Maximum number of keypoints detected by Superpoint ('-1' keeps all keypoints)")
    # not in the reference
    p.add_argument('--synthetic', type=int, default=0, help='write this many synthetic pairs under --img_dir first')
    p.add_argument('--ransac', choices=['gpu', 'host'], default='gpu')
    return p


def _real(path):
    return bool(path) and os.path.exists(path) and os.path.getsize(path) > 4096


def make_config(opt):
    sp_path = opt.superpoint_weights if _real(opt.superpoint_weights) else None
    sg_path = opt.superglue_weights if _real(opt.superglue_weights) else None
    for name, path, got in (('SuperPoint', opt.superpoint_weights, sp_path), ('SuperGlue', opt.superglue_weights, sg_path)):
        if got is None:
            print(f"[imx] {name} weights {path!r} not found (or an LFS pointer): using synthetic weights")
    return {
        'superpoint': {'weights': opt.superpoint_weights, 'weights_path': sp_path, 'descriptor_dim': opt.descriptor_dim,
                       'nms_radius': opt.nms_radius, 'keypoint_threshold': opt.keypoint_threshold,
                       'max_keypoints': opt.max_keypoints},
        'superglue': {'weights': sg_path, 'descriptor_dim': opt.descriptor_dim,
                      'sinkhorn_iterations': opt.sinkhorn_iterations, 'match_threshold': opt.match_threshold},
    }


def main(argv=None):
    opt = build_parser().parse_args(argv)
    print(opt)
    if not torch.cuda.is_available():
        raise SystemExit("superpoint_glue_official_test.py (imx): needs an MI355X / ROCm GPU; there is no CPU path")
    config = make_config(opt)
    if opt.synthetic > 0:
        write_synthetic_dataset(opt.img_dir, opt.synthetic, opt.resize_scale or 1.0)
    matching = Matching(config).eval().to('cuda')
    tt = lambda sd: {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    if config['superpoint']['weights_path'] is None:
        matching.superpoint.load_state_dict(tt(synth.synth_state_dict(synth.superpoint_official_shapes(opt.descriptor_dim), 77)))
    if config['superglue']['weights'] is None and opt.descriptor_dim in synth.SG_CONFIGS:
        matching.superglue.load_state_dict(tt(synth.make_superglue_state_dict(opt.descriptor_dim)))
    return run_registration(opt, matching, 'cuda')


if __name__ == '__main__':
    main()
