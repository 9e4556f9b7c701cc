#!/usr/bin/env python
"""SuperPoint + nearest-neighbour matcher registration test — MI355X-native drop-in for the reference CLI
(superpoint_flann_test.py:13-127 of PH8411/image-matching): same flags, directory convention
(<img_dir>/source1/*, <img_dir>/template1/<one image>) and outputs (<Result_dir>/transformed/trans_<file>,
<Result_dir>/Match/match_<file>).

The reference hands the descriptors to cv2.FlannBasedMatcher (KD-tree, approximate) on the host; here the
2-nearest-neighbour search, the 0.7 ratio test and the RANSAC partial-affine fit all run in libimx on the GPU
(imx_knn_ratio_match: exact search, so it returns what FLANN approximates; imx_estimate_affine_partial).

Extra flags (not in the reference): --synthetic N writes a synthetic dataset first; --ransac gpu|host."""
import argparse
import os

import numpy as np
import torch

from image_matching_amd import hostops, synth
from image_matching_amd.superpoint.models.superpoint_test import SuperPoint
import superpoint_glue_test as glue_cli
from superpoint_glue_test import load_pair, write_synthetic_dataset

MIN_MATCH_COUNT = 4
RATIO = 0.7

torch.set_grad_enabled(False)


def build_parser():
    p = argparse.ArgumentParser(description='SuperPoint_flann test', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--img_dir', type=str, default='datasets/Amazon/', help='path to source image directory')
    p.add_argument('--Result_dir', type=str, default='Results/Camera/superpoint_allss_descriptor_128', help='Directory where to write matching Results ')
    p.add_argument('--resize_scale', type=float, default=0.25, help='resize scale;height,weight=scale*height,scale*weight')
    p.add_argument('--match_viz', default=True, help='Whether write the match result or not')
    p.add_argument('--weights_path', type=str, default='superpoint/models/weights/superPointNet_allss_descriptor_128.pth.tar', help='pretrain model path')
    p.add_argument('--descriptor_dim', type=int, default=128, help='the dimension of descriptor')
    p.add_argument('--max_keypoints', type=int, default=1200, help="Maximum number of keypoints detected by Superpoint ('-1' keeps all keypoints)")
    p.add_argument('--keypoint_threshold', type=float, default=0.005, help='SuperPoint keypoint detector confidence threshold')
    p.add_argument('--nms_radius', type=int, default=4, help='SuperPoint Non Maximum Suppression (NMS) radius (Must be positive)')
    # not in the reference
    p.add_argument('--synthetic', type=int, default=0, help='write this many synthetic pairs under --img_dir first')
    p.add_argument('--ransac', choices=['gpu', 'host'], default='gpu')
    return p


def match_colors(dist, d):
    """superpoint_flann_test.py:100-110 + utils/utils.py:225-230: distance -> score in [0,1] -> gist_rainbow(0.4 s)."""
    worst = d * 2 if dist.max() > 1 else 1
    s = 1 - np.clip(dist / worst, 0, 1)
    import matplotlib.cm as cm
    return np.array(cm.gist_rainbow(s * 0.4))


def main(argv=None):
    opt = build_parser().parse_args(argv)
    print(opt)
    if not torch.cuda.is_available():
        raise SystemExit("superpoint_flann_test.py (imx): needs an MI355X / ROCm GPU; there is no CPU path")
    device = 'cuda'
    weights = opt.weights_path if opt.weights_path and os.path.exists(opt.weights_path) \
        and os.path.getsize(opt.weights_path) > 4096 else None
    if weights is None:
        print(f"[imx] weights file {opt.weights_path!r} not found (or an LFS pointer): using synthetic weights")
    config = {'superpoint': {'weights': weights, 'descriptor_dim': opt.descriptor_dim, 'nms_radius': opt.nms_radius,
                             'keypoint_threshold': opt.keypoint_threshold, 'max_keypoints': opt.max_keypoints}}
    if opt.synthetic > 0:
        write_synthetic_dataset(opt.img_dir, opt.synthetic, opt.resize_scale or 1.0)
    source_dir = opt.img_dir + 'source1/'
    template_dir = opt.img_dir + 'template1/'
    template_img_path = template_dir + os.listdir(template_dir)[0]

    superpoint = SuperPoint(config.get('superpoint', {})).to(device).eval()
    if weights is None and opt.descriptor_dim in (128, 256):
        superpoint.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in
                                    synth.make_superpoint_state_dict(opt.descriptor_dim).items()})

    results = []
    for filename in sorted(os.listdir(source_dir)):
        source_original, source_image, template_image = load_pair(source_dir + filename, template_img_path, opt.resize_scale)
        image1_tensor = torch.from_numpy(source_image)[None].float().to(device)
        image2_tensor = torch.from_numpy(template_image)[None].float().to(device)
        pred1 = superpoint(image1_tensor)
        pred2 = superpoint(image2_tensor)
        eng = superpoint._shared.engine
        kp1, kp2 = pred1['keypoints'][0], pred2['keypoints'][0]
        d1, d2 = pred1['descriptors'][0], pred2['descriptors'][0]            # (d, K) views
        n_good, Matrix = 0, None
        if kp1.shape[0] and kp2.shape[0] >= 2:
            m, dist1, _ = eng.knn_ratio_match(d1[None], d2[None], ratio=RATIO)
            n_good = int((m >= 0).sum())
        if n_good > MIN_MATCH_COUNT:
            m_h, dist_h = m[0].cpu().numpy(), dist1[0].cpu().numpy()
            KeyP1, KeyP2 = kp1.cpu().numpy(), kp2.cpu().numpy()
            good = m_h >= 0
            src_pts, dst_pts, match_dist = KeyP1[good], KeyP2[m_h[good]], dist_h[good]
            Matrix, mask = None, None
            if opt.ransac == 'gpu':     # matched pairs compacted on the device; (None, None) when the GPU path does not apply
                Matrix, mask = glue_cli.gpu_affine_partial(eng, {'matches0': m, 'keypoints0': [kp1], 'keypoints1': [kp2]}, 7)
            if mask is None:
                Matrix, mask = hostops.estimate_affine_partial_2d(src_pts, dst_pts, ransac_thresh=7)
            ok = Matrix is not None
            RansacMask = (np.asarray(mask) == 1).ravel() if ok else np.zeros(len(src_pts), bool)
            if not ok:                  # the reference would raise on a None matrix: skip the pair
                print(f"[imx] {filename}: RANSAC found no model, skipping")
                results.append((filename, n_good, 0, None))
                continue
            Matrix = np.array(Matrix, dtype=np.float64)
            if opt.resize_scale is not None:
                Matrix[:, 2] = Matrix[:, 2] / opt.resize_scale
            src255 = source_original.squeeze() * 255
            if opt.ransac == 'gpu':
                Transform = eng.warp_affine_u8(torch.from_numpy(np.rint(src255).astype(np.uint8)), Matrix).cpu().numpy()
            else:
                Transform = hostops.warp_affine(src255, Matrix, (src255.shape[1], src255.shape[0]))
            Transform_dir = os.path.join(opt.Result_dir, 'transformed/')
            os.makedirs(Transform_dir, exist_ok=True)
            hostops.imwrite(Transform_dir + 'trans_{}'.format(filename), Transform)
            src_r, dst_r, dist_r = src_pts[RansacMask], dst_pts[RansacMask], match_dist[RansacMask]
            if opt.match_viz and len(src_r):
                Match_dir = os.path.join(opt.Result_dir, 'Match/')
                os.makedirs(Match_dir, exist_ok=True)
                hostops.make_matching_plot_fast(source_image.squeeze() * 255, template_image.squeeze() * 255, KeyP1, KeyP2,
                                                src_r, dst_r, match_colors(dist_r, d1.shape[0]), [],
                                                path=Match_dir + 'match_{}'.format(filename), margin=0)
            results.append((filename, n_good, int(RansacMask.sum()), Matrix))
            print(f"{filename}: keypoints {len(KeyP1)}:{len(KeyP2)} ratio-test matches {n_good} inliers {int(RansacMask.sum())}")
        else:
            results.append((filename, n_good, 0, None))
            print(f"{filename}: only {n_good} ratio-test matches (need > {MIN_MATCH_COUNT})")
    return results


if __name__ == '__main__':
    main()
