#!/usr/bin/env python
"""SuperPoint + SuperGlue registration test — MI355X-native drop-in for the reference CLI
(superpoint_glue_test.py:15-140 of PH8411/image-matching): same flags, same directory convention
(<img_dir>/source1/*, <img_dir>/template1/<one image>), same outputs
(<Result_dir>/<exper_name>/Transform/trans_<file>, .../Match/<file>) and the same per-pair
"Time used:" print.  The matching forward runs in libimx (HIP kernels); image I/O and the RANSAC
partial-affine post-step are host plumbing (OpenCV when present, numpy/PIL otherwise).

Extra flag (not in the reference): --synthetic N writes N synthetic 640x480 source images and a
template under --img_dir first, so the pipeline can be exercised without a dataset."""
import argparse
import ast
import os
import time
from pathlib import Path

import numpy as np
import torch

from image_matching_amd import hostops, synth
from image_matching_amd.superglue.models.matching_test import Matching

torch.set_grad_enabled(False)


def build_parser():
    p = argparse.ArgumentParser(description='SuperPoint + SuperGlue registration test',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--exper_name', type=str, default='superpoint_glue_descriptor', help='path to source image directory')
    p.add_argument('--img_dir', type=str, default='datasets/Amazon/', help='path to source image directory')
    p.add_argument('--Result_dir', type=str, default='Results/Amazon/', help='Directory where to write matching Results ')
    p.add_argument('--resize_scale', type=float, default=0.125, help='resize scale;height,weight=scale*height,scale*weight')
    p.add_argument('--match_viz', default=True, help='Whether write the match result or not')
    p.add_argument('--show_keypoints', default=True, help='Show the detected keypoints')
    p.add_argument('--descriptor_dim', type=int, default=128, help='The dimension of feature descriptor')
    # superpoint hyper parameter
    p.add_argument('--superpoint_weights', type=str, default="superpoint/models/weights/superPointNet_allss_descriptor_128.pth.tar")
    p.add_argument('--keypoint_threshold', type=float, default=0.005, help='SuperPoint keypoint detector confidence threshold')
    p.add_argument('--nms_radius', type=int, default=4, help='SuperPoint Non Maximum Suppression (NMS) radius (Must be positive)')
    p.add_argument('--max_keypoints', type=int, default=-1, help="Maximum number of keypoints detected by Superpoint ('-1' keeps all keypoints)")
    # superglue hyper parameter
    p.add_argument('--superglue_weights', type=str, default='superglue/models/weights/SuperGlue_allss_descriptor_128.pth', help='SuperGlue weights')
    p.add_argument('--keypoint_encoder', default=[32, 64, 128], help='The dimension of keypoint encoder')
    p.add_argument('--sinkhorn_iterations', type=int, default=30, help='Number of Sinkhorn iterations performed by SuperGlue')
    p.add_argument('--match_threshold', type=float, default=0.1, help='SuperGlue match threshold')
    # not in the reference
    p.add_argument('--synthetic', type=int, default=0, help='write this many synthetic pairs under --img_dir first')
    p.add_argument('--ransac', choices=['gpu', 'host'], default='gpu',
                   help="partial-affine RANSAC: 'gpu' = libimx kernel (imx_estimate_affine_partial), 'host' = OpenCV/numpy")
    return p


def make_config(opt):
    kenc = opt.keypoint_encoder
    if isinstance(kenc, str):           # the reference leaves this flag untyped: "[32, 64, 128]" arrives as a string
        kenc = list(ast.literal_eval(kenc))

    def weights(path):                  # an absent checkpoint (the reference tree ships LFS pointers) -> synthetic weights
        if path and os.path.exists(path) and os.path.getsize(path) > 4096:
            return path
        if path:
            print(f"[imx] weights file {path!r} not found (or an LFS pointer): using synthetic weights")
        return None
    return {
        'superpoint': {'weights': weights(opt.superpoint_weights), 'descriptor_dim': opt.descriptor_dim,
                       'nms_radius': opt.nms_radius, 'keypoint_threshold': opt.keypoint_threshold,
                       'max_keypoints': opt.max_keypoints},
        'superglue': {'weights': weights(opt.superglue_weights), 'descriptor_dim': opt.descriptor_dim,
                      'keypoint_encoder': kenc, 'sinkhorn_iterations': opt.sinkhorn_iterations,
                      'match_threshold': opt.match_threshold},
    }


def write_synthetic_dataset(img_dir, n, scale):
    H, W = int(round(480 / scale)), int(round(640 / scale))
    os.makedirs(os.path.join(img_dir, 'source1'), exist_ok=True)
    os.makedirs(os.path.join(img_dir, 'template1'), exist_ok=True)
    tmpl, _ = synth.synth_pair(0, 480, 640)
    big = hostops.resize((tmpl * 255).astype(np.uint8), (W, H))
    hostops.imwrite(os.path.join(img_dir, 'template1', 'template.png'), big)
    for i in range(n):
        hostops.imwrite(os.path.join(img_dir, 'source1', f'src_{i:03d}.png'),
                        np.roll(big, (int(8 / scale) * (i + 1), int(16 / scale) * (i + 1)), axis=(0, 1)))


def load_pair(source_path, template_path, resize_scale):
    """datasets/SSHIDataset.py:14-29: grayscale read, resize by scale, /255."""
    so, to = hostops.imread_gray(source_path), hostops.imread_gray(template_path)
    if resize_scale is not None:
        si = hostops.resize(so, (int(resize_scale * so.shape[1]), int(resize_scale * so.shape[0])))
        ti = hostops.resize(to, (int(resize_scale * to.shape[1]), int(resize_scale * to.shape[0])))
    else:
        si, ti = so, to
    return so[None] / 255, si[None] / 255, ti[None] / 255


def main(argv=None):
    opt = build_parser().parse_args(argv)
    print(opt)
    if not torch.cuda.is_available():
        raise SystemExit("superpoint_glue_test.py (imx): needs an MI355X / ROCm GPU; there is no CPU path")
    device = 'cuda'
    config = make_config(opt)
    if opt.synthetic > 0:
        write_synthetic_dataset(opt.img_dir, opt.synthetic, opt.resize_scale or 1.0)
    matching = Matching(config).eval().to(device)
    if config['superpoint']['weights'] is None and opt.descriptor_dim in (128, 256):
        matching.superpoint.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in
                                             synth.make_superpoint_state_dict(opt.descriptor_dim).items()})
    if config['superglue']['weights'] is None and opt.descriptor_dim in synth.SG_CONFIGS \
            and list(config['superglue']['keypoint_encoder']) == synth.SG_CONFIGS[opt.descriptor_dim][0]:
        matching.superglue.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in
                                            synth.make_superglue_state_dict(opt.descriptor_dim).items()})
    return run_registration(opt, matching, device)


RANSAC_GPU_MAX = 8192      # imx_estimate_affine_partial: keypoint slots per pair (include/imx.h)


def gpu_affine_partial(eng, pred, thresh):
    """RANSAC partial-affine fit on the GPU over the MATCHED pairs only (compacted on the device, so the kernel's slot limit
    applies to the number of matches, not to --max_keypoints -1 keypoint counts).  Returns (M or None, mask (n,1)) like
    cv2.estimateAffinePartial2D, or (None, None) when the GPU path does not apply -- the caller then uses the host fit."""
    from image_matching_amd.engine import ImxError
    m0 = pred['matches0'][0].long()
    sel = m0 > -1
    n = int(sel.sum())
    if n <= 3 or n > RANSAC_GPU_MAX:
        return None, None
    mk0 = pred['keypoints0'][0][sel][None].contiguous()
    mk1 = pred['keypoints1'][0][m0[sel]][None].contiguous()
    ident = torch.arange(n, device=m0.device, dtype=torch.int64)[None]
    try:
        M, inl, ninl = eng.estimate_affine_partial(mk0, mk1, ident, ransac_thresh=thresh)
    except ImxError as e:
        print(f"[imx] GPU RANSAC unavailable ({e}); falling back to the host fit")
        return None, None
    return (M[0].cpu().numpy() if int(ninl[0]) > 0 else None), inl[0].cpu().numpy()[:, None]


def run_registration(opt, matching, device):
    """The per-pair loop shared by superpoint_glue_test.py:72-140 and superpoint_glue_official_test.py:66-137."""
    source_dir = opt.img_dir + 'source1/'
    template_dir = opt.img_dir + 'template1/'
    template_img_path = template_dir + os.listdir(template_dir)[0]
    results = []
    Matrix = None
    for filename in sorted(os.listdir(source_dir)):
        source_original, source_image, template_image = load_pair(source_dir + filename, template_img_path, opt.resize_scale)
        source_tensor = torch.from_numpy(source_image)[None].float().to(device)
        template_tensor = torch.from_numpy(template_image)[None].float().to(device)
        start = time.perf_counter()
        pred = matching({'image0': source_tensor, 'image1': template_tensor})
        kpts0 = pred['keypoints0'][0].cpu().numpy()
        kpts1 = pred['keypoints1'][0].cpu().numpy()
        matches = pred['matches0'][0].cpu().numpy()
        confidence = pred['matching_scores0'][0].cpu().numpy()
        valid = matches > -1
        mkpts0, mkpts1 = kpts0[valid], kpts1[matches[valid]]
        if len(mkpts0) > 3:
            M, mask = None, None
            if opt.ransac == 'gpu':
                M, mask = gpu_affine_partial(matching._shared.engine, pred, 7)
            if mask is None:       # --ransac host, more matched pairs than the kernel takes, or a failed GPU call
                M, mask = hostops.estimate_affine_partial_2d(mkpts0, mkpts1, ransac_thresh=7)
            if M is not None:       # the reference crashes on a failed fit; we keep the last matrix
                Matrix = np.array(M, dtype=np.float64)
                if opt.resize_scale is not None:
                    Matrix[:, 2] = Matrix[:, 2] / opt.resize_scale
                flag = (mask > 0).ravel().tolist()
                mkpts0, mkpts1 = mkpts0[flag], mkpts1[flag]
        print("Time used:", time.perf_counter() - start)
        results.append((filename, len(kpts0), len(kpts1), int(valid.sum()), len(mkpts0), None if Matrix is None else Matrix.copy()))

        src255 = source_original.squeeze() * 255
        if Matrix is not None:
            if opt.ransac == 'gpu':        # warp on the GPU too (imx_warp_affine_u8): same fixed-point arithmetic as cv2.warpAffine
                src_u8 = torch.from_numpy(np.rint(src255).astype(np.uint8))
                Transform = matching._shared.engine.warp_affine_u8(src_u8, Matrix).cpu().numpy()
            else:
                Transform = hostops.warp_affine(src255, Matrix, (src255.shape[1], src255.shape[0]))
            Transform_dir = os.path.join(opt.Result_dir, opt.exper_name, 'Transform/')
            os.makedirs(Transform_dir, exist_ok=True)
            hostops.imwrite(Transform_dir + 'trans_{}'.format(filename), Transform)
        if opt.match_viz:
            import matplotlib.cm as cm
            color = cm.jet(confidence[valid])[:len(mkpts0)] if len(mkpts0) else np.zeros((0, 4))
            text = ['SuperGlue', 'Keypoints: {}:{}'.format(len(kpts0), len(kpts1)), 'Matches: {}'.format(len(mkpts0))]
            small_text = ['Keypoint Threshold: {:.4f}'.format(matching.superpoint.config['keypoint_threshold']),
                          'Match Threshold: {:.2f}'.format(matching.superglue.config['match_threshold']), ' ']
            Match_dir = os.path.join(opt.Result_dir, opt.exper_name, 'Match/')
            os.makedirs(Match_dir, exist_ok=True)
            out_file = str(Path(Match_dir, filename))
            print('\nWriting image to {}'.format(out_file))
            hostops.make_matching_plot_fast(source_image.squeeze() * 255, template_image.squeeze() * 255, kpts0, kpts1,
                                            mkpts0, mkpts1, color, text, path=out_file,
                                            show_keypoints=opt.show_keypoints, small_text=small_text)
    return results


if __name__ == '__main__':
    main()
