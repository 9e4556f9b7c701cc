#!/usr/bin/env python
"""BASELINE.json configs[0]: SIFT/ORB + FLANN registration of one pair directory on the CPU (reference traditional.py:8-57).
Plumbing only: the arithmetic is OpenCV's (third-party), there is no GPU path here and no parity claim.  The
reference's flags, directory convention (<img_dir>/source1/*, <img_dir>/template1/<one>) and outputs
(<Result_dir>/<Method>/Transform1/trans_*, .../Match1/match_*) are kept.  Without OpenCV -- the case in this image, where
`pip download opencv-contrib-python` finds no index (tools/try_opencv.sh) -- it prints one skip line and exits 0."""
import argparse
import os
import sys
import time

from Traditional import registration


def build_parser():
    p = argparse.ArgumentParser(description='Traditional Registration', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--Method', type=str, default='SIFT', help='The method of feature based registration')
    p.add_argument('--img_dir', type=str, default='datasets/Amazon/', help='path to source image directory')
    p.add_argument('--Result_dir', type=str, default='Results/Amazon/', help='Directory where to write matching Results ')
    p.add_argument('--resize_scale', type=float, default=0.5, help='resize scale;height,weight=scale*height,scale*weight')
    p.add_argument('--match_viz', default=True, help='Whether write the match result or not')
    return p


def main(argv=None):
    opt = build_parser().parse_args(argv)
    if opt.Method not in ('SIFT', 'ORB'):
        raise SystemExit(f"--Method must be SIFT or ORB (got {opt.Method!r})")
    ok, why = registration.available(opt.Method)
    if not ok:
        print(f"traditional.py: skipped -- {why}; this configuration is OpenCV-only CPU plumbing (BASELINE configs[0])")
        return []
    import cv2
    source_dir, template_dir = opt.img_dir + 'source1/', opt.img_dir + 'template1/'
    template_img = cv2.imread(template_dir + os.listdir(template_dir)[0])
    regis = registration.SIFT_REGIS if opt.Method == 'SIFT' else registration.ORB_REGIS
    results = []
    for name in sorted(os.listdir(source_dir)):
        source_img = cv2.imread(os.path.join(source_dir, name))
        start = time.perf_counter()
        out = regis(source_img, template_img, opt.resize_scale, opt.match_viz)
        if out is None or out[0] is None:       # the reference crashes unpacking None here: skip the pair
            results.append((name, None))
            continue
        Matrix, match_img = out
        if opt.resize_scale is not None:
            Matrix[:, 2] = Matrix[:, 2] / opt.resize_scale
        print("Time used:", time.perf_counter() - start)
        Transform_dir = os.path.join(opt.Result_dir, opt.Method + '/Transform1/')
        Match_dir = os.path.join(opt.Result_dir, opt.Method + '/Match1/')
        os.makedirs(Transform_dir, exist_ok=True)
        os.makedirs(Match_dir, exist_ok=True)
        cv2.imwrite(Transform_dir + 'trans_{}'.format(name),
                    cv2.warpAffine(source_img, Matrix, (template_img.shape[1], template_img.shape[0])))
        if opt.match_viz and match_img is not None:
            cv2.imwrite(Match_dir + 'match_{}'.format(name), match_img)
        results.append((name, Matrix))
    return results


if __name__ == "__main__":
    main()
    sys.exit(0)
