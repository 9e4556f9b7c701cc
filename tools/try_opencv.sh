#!/bin/bash
# VERDICT r1 "missing" #2 / next-round #8: can OpenCV be obtained in the build container, so that the three OpenCV-shaped rows
# (RANSAC partial affine, FLANN matcher, cv2.resize/warpAffine ingest) get byte fixtures from cv2 itself?
# `pip download` only FETCHES (it installs nothing); it needs a package index, and this container has none.
# Recorded outcome (2026-09-28, this container):
#   ERROR: Could not find a version that satisfies the requirement opencv-python-headless==4.5.5.64 (from versions: none)
#   ERROR: No matching distribution found for opencv-python-headless==4.5.5.64
# and `python -c "import cv2"` -> ModuleNotFoundError.  The rows therefore stay "partial" (parity vs cv2 unpinned; each kernel is
# bit-exact against a restatement of OpenCV's published algorithm under oracle/).
# The same on the GPU box (round 5): tools/try_opencv_gpu_box.txt -- no cv2, no index, no DNS there either.
set -x
python -c "import cv2; print(cv2.__version__)" || true
pip download opencv-python-headless==4.5.5.64 -d /tmp/cvwheel --no-deps || true
pip download opencv-contrib-python==4.5.1.48 -d /tmp/cvwheel --no-deps || true
