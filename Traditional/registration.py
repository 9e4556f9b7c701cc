"""BASELINE.json configs[0] plumbing: feature-based registration with OpenCV's SIFT / ORB detectors, FLANN / brute-force
matching and cv2.estimateAffinePartial2D (reference Traditional/registration.py:6-88).  Every arithmetic step lives inside
OpenCV (third-party; README.md:24-25 pins opencv-contrib-python 4.5.1.48, absent from this image and not installable without
network access): there is no GPU path and no parity claim for this configuration -- see DESIGN.md.  What this module keeps is
the call contract of the two reference functions: `(src_img, temp_img, RESIZE_SCALE, MATCH_VIZ) -> (M 2x3, match_img)`, or
`None` when fewer than MIN_MATCH_COUNT + 1 matches survive (the reference returns a bare None there, which its caller cannot
unpack -- SURVEY App. B; traditional.py here checks for it)."""
import numpy as np

try:
    import cv2
except ImportError:          # the image used for this work has no OpenCV: traditional.py prints a skip line
    cv2 = None

MIN_MATCH_COUNT = 10          # registration.py:4
RATIO = 0.7                   # registration.py:28
RANSAC_THRESH = 7             # registration.py:35,71


def available(method="SIFT"):
    """(ok, reason): can `method` run with the OpenCV build in this process?"""
    if cv2 is None:
        return False, "cv2 (opencv-contrib-python) is not installed"
    if method == "SIFT" and not (hasattr(cv2, "xfeatures2d") and hasattr(cv2.xfeatures2d, "SIFT_create")) and not hasattr(cv2, "SIFT_create"):
        return False, "this OpenCV build has no SIFT (needs opencv-contrib-python)"
    return True, ""


def _prepare(src_img, temp_img, scale):
    h, w = temp_img.shape[:2]
    if scale is not None:         # both images are resized to the TEMPLATE's scaled size (registration.py:9-11)
        size = (int(scale * w), int(scale * h))
        src_img = cv2.resize(src_img, size, interpolation=cv2.INTER_CUBIC)
        temp_img = cv2.resize(temp_img, size, interpolation=cv2.INTER_CUBIC)
    return src_img, temp_img, cv2.cvtColor(src_img, cv2.COLOR_RGB2GRAY), cv2.cvtColor(temp_img, cv2.COLOR_RGB2GRAY)


def _fit_and_draw(src_img, kp1, temp_img, kp2, matches, viz):
    src = np.float32([kp1[m.queryIdx].pt for m in matches])
    dst = np.float32([kp2[m.trainIdx].pt for m in matches])
    M, mask = cv2.estimateAffinePartial2D(src, dst, method=cv2.RANSAC, ransacReprojThreshold=RANSAC_THRESH)
    img = None
    if viz and mask is not None:
        img = cv2.drawMatches(src_img, kp1, temp_img, kp2, matches, None, matchColor=(0, 255, 0), singlePointColor=None,
                              matchesMask=mask.ravel().tolist(), flags=0)
    return M, img


def SIFT_REGIS(src_img, temp_img, RESIZE_SCALE=None, MATCH_VIZ=False):
    """SIFT keypoints + FLANN KD-tree 2-NN + ratio test + RANSAC partial affine (registration.py:6-49)."""
    src_img, temp_img, g1, g2 = _prepare(src_img, temp_img, RESIZE_SCALE)
    sift = cv2.xfeatures2d.SIFT_create() if hasattr(cv2, "xfeatures2d") and hasattr(cv2.xfeatures2d, "SIFT_create") else cv2.SIFT_create()
    kp1, d1 = sift.detectAndCompute(g1, None)
    kp2, d2 = sift.detectAndCompute(g2, None)
    flann = cv2.FlannBasedMatcher(dict(algorithm=0, trees=5), dict(checks=50))          # FLANN_INDEX_KDTREE
    good = [m for m, n in flann.knnMatch(d1, d2, k=2) if m.distance < RATIO * n.distance]
    if len(good) <= MIN_MATCH_COUNT:
        print("SIFT:Not enough matches are found - %d/%d" % (len(good), MIN_MATCH_COUNT))
        return None
    return _fit_and_draw(src_img, kp1, temp_img, kp2, good, MATCH_VIZ)


def ORB_REGIS(src_img, temp_img, RESIZE_SCALE=None, MATCH_VIZ=False):
    """ORB keypoints + cross-checked Hamming brute force + RANSAC partial affine (registration.py:51-88)."""
    src_img, temp_img, g1, g2 = _prepare(src_img, temp_img, RESIZE_SCALE)
    orb = cv2.ORB_create()
    kp1, d1 = orb.detectAndCompute(g1, None)
    kp2, d2 = orb.detectAndCompute(g2, None)
    matches = sorted(cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True).match(d1, d2), key=lambda m: m.distance)
    if len(matches) <= MIN_MATCH_COUNT:
        print("ORB:Not enough matches are found - %d/%d" % (len(matches), MIN_MATCH_COUNT))
        return None
    return _fit_and_draw(src_img, kp1, temp_img, kp2, matches, MATCH_VIZ)
