"""Drop-in for superglue/models/matching.py:47-82 (Matching over the official SuperPoint,
which takes {'image': x})."""
from .matching_test import Matching as _MatchingBase
from .superpoint import SuperPoint


class Matching(_MatchingBase):
    _superpoint_cls = SuperPoint

    def _run_superpoint(self, image):
        return self.superpoint({'image': image})
