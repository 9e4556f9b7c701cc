"""Drop-in for the reference's Matching frontend (superglue/models/matching_test.py:47-82):
SuperPoint (BN variant) on image0/image1 unless keypoints are supplied, then SuperGlue."""
import torch

from ..._shared import ModelBase, Shared
from ...superpoint.models.superpoint_test import SuperPoint
from .superglue_test import SuperGlue


class Matching(ModelBase):
    """ Image Matching Frontend (SuperPoint + SuperGlue), MI355X-native """
    _superpoint_cls = SuperPoint

    def __init__(self, config={}):
        super().__init__()
        self._init_shared(Shared())
        self.superpoint = self._superpoint_cls(config.get('superpoint', {}), _shared=self._shared)
        self.superglue = SuperGlue(config.get('superglue', {}), _shared=self._shared)
        self.train(False)

    def _run_superpoint(self, image):
        return self.superpoint(image)

    def forward(self, data):
        """data: {'image0','image1'} (+ optionally keypoints/scores/descriptors 0/1 to skip SuperPoint)."""
        pred = {}
        if 'keypoints0' not in data:
            pred0 = self._run_superpoint(data['image0'])
            pred = {**pred, **{k + '0': v for k, v in pred0.items()}}
        if 'keypoints1' not in data:
            pred1 = self._run_superpoint(data['image1'])
            pred = {**pred, **{k + '1': v for k, v in pred1.items()}}
        # one image per batch, or the same number of local features for all images in the batch
        data = {**data, **pred}
        for k in data:
            if isinstance(data[k], (list, tuple)):
                data[k] = torch.stack(data[k])
        pred = {**pred, **self.superglue(data)}
        return pred

    def match_batch(self, image0, image1, want_desc=False):
        """Throughput path (no reference equivalent): B pairs through the fused C-ABI call
        imx_match_pairs with fixed max_keypoints; returns padded tensors + per-image counts and
        does not synchronise the host."""
        eng = self._shared.get_engine([0, 1])
        return eng.match_pairs(image0, image1, want_desc)
