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
        fused = self._forward_fused(data)
        if fused is not None:
            return fused
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

    def _forward_fused(self, data):
        """Latency path for the reference CLI's case (one pair per call, equal image shapes, max_keypoints > 0, no keypoints
        supplied): ONE fused C-ABI call (imx_match_pairs: SuperPoint on both images as a batch of two, SuperGlue with
        device-side keypoint counts) and one host synchronisation at the end for the counts, instead of two SuperPoint calls
        with a sync each plus SuperGlue.  Returns the reference's dict, or None when the generic path must run."""
        if 'keypoints0' in data or 'keypoints1' in data:
            return None
        im0, im1 = data.get('image0'), data.get('image1')
        K = self.superpoint.config['max_keypoints']
        if not (torch.is_tensor(im0) and torch.is_tensor(im1)) or im0.shape != im1.shape or im0.dim() != 4 or im0.shape[0] != 1 \
                or K <= 0 or not im0.is_cuda or not im1.is_cuda:
            return None
        out = self._shared.get_engine([0, 1]).match_pairs(im0, im1, want_desc=True)
        n0, n1 = int(out['counts0'][0]), int(out['counts1'][0])          # the one host sync
        if n0 == 0 or n1 == 0:
            return None                 # empty-set early-out (int32 matches, superglue_test.py:235-242): generic path
        return {
            'keypoints0': [out['keypoints0'][0, :n0]], 'scores0': (out['scores0'][0, :n0],),
            'descriptors0': [out['descriptors0'][0, :n0].t()],
            'keypoints1': [out['keypoints1'][0, :n1]], 'scores1': (out['scores1'][0, :n1],),
            'descriptors1': [out['descriptors1'][0, :n1].t()],
            'matches0': out['matches0'][:, :n0], 'matches1': out['matches1'][:, :n1],
            'matching_scores0': out['matching_scores0'][:, :n0], 'matching_scores1': out['matching_scores1'][:, :n1],
        }

    def match_batch(self, image0, image1, want_desc=False):
        """Throughput path (no reference equivalent): B pairs through the fused C-ABI call
        imx_match_pairs with fixed max_keypoints; returns padded tensors + per-image counts and
        does not synchronise the host."""
        eng = self._shared.get_engine([0, 1])
        return eng.match_pairs(image0, image1, want_desc)

    def pack_records(self, pair_ids, out, pad_to=None):
        """match_batch's output as the fixed-size match records of the multi-GPU gather (image_matching_amd/shard.py), packed
        on the GPU by one kernel; equal, word for word, to shard.pack_records (the host-side statement of the layout)."""
        return self._shared.get_engine([0, 1]).pack_records(pair_ids, out, pad_to)
