"""Host-side engine: one libimx handle per (device, config), weight upload, and tensor-level
wrappers of the C-ABI entry points.  torch tensors are containers only (data_ptr + stream)."""
import ctypes

import numpy as np
import torch

from . import _lib as L

SP_DEFAULT = {                      # superpoint/models/superpoint_test.py:57-63
    "descriptor_dim": 256, "nms_radius": 4, "keypoint_threshold": 0.005,
    "max_keypoints": -1, "remove_borders": 4,
}
SG_DEFAULT = {                      # superglue/models/superglue_test.py:195-202
    "descriptor_dim": 256, "weights": "indoor", "keypoint_encoder": [32, 64, 128, 256],
    "GNN_layers": ["self", "cross"] * 9, "sinkhorn_iterations": 100, "match_threshold": 0.2,
}


def reference_align_corners():
    """What `int(torch.__version__[2]) > 2` (superpoint_test.py:47) evaluates to under the torch
    in this process: True for torch 1.3-1.9, False for 1.10+/2.x (third character '1'/'0')."""
    try:
        return int(torch.__version__[2]) > 2
    except ValueError:
        return False


class ImxError(RuntimeError):
    pass


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Engine:
    """Owns an imx handle.  `sp_cfg` / `sg_cfg` are the merged reference-style config dicts."""

    def __init__(self, sp_cfg, sg_cfg, device, sp_variant=L.SP_VARIANT_BN, align_corners=None):
        self.lib = L.load_library()
        if not torch.cuda.is_available():
            raise ImxError("image_matching_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback on the product path")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ImxError(f"image_matching_amd runs on 'cuda' (HIP) devices only, got {device!r}")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        sp = {**SP_DEFAULT, **(sp_cfg or {})}
        sg = {**SG_DEFAULT, **(sg_cfg or {})}
        cfg = L.ImxConfig()
        cfg.descriptor_dim = int(sp["descriptor_dim"])
        cfg.nms_radius = int(sp["nms_radius"])
        cfg.keypoint_threshold = float(sp["keypoint_threshold"])
        cfg.max_keypoints = int(sp["max_keypoints"])
        cfg.remove_borders = int(sp["remove_borders"])
        cfg.align_corners = int(reference_align_corners() if align_corners is None else bool(align_corners))
        cfg.sp_variant = int(sp_variant)
        layers = list(sg["GNN_layers"])
        if len(layers) > L.IMX_MAX_GNN_LAYERS:
            raise ImxError(f"at most {L.IMX_MAX_GNN_LAYERS} GNN layers supported")
        cfg.num_gnn_layers = len(layers)
        for i, name in enumerate(layers):
            cfg.gnn_layer_is_cross[i] = 1 if name == "cross" else 0
        kenc = [int(c) for c in sg["keypoint_encoder"]]
        cfg.kenc_n = len(kenc)
        for i, c in enumerate(kenc):
            cfg.kenc_channels[i] = c
        cfg.sinkhorn_iterations = int(sg["sinkhorn_iterations"])
        cfg.match_threshold = float(sg["match_threshold"])
        self.cfg = cfg
        self.d = cfg.descriptor_dim
        self.max_keypoints = cfg.max_keypoints
        self.handle = ctypes.c_void_p()
        if self.lib.imx_create(idx, ctypes.byref(cfg), ctypes.byref(self.handle)) != 0:
            raise ImxError(self.lib.imx_last_error(None).decode())
        self.loaded = {L.NET_SUPERPOINT: False, L.NET_SUPERGLUE: False}

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.handle.value:
                self.lib.imx_destroy(self.handle)
                self.handle = ctypes.c_void_p()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise ImxError(self.lib.imx_last_error(self.handle).decode())

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, net, state_dict):
        for key, val in state_dict.items():
            if key.endswith("num_batches_tracked"):
                continue
            arr = val.detach().cpu().numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (ctypes.c_int64 * max(arr.ndim, 1))(*arr.shape)
            self._check(self.lib.imx_load_weight(self.handle, net, key.encode(), arr.ctypes.data_as(ctypes.c_void_p),
                                                 arr.ndim, shape))
        self._check(self.lib.imx_finalize_weights(self.handle, net))
        self.loaded[net] = True

    # ------------------------------------------------------------------ SuperPoint
    def superpoint(self, x):
        """x (B,1,H,W) float32 cuda -> (kpts (B,K,2), scores (B,K), desc (B,K,d), counts list)."""
        x = self._image(x)
        B, _, H, W = x.shape
        counts = torch.empty(B, dtype=torch.int32, device=self.device)
        st = _stream(self.device)
        self._check(self.lib.imx_superpoint_detect(self.handle, _ptr(x), B, H, W, _ptr(counts), st))
        n = counts.cpu().tolist()                       # device sync (the reference syncs at nonzero())
        K = max(n) if n else 0
        kpts = torch.empty(B, K, 2, dtype=torch.float32, device=self.device)
        scores = torch.empty(B, K, dtype=torch.float32, device=self.device)
        desc = torch.empty(B, K, self.d, dtype=torch.float32, device=self.device)
        if K > 0:
            self._check(self.lib.imx_superpoint_describe(self.handle, B, K, _ptr(kpts), _ptr(scores), _ptr(desc), st))
        return kpts, scores, desc, n

    def superpoint_dense(self, x):
        """Dense forward (superpoint_train.py:31-57): semi (B,65,H/8,W/8), desc (B,d,H/8,W/8) unit-norm."""
        x = self._image(x)
        B, _, H, W = x.shape
        Hc, Wc = H // 2 // 2 // 2, W // 2 // 2 // 2
        semi = torch.empty(B, 65, Hc, Wc, dtype=torch.float32, device=self.device)
        desc = torch.empty(B, self.d, Hc, Wc, dtype=torch.float32, device=self.device)
        self._check(self.lib.imx_superpoint_dense(self.handle, _ptr(x), B, H, W, _ptr(semi), _ptr(desc), _stream(self.device)))
        return semi, desc

    def superpoint_batch(self, x):
        """Throughput path: SuperPoint with fixed max_keypoints = K > 0, no host sync.  Returns padded
        (kpts (B,K,2), scores (B,K), desc (B,K,d), counts (B) int32); rows >= count are zero."""
        x = self._image(x)
        B, _, H, W = x.shape
        K = self.max_keypoints
        if K <= 0:
            raise ImxError("superpoint_batch needs max_keypoints > 0")
        counts = torch.empty(B, dtype=torch.int32, device=self.device)
        kpts = torch.empty(B, K, 2, dtype=torch.float32, device=self.device)
        scores = torch.empty(B, K, dtype=torch.float32, device=self.device)
        desc = torch.empty(B, K, self.d, dtype=torch.float32, device=self.device)
        st = _stream(self.device)
        self._check(self.lib.imx_superpoint_detect(self.handle, _ptr(x), B, H, W, _ptr(counts), st))
        self._check(self.lib.imx_superpoint_describe(self.handle, B, K, _ptr(kpts), _ptr(scores), _ptr(desc), st))
        return kpts, scores, desc, counts

    def _image(self, x):
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != 1:
            raise ImxError(f"expected an image tensor of shape (B,1,H,W), got {getattr(x, 'shape', type(x))}")
        if x.device.type != "cuda":
            raise ImxError("image tensor must live on the GPU (the reference CLI moves it with .to(device)); "
                           "there is no CPU path")
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    # ------------------------------------------------------------------ SuperGlue
    def superglue(self, kpts0, scores0, desc0, shape0, kpts1, scores1, desc1, shape1, n0=None, n1=None):
        """desc{0,1}: (B,d,N) tensors with arbitrary strides.  Returns matches0/1 (int64), mscores0/1."""
        dev = self.device
        kpts0 = kpts0.to(dev, torch.float32).contiguous()
        kpts1 = kpts1.to(dev, torch.float32).contiguous()
        scores0 = scores0.to(dev, torch.float32).contiguous()
        scores1 = scores1.to(dev, torch.float32).contiguous()
        desc0 = desc0.to(dev, torch.float32)
        desc1 = desc1.to(dev, torch.float32)
        B, N0 = kpts0.shape[0], kpts0.shape[1]
        N1 = kpts1.shape[1]
        if desc0.shape[1] != self.d or desc1.shape[1] != self.d:
            raise ImxError(f"descriptor dim {desc0.shape[1]} != configured descriptor_dim {self.d}")
        m0 = torch.empty(B, N0, dtype=torch.int64, device=dev)
        m1 = torch.empty(B, N1, dtype=torch.int64, device=dev)
        ms0 = torch.empty(B, N0, dtype=torch.float32, device=dev)
        ms1 = torch.empty(B, N1, dtype=torch.float32, device=dev)
        s0, s1 = desc0.stride(), desc1.stride()
        self._check(self.lib.imx_superglue_forward(
            self.handle, B,
            _ptr(kpts0), _ptr(scores0), _ptr(desc0), s0[0], s0[1], s0[2], _ptr(n0), N0, int(shape0[-2]), int(shape0[-1]),
            _ptr(kpts1), _ptr(scores1), _ptr(desc1), s1[0], s1[1], s1[2], _ptr(n1), N1, int(shape1[-2]), int(shape1[-1]),
            _ptr(m0), _ptr(m1), _ptr(ms0), _ptr(ms1), _stream(dev)))
        return m0, m1, ms0, ms1

    # ------------------------------------------------------------------ fused pairs
    def match_pairs(self, img0, img1, want_desc=False):
        """Fused Matching.forward for B pairs (max_keypoints = K > 0); no host sync.
        Returns dict of padded tensors: keypoints{0,1} (B,K,2), scores{0,1} (B,K), counts{0,1} (B),
        matches{0,1} (B,K) int64, matching_scores{0,1} (B,K) [, descriptors{0,1} (B,K,d)]."""
        img0, img1 = self._image(img0), self._image(img1)
        if img0.shape != img1.shape:
            raise ImxError("match_pairs needs equal image shapes on both sides")
        B, _, H, W = img0.shape
        K, dev = self.max_keypoints, self.device
        if K <= 0:
            raise ImxError("match_pairs needs max_keypoints > 0")
        f32, out = torch.float32, {}
        for s in "01":
            out["keypoints" + s] = torch.empty(B, K, 2, dtype=f32, device=dev)
            out["scores" + s] = torch.empty(B, K, dtype=f32, device=dev)
            out["counts" + s] = torch.empty(B, dtype=torch.int32, device=dev)
            out["matches" + s] = torch.empty(B, K, dtype=torch.int64, device=dev)
            out["matching_scores" + s] = torch.empty(B, K, dtype=f32, device=dev)
            if want_desc:
                out["descriptors" + s] = torch.empty(B, K, self.d, dtype=f32, device=dev)
        self._check(self.lib.imx_match_pairs(
            self.handle, _ptr(img0), _ptr(img1), B, H, W,
            _ptr(out["keypoints0"]), _ptr(out["keypoints1"]), _ptr(out["scores0"]), _ptr(out["scores1"]),
            _ptr(out["counts0"]), _ptr(out["counts1"]),
            _ptr(out.get("descriptors0")), _ptr(out.get("descriptors1")),
            _ptr(out["matches0"]), _ptr(out["matches1"]),
            _ptr(out["matching_scores0"]), _ptr(out["matching_scores1"]), _stream(dev)))
        return out

    def pack_records(self, pair_ids, out, pad_to=None):
        """The match records of shard.py (one row of 3 + 8K 32-bit words per pair) from match_pairs' output dict, packed by one
        kernel (imx_pack_records) on the current stream.  pair_ids: (B) int32 tensor on the device (or a list: copied once per
        call -- callers with a fixed shard keep the tensor).  Rows past B up to pad_to are padding (pair id -1)."""
        B, K = out["matches0"].shape
        rows = max(B, pad_to or 0)
        if not isinstance(pair_ids, torch.Tensor):
            pair_ids = torch.as_tensor(list(pair_ids), dtype=torch.int32)
        pair_ids = pair_ids.to(self.device, torch.int32).contiguous()
        # the kernel reads pair_ids[row] for every row < B and reinterprets raw pointers: refuse anything that is not what
        # match_pairs returns instead of reading out of bounds or through the wrong dtype
        if pair_ids.numel() != B:
            raise ImxError(f"pack_records: {pair_ids.numel()} pair ids for a batch of {B} pairs")
        want = {"keypoints0": (torch.float32, (B, K, 2)), "keypoints1": (torch.float32, (B, K, 2)), "counts0": (torch.int32, (B,)), "counts1": (torch.int32, (B,)),
                "matches0": (torch.int64, (B, K)), "matches1": (torch.int64, (B, K)), "matching_scores0": (torch.float32, (B, K)), "matching_scores1": (torch.float32, (B, K))}
        for key, (dt, shp) in want.items():
            t = out[key]
            if not (isinstance(t, torch.Tensor) and t.dtype == dt and tuple(t.shape) == shp and t.device == self.device and t.is_contiguous()):
                raise ImxError(f"pack_records: out[{key!r}] must be a contiguous {dt} tensor of shape {shp} on {self.device} "
                               f"(got {getattr(t, 'dtype', type(t))}, {tuple(getattr(t, 'shape', ()))}, {getattr(t, 'device', None)})")
        rec = torch.empty(rows, 3 + 8 * K, dtype=torch.int32, device=self.device)
        self._check(self.lib.imx_pack_records(
            self.handle, _ptr(pair_ids), B, K, _ptr(out["keypoints0"]), _ptr(out["keypoints1"]), _ptr(out["counts0"]), _ptr(out["counts1"]),
            _ptr(out["matches0"]), _ptr(out["matches1"]), _ptr(out["matching_scores0"]), _ptr(out["matching_scores1"]),
            _ptr(rec), rows, _stream(self.device)))
        return rec

    def estimate_affine_partial(self, kpts0, kpts1, matches0, counts0=None, ransac_thresh=7.0, hypotheses=512, seed=0):
        """Batched RANSAC partial-affine fit (superpoint_glue_test.py:86-92) on the GPU.
        kpts{0,1} (B,K,2), matches0 (B,K) int64.  Returns M (B,2,3), inlier mask (B,K) uint8, n_inliers (B) int32."""
        dev = self.device
        kpts0 = kpts0.to(dev, torch.float32).contiguous()
        kpts1 = kpts1.to(dev, torch.float32).contiguous()
        matches0 = matches0.to(dev, torch.int64).contiguous()
        B, K0 = matches0.shape
        K = max(K0, kpts1.shape[1])
        if K0 < K:                        # pad side 0 (unmatched) so both sides share K; indices stay valid
            kpts0 = torch.cat([kpts0, kpts0.new_zeros(B, K - K0, 2)], 1)
            matches0 = torch.cat([matches0, matches0.new_full((B, K - K0), -1)], 1)
        if kpts1.shape[1] < K:
            kpts1 = torch.cat([kpts1, kpts1.new_zeros(B, K - kpts1.shape[1], 2)], 1)
        M = torch.empty(B, 2, 3, dtype=torch.float32, device=dev)
        inl = torch.empty(B, K, dtype=torch.uint8, device=dev)
        ninl = torch.empty(B, dtype=torch.int32, device=dev)
        self._check(self.lib.imx_estimate_affine_partial(self.handle, _ptr(kpts0), _ptr(kpts1), _ptr(matches0), _ptr(counts0),
                                                         B, K, float(ransac_thresh), int(hypotheses), int(seed) & 0xFFFFFFFF,
                                                         _ptr(M), _ptr(inl), _ptr(ninl), _stream(dev)))
        return M, inl[:, :K0], ninl

    def knn_ratio_match(self, desc0, desc1, ratio=0.7, n0=None, n1=None):
        """Exact 2-NN + ratio test (superpoint_flann_test.py:66-74 without FLANN's approximation).
        desc{0,1}: (B,d,N) tensors, any strides.  Returns matches (B,N0) int64 (-1 = rejected), dist1, dist2 (B,N0)."""
        dev = self.device
        desc0, desc1 = desc0.to(dev, torch.float32), desc1.to(dev, torch.float32)
        B, _, N0 = desc0.shape
        N1 = desc1.shape[2]
        m = torch.empty(B, N0, dtype=torch.int64, device=dev)
        d1 = torch.empty(B, N0, dtype=torch.float32, device=dev)
        d2 = torch.empty(B, N0, dtype=torch.float32, device=dev)
        if N0 == 0:
            return m, d1, d2
        s0, s1 = desc0.stride(), desc1.stride()
        self._check(self.lib.imx_knn_ratio_match(self.handle, B, _ptr(desc0), s0[0], s0[1], s0[2], _ptr(n0), N0,
                                                 _ptr(desc1), s1[0], s1[1], s1[2], _ptr(n1), N1, float(ratio),
                                                 _ptr(m), _ptr(d1), _ptr(d2), _stream(dev)))
        return m, d1, d2

    def ingest(self, img_u8, size_hw=None, out=None):
        """cv2.resize + /255 of datasets/SSHIDataset.py:19-27 on the GPU.  img_u8: (B,Hs,Ws) or (Hs,Ws) uint8 tensor
        (moved to the device if needed); size_hw None = no resize.  Returns (B,1,H,W) float32 in [0,1]."""
        if img_u8.dtype != torch.uint8:
            raise TypeError("ingest expects uint8 gray images")
        src = img_u8.to(self.device, non_blocking=True)
        if src.dim() == 2:
            src = src[None]
        src = src.contiguous()
        B, Hs, Ws = src.shape
        H, W = (Hs, Ws) if size_hw is None else size_hw
        dst = out if out is not None else torch.empty(B, 1, H, W, dtype=torch.float32, device=self.device)
        self._check(self.lib.imx_ingest_resize_u8(self.handle, _ptr(src), B, Hs, Ws, Hs * Ws, _ptr(dst), H, W, _stream(self.device)))
        return dst

    def warp_affine_u8(self, src_u8, M, size_hw=None):
        """cv2.warpAffine(source_original*255, M, (W,H)) as cv2.imwrite stores it (superpoint_glue_test.py:101-113).
        src_u8 (Hs,Ws) uint8 tensor; M forward 2x3 (array-like, host).  Returns (H,W) uint8 on the device."""
        src = src_u8.to(self.device).contiguous()
        Hs, Ws = src.shape
        H, W = (Hs, Ws) if size_hw is None else size_hw
        import numpy as np
        Mh = np.ascontiguousarray(np.asarray(M, np.float64).reshape(6))
        dst = torch.empty(H, W, dtype=torch.uint8, device=self.device)
        self._check(self.lib.imx_warp_affine_u8(self.handle, _ptr(src), Hs, Ws, Mh.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                _ptr(dst), H, W, _stream(self.device)))
        return dst

    def op_nms(self, scores, radius):
        """simple_nms on a (B,H,W) score map (single-stage entry point, used by parity tests)."""
        scores = scores.to(self.device, torch.float32).contiguous()
        out = torch.empty_like(scores)
        B, H, W = scores.shape
        self._check(self.lib.imx_op_nms(self.handle, _ptr(scores), _ptr(out), B, H, W, int(radius), _stream(self.device)))
        return out

    # ------------------------------------------------------------------ kernel-form options (include/imx.h: imx_set_option)
    def set_option(self, key, value):
        """'mfma' = 'x3' | 'f32', 'latency_forms' = 'auto' | 'off' | 'on' | 'unfused', 'conv' = 'wino' | 'wino_h' | 'wino32' | 'direct',
        'gnn_tail' = 'auto' | 'fused' | 'bf16x3' | 'unfused', 'attention' = 'auto' | 'f16x2' | 'bf16x3', 'linear' = 'auto' | 'f16x2' | 'bf16x3';
        the A/B switches 'conv_swizzle' = 'on' | 'off', 'qkv_amax' = 'epilogue' | 'kernel', 'sinkhorn_group' = 'auto' | 1 | 2 | 4,
        'sinkhorn_prefetch' = 'auto' | 'off' | 'on' (include/imx.h)."""
        self._check(self.lib.imx_set_option(self.handle, key.encode(), str(value).encode()))
        return self

    def get_option(self, key):
        return self.lib.imx_get_option(self.handle, key.encode()).decode()

    # ------------------------------------------------------------------ debug / timing
    def set_debug(self, on=True):
        self._check(self.lib.imx_set_debug(self.handle, int(on)))

    def fetch(self, name):
        shape = (ctypes.c_int64 * 4)()
        nd = ctypes.c_int(0)
        self._check(self.lib.imx_debug_fetch(self.handle, name.encode(), None, 0, shape, ctypes.byref(nd)))
        shp = tuple(shape[i] for i in range(nd.value))
        out = np.empty(shp, dtype=np.float32)
        self._check(self.lib.imx_debug_fetch(self.handle, name.encode(), out.ctypes.data_as(ctypes.c_void_p),
                                             out.size, shape, ctypes.byref(nd)))
        return out

    def set_timing(self, on=True):
        self._check(self.lib.imx_set_timing(self.handle, int(on)))

    def timing_reset(self):
        self._check(self.lib.imx_timing_reset(self.handle))

    def timing_report(self, forms=False):
        """[(name, launches, total_ms)] of the instrumented launches; forms=True appends the kernel form of each row
        ('gemm_x3:bf16x3', ...) as the library reports it (imx_timing_form)."""
        n = self.lib.imx_timing_report(self.handle, -1, None, None, None)
        if n < 0:
            self._check(n)
        rows = []
        for i in range(n):
            name, cnt, ms = ctypes.c_char_p(), ctypes.c_int64(), ctypes.c_double()
            self._check(self.lib.imx_timing_report(self.handle, i, ctypes.byref(name), ctypes.byref(cnt), ctypes.byref(ms)))
            row = (name.value.decode(), cnt.value, ms.value)
            rows.append(row + (self.lib.imx_timing_form(self.handle, i).decode(),) if forms else row)
        return rows
