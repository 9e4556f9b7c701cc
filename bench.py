#!/usr/bin/env python
"""bench.py — image-pairs/sec of the SuperPoint+SuperGlue hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is one pass of the hot path over one batch of synthetic pairs already resident in HBM:
B pairs per GPU (weak scaling; pair i of the job goes to rank i % world) through the fused C-ABI
call imx_match_pairs (SuperPoint on both images, SuperGlue, match extraction), followed — when
world > 1 — by the one collective of the path, an RCCL gather of the fixed-size match records to rank 0.
Workload = BASELINE.json configs[2]/[3] shape ("C3"): 640x480 grayscale, d=128, 1024 keypoints,
30 Sinkhorn iterations, synthetic BN-calibrated weights (no trained weights exist: LFS pointers).

Rank 0 prints ONE JSON line (contract in the task statement) including
  "roofline":     the dominant kernel against the gfx950 peak, from per-launch HIP events on the launch stream
                  over a second, instrumented pass of the same K steps.  `achieved`/`frac` are the rate the
                  matrix cores actually EXECUTE (never above 1); where a kernel executes fewer multiplies than
                  the reference's direct-form count (Winograd F(2x4,3x3): 3x fewer) the algorithmic rate is
                  reported next to it under `algorithmic`.  `executed_pair_frac` is the whole-pair view.
                  Which matrix pipe a kernel is priced against comes from the LIBRARY (imx_timing_form: the kernel form
                  that actually ran), not from a copy of its dispatch rule.  `traffic` comes from rocprofv3 PMC passes
                  (profiles/r*_pmc_traffic.json, taken at this run's 64 pairs per step) and is dropped when that file was
                  measured on a different library build than the one being timed;
  "parity_in_run": the first 32 pairs of the timed batch are the UNSELECTED sweep seeds 1000..1031 whose reference outputs
                  are committed in tests/golden/sweep_c3.npz; after the timed region the keypoints and match indices of the
                  LAST timed step are compared with them (tests/util.py: sweep_compare_end_to_end -- fixture only, no oracle):
                  the number the driver times and the parity evidence meet in one process;
  "parity_in_run_strict": the same resident pairs pushed once more (untimed) through the same call with the second synthetic
                  SuperGlue weight set loaded ("t": trained-model-like score statistics, synth.SGT_GAINS) and compared with
                  the reference's committed outputs on it (tests/golden/strict_c3.npz): keypoint sets and match indices
                  identical (exact ties of the reference's own Z and scores within 1e-4 of the threshold excepted, counted),
                  matching scores and samples of gnn17 / scores_in / Z against the north_star tolerance;
  "step_ms":      min / median / max of the K timed steps (HIP events around each step), "clocks": sclk / socket power sampled
                  from sysfs during the timed region, "roofline.frac_at_clock": the fraction against the peak at that clock;
  "c2":           SuperPoint-only images/s (BASELINE configs[1]) on the same resident images;
  "c5":           a short leg on the stress configuration (BASELINE configs[4]) with its own parity_in_run;
  N > 1 lines:    "compute_ms_per_step" / "gather_ms_per_step" (HIP events either side of the collective, max over ranks),
                  "backend" = what the process group really runs on, "world_checked";
  "gather_ms":    the path's one collective (RCCL gather of the step's match records) timed at world 1 as the N>1 baseline;
  "latency_b1_ms":          Matching.forward on ONE pair (BASELINE configs[2]), median of 50 synchronised calls;
  "pcie_inclusive_pairs_s": the same step fed from uint8 frames in pinned host memory (never `value`);
  "cpu_baseline": the oracle (CPU restatement of the reference, torch CPU ops) timed on this
                  host's cores on a bounded sample of the same workload, best of a thread-count sweep.
"""
import argparse
import json
import os
import sys
import time

# the host driver of this pool supports dmabuf IPC only: without this RCCL's cross-process buffer registration fails with
# `hipIpcGetMemHandle: invalid argument` (exported by the image already; kept here for a launch from a bare environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np                      # noqa: E402
import torch                            # noqa: E402
import torch.distributed as dist        # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from image_matching_amd import shard, synth                                    # noqa: E402
from image_matching_amd.superglue.models.matching_test import Matching         # noqa: E402

# seed0: pair i of the job is synth_pair(seed0 + i) -- the first pairs are the unselected sweep seeds of tests/golden/<sweep>
WORKLOADS = {
    "c3": dict(H=480, W=640, d=128, K=1024, seed0=1000, sweep="sweep_c3.npz", name="C3 SuperPoint+SuperGlue 640x480 d=128 1024 kpts 30 Sinkhorn iters"),
    "c5": dict(H=960, W=1280, d=256, K=2048, seed0=2000, sweep="sweep_c5.npz", name="C5 SuperPoint+SuperGlue 1280x960 d=256 2048 kpts 100 Sinkhorn iters"),
    "c2": dict(H=480, W=640, d=128, K=1024, seed0=1000, sweep=None, name="C2 SuperPoint-only 640x480 d=128 NMS + top-1024 keypoints"),
}
PEAK_MFMA_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA dense peak
PEAK_MFMA_BF16_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA dense peak (no sparsity)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak


def algorithmic_work(B, H, W, d, K, kenc, iters, n_layers=18):
    """Per-launch algorithmic work of every kernel name of one step (B pairs = 2B images):
    ('mfma', FLOPs) for matrix kernels, ('hbm', bytes) for streaming kernels.  FLOPs = 2 x MACs of
    the reference's dense ops; bytes = compulsory reads+writes of the stage."""
    I = 2 * B
    H2, W2, H4, W4, Hc, Wc = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8
    R = 2 * B * K
    w = {
        # algorithmic = the reference's direct-convolution FLOPs; the Winograd F(2x2,3x3)
        # kernels execute 2.25x fewer multiplies, so "achieved" can exceed the executed-FLOP rate
        "conv1ab_pool": ("mfma", 2.0 * I * H * W * (9 * 64 + 576 * 64)),
        "conv2a": ("mfma", 2.0 * I * H2 * W2 * 576 * 64),
        "conv2b_pool": ("mfma", 2.0 * I * H2 * W2 * 576 * 64),
        "conv3a": ("mfma", 2.0 * I * H4 * W4 * 576 * 128),
        "conv3b_pool": ("mfma", 2.0 * I * H4 * W4 * 1152 * 128),
        "conv4a": ("mfma", 2.0 * I * Hc * Wc * 1152 * 128),
        "conv4b": ("mfma", 2.0 * I * Hc * Wc * 1152 * 128),
        "convPaDa": ("mfma", 2.0 * I * Hc * Wc * 1152 * 512),
        "convPb": ("mfma", 2.0 * I * Hc * Wc * 256 * 65),
        "convDb": ("mfma", 2.0 * I * Hc * Wc * 256 * d),
        "softmax_shuffle": ("hbm", 4.0 * I * Hc * Wc * (65 + 64)),
        "nms": ("hbm", 4.0 * I * 2 * H * W),
        "keypoints": ("hbm", 4.0 * I * H * W),
        "describe": ("hbm", 4.0 * I * K * (4 * d + d + 3)),
        "sg_prologue": ("hbm", 4.0 * 2 * R * d / 2 + 4.0 * R * (3 + kenc[0])),     # descriptor gather + kenc layer 0, both sides
        "qkv_proj": ("mfma", 2.0 * R * d * 3 * d),
        "attention": ("mfma", 2.0 * 2 * B * 2 * K * K * d),
        # attn.merge (d->d) is folded into mlp.0's weights at load; its reference FLOPs stay in the count
        "gnn_mlp1": ("mfma", 2.0 * R * (2 * d * 2 * d + d * d)),
        "gnn_mlp2": ("mfma", 2.0 * R * 2 * d * d),
        "final_proj": ("mfma", 2.0 * R * d * d),
        # the fused layer tail (gnn_tail_x3: mlp.0' -> mlp.3 -> the next layer's q|k|v, or final_proj after the last layer), averaged over
        # the n_layers launches of a step: n_layers x (mlp1 + mlp2) + (n_layers - 1) x q|k|v + final_proj
        "gnn_tail": ("mfma", (n_layers * (2.0 * R * (2 * d * 2 * d + d * d) + 2.0 * R * 2 * d * d) + (n_layers - 1) * 2.0 * R * d * 3 * d + 2.0 * R * d * d) / n_layers),
        "score_gemm": ("mfma", 2.0 * B * K * K * d),
        "sinkhorn": ("hbm", 4.0 * B * K * K * iters),           # one launch group = all iterations; the slab form reads S ONCE per iteration
        "matches": ("hbm", 4.0 * B * K * K * 2),
    }
    ch = list(kenc) + [d]
    w["kenc"] = ("mfma", 2.0 * R * sum(ch[i] * ch[i + 1] for i in range(len(ch) - 1)) / (len(ch) - 1))
    return w


def executed_work(B, H, W, d, K, kenc, iters, n_layers=18):
    """FLOPs the matrix cores EXECUTE per launch (MFMA kernels only), as the kernels are written:
      3x3 layers     Winograd F(2x4,3x3): 24 multiplies per 8 outputs (direct: 72) over whole 8x16-pixel tiles;
      conv1a         inside the fused first layer, as a GEMM on the matrix cores over each tile's halo: 144
                     v_mfma_f32_16x16x4 (2048 FLOP each) per 8x16-pixel tile (36 per wave, conv1ab_wino24.hip);
      1x1 convs      output columns padded to 64 (convPb: 65 -> 128);
      gnn_mlp1       attn.merge is folded into mlp.0's weights at load: its d*d product is not executed;
      attention      as counted (N is a multiple of the 128-query block at C3/C5)."""
    I = 2 * B
    R = 2 * B * K
    up = lambda a, b: (a + b - 1) // b
    tiles = lambda h, w: up(h, 8) * up(w, 16)
    conv = lambda h, w, cin, cout: 2.0 * I * tiles(h, w) * 128 * 9 * cin * cout / 3.0
    H2, W2, H4, W4, Hc, Wc = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8
    pad64 = lambda n: up(n, 64) * 64
    ex = {
        "conv1ab_pool": conv(H, W, 64, 64) + I * tiles(H, W) * 144 * 2048.0,
        "conv2a": conv(H2, W2, 64, 64), "conv2b_pool": conv(H2, W2, 64, 64),
        "conv3a": conv(H4, W4, 64, 128), "conv3b_pool": conv(H4, W4, 128, 128),
        "conv4a": conv(Hc, Wc, 128, 128), "conv4b": conv(Hc, Wc, 128, 128), "convPaDa": conv(Hc, Wc, 128, 512),
        "convPb": 2.0 * I * Hc * Wc * 256 * pad64(65), "convDb": 2.0 * I * Hc * Wc * 256 * pad64(d),
        "qkv_proj": 2.0 * R * d * 3 * d, "attention": 2.0 * 2 * B * 2 * K * K * d,
        "gnn_mlp1": 2.0 * R * 2 * d * 2 * d, "gnn_mlp2": 2.0 * R * 2 * d * d,
        "final_proj": 2.0 * R * d * d, "score_gemm": 2.0 * B * K * K * d,
        "gnn_tail": (n_layers * (2.0 * R * 2 * d * 2 * d + 2.0 * R * 2 * d * d) + (n_layers - 1) * 2.0 * R * d * 3 * d + 2.0 * R * d * d) / n_layers,
    }
    ch = list(kenc) + [d]
    ex["kenc"] = 2.0 * R * sum(ch[i] * ch[i + 1] for i in range(len(ch) - 1)) / (len(ch) - 1)
    return ex


def build_matching(wl, device):
    d, K = wl["d"], wl["K"]
    kenc, iters, thr = synth.SG_CONFIGS[d]
    cfg = {"superpoint": {"weights": None, "descriptor_dim": d, "nms_radius": 4, "keypoint_threshold": 0.005,
                          "max_keypoints": K},
           "superglue": {"weights": None, "descriptor_dim": d, "keypoint_encoder": kenc,
                         "sinkhorn_iterations": iters, "match_threshold": thr}}
    m = Matching(cfg).eval().to(device)
    sd_sp = {k: torch.from_numpy(np.array(v)) for k, v in synth.make_superpoint_state_dict(d).items()}
    sd_sg = {k: torch.from_numpy(np.array(v)) for k, v in synth.make_superglue_state_dict(d).items()}
    m.superpoint.load_state_dict(sd_sp)
    m.superglue.load_state_dict(sd_sg)
    return m, cfg, sd_sp, sd_sg


def cpu_baseline_worker(workload, n_pairs):
    """Child process of cpu_baseline(): the oracle on `n_pairs` + 1 pairs with the thread count fixed by OMP_NUM_THREADS
    in the environment (set before torch is imported); prints the per-pair seconds as JSON.  Never touches the GPU."""
    from oracle import matching_ref            # checker code: used here only as the CPU baseline leg
    wl = WORKLOADS[workload]
    d, K = wl["d"], wl["K"]
    kenc, iters, thr = synth.SG_CONFIGS[d]
    cfg = {"superpoint": {"weights": None, "descriptor_dim": d, "nms_radius": 4, "keypoint_threshold": 0.005, "max_keypoints": K},
           "superglue": {"weights": None, "descriptor_dim": d, "keypoint_encoder": kenc, "sinkhorn_iterations": iters,
                         "match_threshold": thr}}
    sd_sp = {k: torch.from_numpy(np.array(v)) for k, v in synth.make_superpoint_state_dict(d).items()}
    sd_sg = {k: torch.from_numpy(np.array(v)) for k, v in synth.make_superglue_state_dict(d).items()}
    times = []
    for i in range(n_pairs + 1):
        im0, im1 = synth.synth_pair(1000 + i, wl["H"], wl["W"])
        x0, x1 = torch.from_numpy(im0)[None, None], torch.from_numpy(im1)[None, None]
        t = time.perf_counter()
        matching_ref.matching_forward({"image0": x0, "image1": x1}, sd_sp, sd_sg, cfg)
        times.append(time.perf_counter() - t)
        print(json.dumps({"threads": torch.get_num_threads(), "times": times}), flush=True)      # last line wins (partial on timeout)


def cpu_baseline(workload, budget_s=60.0, n_pairs=5):
    """The oracle (port of the reference's PyTorch CPU forward) on this host's cores.  More threads are not better for
    this model (128 threads gave 0.32 pairs/s in round 1, slower than 8 threads in the survey container), so the thread
    count is swept and the best is reported.  Each count runs in its own child process (OMP_NUM_THREADS set before torch
    loads, hard timeout): 1 warm-up pair + the median of up to `n_pairs` pairs (>= 5; a count that times out
    before 5 is reported with the pairs it finished); ~`budget_s` seconds in total."""
    import subprocess
    try:
        avail = len(os.sched_getaffinity(0))       # cores this process may run on (cgroup/affinity aware)
    except AttributeError:
        avail = os.cpu_count() or 1
    counts = sorted({c for c in (8, 16, 32, 64) if c <= avail} or {avail})
    per = budget_s / len(counts)
    sweep, best = {}, None
    for c in counts:
        env = dict(os.environ, OMP_NUM_THREADS=str(c), MKL_NUM_THREADS=str(c), HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(n_pairs), "--workload", workload]
        try:
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=per + 15).stdout
        except subprocess.TimeoutExpired as e:
            out = (e.stdout or b"").decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        times = json.loads(lines[-1])["times"][1:] if lines else []          # first pair = warm-up
        if not times:
            sweep[str(c)] = None
            continue
        med = float(np.median(times))
        sweep[str(c)] = round(1.0 / med, 4)
        if best is None or med < best[1]:
            best = (c, med, len(times))
    if best is None:
        return {"value": None, "unit": "image-pairs/s", "cores": 0, "kind": "port", "sample": f"no thread count finished: {sweep}"}
    c, med, n = best
    wl = WORKLOADS[workload]
    return {"value": round(1.0 / med, 4), "unit": "image-pairs/s", "cores": c, "kind": "port",
            "sample": f"best of a thread sweep {sweep} (pairs/s by OMP thread count, one child process each; {avail} cores "
                      f"available); median of {n} pairs after 1 warm-up, {wl['H']}x{wl['W']}, torch {torch.__version__} CPU"}


def latency_b1(matching, wl, device, n=50):
    """BASELINE configs[2] ("single pair"): the drop-in Matching.forward on ONE pair, host-synchronised per call."""
    im0, im1 = synth.synth_pair(7, wl["H"], wl["W"])
    data = {"image0": torch.from_numpy(im0)[None, None].to(device), "image1": torch.from_numpy(im1)[None, None].to(device)}
    for _ in range(5):
        matching(data)
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        matching(data)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return round(1e3 * float(np.median(ts)), 4)


def pcie_inclusive(matching, wl, B, steps=6, scale=0.5):
    """The step fed across PCIe: decoded uint8 frames (2x the network resolution, resize_scale 0.5 as in
    datasets/SSHIDataset.py:19-27) sit in the pinned staging buffers of IngestPipeline; per step they are copied
    host->device on a side stream, resized + /255 on the GPU (imx_ingest_resize_u8) and matched.  Reported, never `value`."""
    from image_matching_amd import hostops
    from image_matching_amd.ingest import IngestPipeline
    H, W = wl["H"], wl["W"]
    Hs, Ws = int(round(H / scale)), int(round(W / scale))
    eng = matching._shared.get_engine([0, 1])
    pipes = [IngestPipeline(eng, B, (Hs, Ws), (H, W)) for _ in range(2)]
    frames = [[], []]
    for i in range(min(B, 8)):                    # 8 distinct pairs, tiled over the batch (host-side synthesis is slow)
        im = synth.synth_pair(2000 + i, H, W)
        for s in range(2):
            frames[s].append(hostops.resize_linear_u8((im[s] * 255).astype(np.uint8), (Ws, Hs)))
    for s in range(2):
        stack = torch.from_numpy(np.stack([frames[s][i % len(frames[s])] for i in range(B)]))
        for slot in pipes[s].slots:
            slot["pinned"].copy_(stack)

    def ship():
        pipes[0].staging(), pipes[1].staging()
        return pipes[0].submit_staged(B), pipes[1].submit_staged(B)

    def run(n):
        t = ship()
        for k in range(n):             # match(k) is enqueued before batch k+1 is shipped: the copy overlaps the match
            out = matching.match_batch(pipes[0].take(t[0]), pipes[1].take(t[1]))
            pipes[0].release(t[0]), pipes[1].release(t[1])
            t = ship() if k + 1 < n else None
        torch.cuda.synchronize()
        return out
    run(2)
    t0 = time.perf_counter()
    out = run(steps)
    dt = time.perf_counter() - t0
    assert int((out["matches0"] > -1).sum()) > 0
    return {"value": round(B * steps / dt, 3), "unit": "image-pairs/s", "frame": [Hs, Ws], "resize_scale": scale,
            "pairs_per_step": B, "steps": steps, "h2d_GBps": round(2 * B * Hs * Ws * steps / dt / 1e9, 3),
            "note": "uint8 frames already in pinned staging buffers -> async H2D + GPU resize/255 -> imx_match_pairs"}


class ClockSampler:
    """sclk (MHz) and socket power (W) of this process's GPU from sysfs (hwmon freq1_input / power1_input of the amdgpu card),
    sampled by a child process every 20 ms while the timed region runs.  Reported, never used to scale `value`."""

    def __init__(self, device):
        import glob
        self.files, self.samples, self._stop, self._thread = None, [], False, None
        want = None
        try:
            pr = torch.cuda.get_device_properties(device)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        except Exception:       # noqa: BLE001 -- older torch: fall back to the first card that has the files
            pass
        cands = []
        for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")):
            dev = os.path.realpath(os.path.join(os.path.dirname(f), "..", ".."))
            cands.append((os.path.basename(dev), f, os.path.join(os.path.dirname(f), "power1_input")))
        pick = [c for c in cands if want and c[0].startswith(want)] or cands[:1]
        if pick:
            self.files = pick[0]

    def _read(self):
        try:
            with open(self.files[1]) as fh:
                mhz = int(fh.read()) / 1e6
            with open(self.files[2]) as fh:
                watts = int(fh.read()) / 1e6
            return mhz, watts
        except (OSError, ValueError):
            return None

    # (a CHILD process does the polling: a sampling thread in this process would compete for the GIL with the thread that
    # launches the timed steps -- ADVICE r4)
    _CHILD = ("import sys, time, json\n"
              "f, p = sys.argv[1], sys.argv[2]\n"
              "import select\n"
              "out = []\n"
              "while not select.select([sys.stdin], [], [], 0.02)[0]:\n"
              "    try:\n"
              "        out.append((int(open(f).read()) / 1e6, int(open(p).read()) / 1e6))\n"
              "    except (OSError, ValueError):\n"
              "        pass\n"
              "print(json.dumps(out))\n")

    def start(self):
        if not self.files:
            return
        import subprocess
        self.samples = []
        self._thread = subprocess.Popen([sys.executable, "-c", self._CHILD, self.files[1], self.files[2]], stdin=subprocess.PIPE,
                                        stdout=subprocess.PIPE, text=True)

    def stop(self):
        if self._thread:
            try:
                out, _ = self._thread.communicate("stop\n", timeout=5)
                self.samples = [tuple(x) for x in json.loads(out)]
            except Exception as e:  # noqa: BLE001 -- the sampler is a report, never a reason to fail the bench
                self._thread.kill()
                try:
                    self._thread.wait(timeout=5)        # (reap it: a killed child left un-waited is a zombie -- ADVICE r5)
                except Exception:   # noqa: BLE001
                    pass
                print(f"bench: clock sampler lost its samples ({type(e).__name__}: {e})", file=sys.stderr)
            self._thread = None

    def report(self):
        if not self.samples:
            return {"sclk_mhz": None, "note": "no sysfs hwmon files for this GPU"}
        mhz, w = np.array([a for a, _ in self.samples]), np.array([b for _, b in self.samples])
        return {"sclk_mhz": {"min": round(float(mhz.min())), "median": round(float(np.median(mhz))), "max": round(float(mhz.max()))},
                "socket_power_w": {"median": round(float(np.median(w))), "max": round(float(w.max()))}, "samples": len(self.samples),
                "source": f"sysfs {self.files[0]} hwmon freq1_input / power1_input, every 20 ms inside the timed region"}


def plan(world, rank, B):
    """Who does what at `world` ranks: this rank's pair ids of the global batch (pair i -> rank i % world) and the padded record rows
    every rank contributes to the gather.  Pure arithmetic (tests/test_host.py drives it through --plan-only without a GPU)."""
    return {"world": world, "rank": rank, "pairs_per_gpu": B, "global_pairs": world * B,
            "pair_ids": shard.shard_indices(world * B, rank, world), "rows_per_rank": shard.shard_rows(world * B, world)}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def resident_inputs(wl, pair_ids, device):
    """This rank's pairs of the global batch, resident in HBM: pair i of the job = synth_pair(seed0 + i)."""
    ims = [synth.synth_pair(wl["seed0"] + pid, wl["H"], wl["W"]) for pid in pair_ids]
    img0 = torch.from_numpy(np.stack([p[0] for p in ims]))[:, None].to(device)
    img1 = torch.from_numpy(np.stack([p[1] for p in ims]))[:, None].to(device)
    return img0, img1


def parity_in_run(wl, pair_ids, out):
    """The outputs of the LAST timed step against the reference's outputs on the unselected sweep seeds (committed fixture; the
    comparison rule is the GPU tests': keypoint sets may differ only on a top-k boundary tie, a differing match must sit on a
    reference margin below tau = 2e-3 in Z units).  Pairs whose job index is below the sweep length are checked."""
    from tests import util                                  # fixture reader + comparison rule (no oracle involved)
    g = util.golden(wl["sweep"])
    n = len(g["seeds"])
    assert int(g["seeds"][0]) == wl["seed0"]
    k0, k1, m0 = out["keypoints0"].cpu().numpy(), out["keypoints1"].cpu().numpy(), out["matches0"].cpu().numpy()
    res = [util.sweep_compare_end_to_end(g, pid, k0[b], k1[b], m0[b]) for b, pid in enumerate(pair_ids) if pid < n]
    rep = {"pairs": len(res), "fixture": "tests/golden/" + wl["sweep"],
           "keypoint_set_mismatch_images": sum(r["kp_diff_images"] for r in res),
           "keypoint_set_mismatch_beyond_topk_ties": sum(len(r["kp_bad"]) for r in res),
           "reference_matches": sum(r["n_ref"] for r in res), "index_mismatches": sum(r["diff"] for r in res),
           "unexplained": sum(len(r["unexplained"]) for r in res),
           "rule": "keypoint sets equal unless the top-k boundary gap < 2e-5; a differing match index must lie on a reference margin < 2e-3 (Z units)"}
    assert rep["unexplained"] == 0 and rep["keypoint_set_mismatch_beyond_topk_ties"] == 0, f"parity_in_run failed: {rep}"
    return rep


def parity_in_run_strict(matching, wl, pair_ids, img0, img1, sd_sg_default):
    """One more pass of the SAME resident pairs through the SAME call, untimed, with the "t" SuperGlue weight set (trained-model-like
    score statistics) and debug taps on, against the reference's committed outputs on it (tests/golden/strict_*.npz; fixture only,
    no oracle).  The default weights are restored afterwards."""
    from tests import util
    g = util.golden("strict_" + wl["sweep"].split("_", 1)[1])
    assert int(g["seeds"][0]) == wl["seed0"]
    d = wl["d"]
    sd_t = {k: torch.from_numpy(np.array(v)) for k, v in synth.make_superglue_state_dict(d, variant="t").items()}
    matching.superglue.load_state_dict(sd_t)
    eng = matching._shared.get_engine([0, 1])
    eng.set_debug(True)
    try:
        out = matching.match_batch(img0, img1)
        torch.cuda.synchronize()
        n = len(g["seeds"])
        rep = util.strict_compare_batch(g, out, eng, len(pair_ids), float(sd_t["bin_score"]), float(synth.SG_CONFIGS[d][2]),
                                        seed_idx=[pid if pid < n else -1 for pid in pair_ids])
    finally:
        eng.set_debug(False)
        matching.superglue.load_state_dict(sd_sg_default)
        matching._shared.get_engine([0, 1])
    rep["worst_tolerance_used"] = {k: round(v, 3) for k, v in rep["worst_tolerance_used"].items()}
    rep["fixture"] = "tests/golden/strict_" + wl["sweep"].split("_", 1)[1]
    rep["weights"] = "SuperGlue set 't' (synth.SGT_GAINS): scores_in std ~5, bin_score = mean + 2 sigma; SuperPoint unchanged"
    rep["rule"] = ("keypoint sets identical; match indices identical except exact ties of the reference's own fp32 Z and rows whose reference score is within 1e-4 of "
                   "match_threshold (both counted); matching scores and the reference's samples of gnn17 / scores_in / Z: images in, so SuperPoint's within-tolerance "
                   "differences are amplified by the GNN -- scores_in / Z asserted against the fixture's float64 samples at 1e-4 + 1e-4|f64| + 2.5 x the reference's own "
                   "fp32-vs-float64 envelope of that seed, gnn17 and the matching scores at 3x, all counted at 1x of 1e-4 + 1e-4|ref| with every outlier listed "
                   "beside the reference's own distance from float64 at that element (the SuperGlue stage alone is held to 1x in tests/test_gpu_strict.py); "
                   "a separate untimed pass with the 't' SuperGlue weights on the resident pairs, not the timed step itself")
    return rep


def c2_leg(matching, img01, steps):
    """BASELINE configs[1]: SuperPoint only (detect + describe, top-K keypoints) on the 2B resident images of the step."""
    eng = matching._shared.get_engine([0])
    for _ in range(2):
        eng.superpoint_batch(img01)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        kp, sc, ds, cnt = eng.superpoint_batch(img01)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert (cnt.cpu().numpy() == matching._shared.engine.max_keypoints).all()
    return {"workload": WORKLOADS["c2"]["name"], "images_s": round(img01.shape[0] * steps / dt, 2), "ms_per_step": round(1e3 * dt / steps, 4),
            "images_per_step": int(img01.shape[0]), "steps": steps}


PIPES = {"f32": ("fp32 MFMA", PEAK_MFMA_F32_TFLOPS, 1.0),
         "bf16x3": ("bf16 MFMA, six bf16 term products per fp32 product", PEAK_MFMA_BF16_TFLOPS, 6.0),
         # attention_x3.hip's two-plane fp16 form: operands as h + m in fp16 (22 bits), products (h,h) (h,m) (m,h); the fp16 MFMA's dense
         # peak is the bf16 one's
         "f16x2": ("fp16 MFMA, three fp16 term products per fp32 product", PEAK_MFMA_BF16_TFLOPS, 3.0)}
SPLIT_PIPES = ("bf16x3", "f16x2")


def roofline_block(rows, wl, B, steps, dt, lib_build, workload):
    """rows: Engine.timing_report(forms=True) of `steps` instrumented steps: (name, launches, total_ms, form) with form =
    '<kernel family>:<pipe>' as the LIBRARY reports it (imx_timing_form)."""
    H, W, d, K = wl["H"], wl["W"], wl["d"], wl["K"]
    kenc, iters, _ = synth.SG_CONFIGS[d]
    work = algorithmic_work(B, H, W, d, K, kenc, iters)
    exe = executed_work(B, H, W, d, K, kenc, iters)
    by = {}                                # name -> [launches, ms, {form: ms}]
    for name, launches, ms, form in rows:
        e = by.setdefault(name, [0, 0.0, {}])
        e[0] += launches
        e[1] += ms
        e[2][form] = e[2].get(form, 0.0) + ms
    form_of = {n: max(e[2], key=e[2].get) for n, e in by.items()}          # a name's dominant form (one form per name in practice)
    pipe_of = {n: (f.split(":")[1] if ":" in f else "") for n, f in form_of.items()}
    for n, f in form_of.items():           # the direct-form convolution executes the algorithmic FLOPs
        if f.startswith("conv3x3_direct") and n in exe:
            exe[n] = work[n][1]
    pipe_peak = lambda k: PIPES.get(pipe_of.get(k), PIPES["f32"])[1]
    pipe_flops = lambda k: exe[k] * PIPES.get(pipe_of.get(k), PIPES["f32"])[2]
    tot_ms = sum(e[1] for e in by.values())
    name = max(by, key=lambda n: by[n][1])
    launches, ms = by[name][0], by[name][1]
    bound, units = work[name]
    avg_s = ms / launches * 1e-3
    if bound == "mfma":
        algorithmic, peak, unit = units / avg_s / 1e12, pipe_peak(name), "TFLOP/s"
        achieved = pipe_flops(name) / avg_s / 1e12      # what the matrix cores execute: the rate the MFMA roofline bounds
    else:
        algorithmic, peak, unit = units / avg_s / 1e9, PEAK_HBM_GBS, "GB/s"
        achieved = algorithmic
    traffic, traffic_note, pmc = None, None, None   # HBM bytes per launch from rocprofv3 PMC passes (tools/pmc_traffic.py -> profiles/)
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True)      # newest round first
        pmc = None
        for path in cands:
            with open(path) as fh:
                pmc = json.load(fh)
            if pmc.get("build") == lib_build:
                break
        if pmc is None:
            raise OSError("no traffic file")
        if pmc.get("build") != lib_build:
            traffic_note = f"dropped: PMC passes were taken on build {pmc.get('build')!r}, this run is {lib_build!r}"
            pmc = None
        elif workload != "c3" or pmc["pairs_per_gpu"] != B:
            traffic_note = f"dropped: PMC passes were taken at {pmc['pairs_per_gpu']} C3 pairs per step, this run is {workload} at {B}"
            pmc = None
        elif name in pmc["kernels"]:
            traffic = pmc["kernels"][name]["traffic_bytes"]
    except (OSError, ValueError, KeyError):
        traffic_note = "no PMC traffic file for this round"
    rf = {"bound": bound, "achieved": round(achieved, 3), "peak": peak, "unit": unit, "frac": round(achieved / peak, 4), "traffic": traffic,
          "kernel": name, "form": form_of[name], "avg_launch_ms": round(ms / launches, 4), "share_of_gpu_time": round(ms / tot_ms, 4), "build": lib_build}
    if traffic_note:
        rf["traffic_note"] = traffic_note
    if bound == "mfma":
        rf["pipe"] = PIPES.get(pipe_of[name], PIPES["f32"])[0]
    if abs(algorithmic - achieved) > 1e-9 and (pipe_of[name] not in SPLIT_PIPES or name.startswith("conv")):
        rf["algorithmic"] = {
            "rate": round(algorithmic, 3), "ratio_to_peak": round(algorithmic / peak, 4), "ratio_to_fp32_mfma_peak": round(algorithmic / PEAK_MFMA_F32_TFLOPS, 4),
            "note": "reference direct-form FLOPs / launch time; the kernel executes fewer multiplies (Winograd F(2x4,3x3)), "
                    "so this ratio may exceed 1 and is NOT a utilisation"}
    if bound == "mfma" and pipe_of[name] == "f16x2" and name.startswith("conv"):
        rf["frac_note"] = ("round 4 moved this kernel's products from the fp32 MFMA (157.3 TFLOP/s) to three fp16 plane products on the 16-bit pipe "
                           "(2500 TFLOP/s): `achieved` counts the executed plane products against THAT peak, so `frac` fell while the launch got shorter "
                           "(r04_v9: 10.4 ms at 0.647 of the fp32 peak); what the launch's cycles go to is `limiter` (counters), and why the pair form "
                           "is DESIGN.md section 4")
    # what limits the dominant kernel, FROM COUNTERS: tools/gpu_pmc_limiter.sh -> profiles/r*_pmc_limiter.json, taken on
    # this build (same rule as `traffic`); rounds 1-4 computed this figure from a byte count (tiles x 393216 B of transformed weights)
    try:
        import glob
        lim = None
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_limiter.json")), reverse=True):
            with open(path) as fh:
                cand = json.load(fh)
            # build, workload and batch must all match (the first same-build file used to end the search even when it
            # belonged to another workload and a matching one existed)
            if cand.get("build") == lib_build and cand.get("workload", "c3") == workload and cand.get("pairs_per_gpu") == B:
                lim, lim_path = cand, path
                break
        if lim is None:
            rf["limiter_note"] = f"dropped: no profiles/r*_pmc_limiter.json was taken on this build for {workload} at {B} pairs per step"
        elif name in lim["kernels"]:
            k = lim["kernels"][name]
            rf["limiter"] = {
                "what": "share of the launch's cycles in which each unit is busy, rocprofv3 PMC passes on this build: the L1 (TCP) data path that "
                        "returns the transformed weights U to the registers (64-byte accesses against 64 B/clk per CU), the matrix pipes, the "
                        "vector ALU (input transform + fp16 split + epilogue), the LDS; `wait` = wave time at an s_waitcnt",
                "source": os.path.relpath(lim_path, ROOT),
                "l1_data_path": k.get("l1_data_path_frac"), "mfma_busy": k.get("mfma_busy"), "valu_active": k.get("valu_active_per_simd_cycle"),
                "lds_active": k.get("lds_active_per_simd_cycle"), "wait": k.get("wait_any_of_wave_cycles"),
                "l1_l2_read_bytes_per_launch": k.get("l1_l2_read_bytes"), "l1_l2_read_bytes_per_cu_cycle": k.get("l1_l2_read_bytes_per_cu_cycle"),
                "l2_read_latency_cycles": k.get("l2_read_latency_cycles"), "frac": k.get("l1_data_path_frac")}
            also = {n: {"l1_data_path": v.get("l1_data_path_frac"), "mfma_busy": v.get("mfma_busy"), "valu_active": v.get("valu_active_per_simd_cycle"),
                        "wait": v.get("wait_any_of_wave_cycles")} for n, v in lim["kernels"].items() if n != name}
            if also:
                rf["limiter"]["other_kernels"] = also
    except (OSError, ValueError, KeyError) as e:
        rf["limiter_note"] = f"no usable PMC limiter file: {e}"
    # whole-pair view: time the matrix pipes would need at their dense peaks (fp32 MFMA 157.3, bf16 MFMA 2500 TFLOP/s) / step time
    per_step = {n: e[0] / steps for n, e in by.items()}          # launches per step
    step_s = dt / steps
    exe_step = sum(u * per_step.get(k, 0.0) for k, u in exe.items())
    alg_step = sum(u * per_step.get(k, 0.0) for k, (bd, u) in work.items() if bd == "mfma")
    at_peak_s = sum(pipe_flops(k) * per_step.get(k, 0.0) / (pipe_peak(k) * 1e12) for k in exe)
    rf["executed_pair_frac"] = round(at_peak_s / step_s, 4)
    rf["fp32_equivalent_pair_tflops"] = round(exe_step / step_s / 1e12, 2)
    rf["bf16x3_kernels"] = sorted(k for k, v in pipe_of.items() if v == "bf16x3")
    rf["f16x2_kernels"] = sorted(k for k, v in pipe_of.items() if v == "f16x2")
    rf["algorithmic_pair_ratio"] = round(alg_step / step_s / 1e12 / PEAK_MFMA_F32_TFLOPS, 4)
    rf["timed_groups_per_step"] = int(round(sum(per_step.values())))      # one group = one RUN() of imx_api.cpp (the Sinkhorn group holds 60 kernel launches)
    kern = {}
    for n, (launches, ms, forms) in by.items():
        k = {"launches": launches, "ms_per_step": round(ms / steps, 4)}
        if form_of[n]:
            k["form"] = form_of[n] if len(forms) == 1 else sorted(forms)
        if n in exe:
            k["executed_frac"] = round(pipe_flops(n) / (ms / launches * 1e-3) / 1e12 / pipe_peak(n), 4)
            if pipe_of[n] in SPLIT_PIPES:
                k["fp32_equivalent_tflops"] = round(exe[n] / (ms / launches * 1e-3) / 1e12, 2)
        elif n in work:
            k["hbm_frac"] = round(work[n][1] / (ms / launches * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
        if pmc and n in pmc["kernels"]:
            k["traffic_bytes_per_launch"] = pmc["kernels"][n]["traffic_bytes"]
        kern[n] = k
    rf["kernels"] = kern
    return rf


def arithmetic_text(rf):
    x3 = rf.get("bf16x3_kernels", []) if rf else []
    h2 = rf.get("f16x2_kernels", []) if rf else []
    f32k = sorted(k for k, v in (rf.get("kernels", {}) if rf else {}).items() if str(v.get("form", "")).endswith(":f32"))
    return ("fp32 in, fp32 accumulate, fp32 out everywhere.  The 3x3 convolutions are Winograd F(2x4,3x3).  On the fp32 MFMA: "
            + (", ".join(f32k) if f32k else "nothing") + " and score_gemm.  "
            + (f"On the bf16 MFMA, each fp32 product carried as six bf16 term products (x = h + m + l exactly; error vs float64 below the "
               f"fp32 MFMA's: profiles/r02_mfma_bf16x3.txt), as reported by the library for this run: {', '.join(x3)}.  " if x3 else "")
            + (f"On the fp16 MFMA, each fp32 product carried as three fp16 term products of two-plane operands (x s = h + m, 22 bits, s a power of two "
               f"from the tensor's maximum; error vs float64 below the fp32 MFMA's: tools/ubench/attn_x3_bench.cpp, conv_h_bench.cpp): {', '.join(h2)}.  " if h2 else "")
            + "imx_set_option(h, 'mfma', 'f32') keeps every product on the fp32 MFMA (the parity tests hold both to the same bar).  "
              "On these default (heavy-tailed) weights |Z_hip - Z_reference| reaches ~1e-3 where the reference's own fp32 result is 2e-4..3e-3 from float64 "
              "(parity_in_run: match indices differ only on reference margins below that noise); on the trained-model-like weight set every element of "
              "gnn17 / scores_in / Z is within 1e-4 + 1e-4|ref| and every index is the reference's (parity_in_run_strict, DESIGN.md section 2)")


def c5_leg(device, steps=3, B=8):
    """BASELINE configs[4] (1280x960, d=256, 2048 kpts, 100 iterations) next to the headline: a short timed leg with its own
    roofline entry for its dominant kernel and its own parity_in_run (the 8 unselected C5 sweep seeds)."""
    wl = WORKLOADS["c5"]
    matching, *_ = build_matching(wl, device)
    pair_ids = list(range(B))
    img0, img1 = resident_inputs(wl, pair_ids, device)
    for _ in range(1):
        matching.match_batch(img0, img1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = matching.match_batch(img0, img1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng = matching._shared.engine
    eng.timing_reset()
    eng.set_timing(True)
    for _ in range(steps):
        matching.match_batch(img0, img1)
    rows = eng.timing_report(forms=True)
    eng.set_timing(False)
    eng.timing_reset()
    rf = roofline_block(rows, wl, B, steps, dt, eng.lib.imx_version().decode(), "c5")
    top = sorted(rf["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:6]
    return {"workload": wl["name"], "pairs_s": round(B * steps / dt, 3), "ms_per_step": round(1e3 * dt / steps, 3), "pairs_per_step": B, "steps": steps,
            "roofline": {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "form", "avg_launch_ms", "executed_pair_frac")},
            "top_kernels_ms_per_step": {k: v["ms_per_step"] for k, v in top},
            "parity_in_run": parity_in_run(wl, pair_ids, out)}


def gather_baseline(rec, n=20):
    """The path's one collective at world 1: a real RCCL gather of the step's record buffer to rank 0 (process group of one rank),
    so that the N>1 lines have a baseline for what the collective itself costs.  None (+ reason) if RCCL cannot initialise."""
    own = not dist.is_initialized()
    try:
        if own:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29519")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=rec.device)
        for _ in range(3):
            shard.gather_records(rec, force=True, check=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            shard.gather_records(rec, force=True, check=False)
        torch.cuda.synchronize()
        return round(1e3 * (time.perf_counter() - t0) / n, 4), f"world 1, {rec.numel() * 4} bytes per rank, mean of {n}"
    except Exception as e:            # noqa: BLE001 -- reported, never fatal: the headline does not depend on it
        return None, f"RCCL gather at world 1 unavailable: {type(e).__name__}: {e}"
    finally:
        if own and dist.is_initialized():
            dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs-per-gpu", type=int, default=64, help="pairs per step per GPU (weak scaling)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-pass", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip latency_b1_ms / pcie_inclusive_pairs_s / c5 / gather_ms")
    ap.add_argument("--cpu-baseline-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--plan-only", action="store_true", help="print this rank's share of the job (pair ids, record rows) as JSON and exit; no GPU needed")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args.workload, args.cpu_baseline_worker)
    if args.plan_only:
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        if world != args.gpus and world > 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        print(json.dumps(plan(world, rank, args.pairs_per_gpu)))
        return
    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner to the C stdout when a communicator is
    # created (flushed at exit, i.e. AFTER a Python print), so file descriptor 1 is pointed at stderr for the whole run and the
    # line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Bring-up hook for 1-GPU boxes: IMX_BENCH_BACKEND=gloo runs every rank on cuda:0 with a host-side gather, to exercise
    # the multi-rank control flow (barriers, max-over-ranks, who prints) where RCCL cannot run.  Never set by the driver.
    backend = os.environ.get("IMX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # IMX_BENCH_FORCE_PG=1: take the N>1 control flow (RCCL init, barriers, gather, max-over-ranks) at world 1 too, so
    # that a 1-GPU box can check the exact code the driver's N=2,4,8 launches run.  Never set by the driver.
    use_pg = world > 1 or os.environ.get("IMX_BENCH_FORCE_PG", "0") == "1"
    if use_pg:
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # the communicator really spans --gpus ranks (one per GPU), and every rank agrees on it
        assert dist.get_world_size() == world == max(args.gpus, 1) or os.environ.get("IMX_BENCH_FORCE_PG") == "1", (dist.get_world_size(), world, args.gpus)
        cnt = torch.ones(1, dtype=torch.int32, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(cnt)
        assert int(cnt.item()) == world, f"all-reduce of ones over the communicator gives {int(cnt.item())}, expected {world}"
        log(f"process group up: backend {dist.get_backend()} ({'RCCL' if backend == 'nccl' else 'host-side bring-up hook, NOT RCCL'}), world {dist.get_world_size()}, ranks counted {int(cnt.item())}")

    wl = WORKLOADS[args.workload]
    H, W, d, K = wl["H"], wl["W"], wl["d"], wl["K"]
    B = args.pairs_per_gpu
    matching, cfg, sd_sp, sd_sg = build_matching(wl, device)
    # this rank's pairs of the global batch (pair i -> rank i % world), resident in HBM
    pl = plan(world, rank, B)
    pair_ids, rows_per_rank = pl["pair_ids"], pl["rows_per_rank"]
    img0, img1 = resident_inputs(wl, pair_ids, device)
    pair_ids_dev = torch.tensor(pair_ids, dtype=torch.int32, device=device)

    sp_only = args.workload == "c2"
    img01 = torch.cat([img0, img1]) if sp_only else None

    def step(mid=None):
        if sp_only:                                  # C2: SuperPoint on 2B images, no SuperGlue / gather
            eng = matching._shared.get_engine([0])
            kp, sc, ds, cnt = eng.superpoint_batch(img01)
            if mid is not None:
                mid.record()
            return {"counts0": cnt[:B], "counts1": cnt[B:], "matches0": cnt.new_zeros(1) + 1}, torch.zeros(world * B, 1)
        out = matching.match_batch(img0, img1)
        rec = matching.pack_records(pair_ids_dev, out, pad_to=rows_per_rank)     # one kernel (imx_pack_records)
        if mid is not None:
            mid.record()                             # compute | collective boundary on the launch stream
        # the ONE collective of the path: RCCL gather of the records to rank 0 (shards equal by construction: check=False)
        return out, shard.gather_records(rec, force=use_pg, check=False)

    def barrier():
        if use_pg:
            dist.barrier()

    sampler = ClockSampler(device)
    stats = {}

    def timed(n):
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]      # step start | before the collective | step end
        barrier()
        torch.cuda.synchronize()
        sampler.start()
        t0 = time.perf_counter()
        for i in range(n):
            ev[i][0].record()
            out, rec = step(ev[i][1])
            ev[i][2].record()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        sampler.stop()
        per = np.array([e[0].elapsed_time(e[2]) for e in ev])
        comp = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        gath = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        if use_pg:
            t = torch.tensor([dt, comp, gath], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, comp, gath = (float(x) for x in t.tolist())
        stats.update(step_ms={"min": round(float(per.min()), 4), "median": round(float(np.median(per)), 4), "max": round(float(per.max()), 4),
                              "note": "HIP events around each of the K timed steps on the launch stream (rank 0)"},
                     compute_ms_per_step=round(comp, 4), gather_ms_per_step=round(gath, 4))
        return dt, out, rec

    log(f"inputs resident: {B} pairs/GPU, world {world}; warm-up x{args.warmup}")
    for _ in range(max(args.warmup, 0)):
        step()
    torch.cuda.synchronize()
    log("timed region")
    dt, out, rec = timed(args.steps)
    log(f"timed region done: {dt:.3f}s for {args.steps} steps")

    # harness checks: every image yields exactly K keypoints; the gather holds every pair once
    c0, c1 = out["counts0"].cpu().numpy(), out["counts1"].cpu().numpy()
    assert (c0 == K).all() and (c1 == K).all(), f"keypoint counts != {K}: {c0} {c1}"
    if not sp_only and rank == 0:
        assert rec.shape == (world * B, shard.record_width(K))
        assert sorted(shard.pair_ids_of(rec).cpu().tolist()) == list(range(world * B))
        if world == 1:
            assert torch.equal(rec, shard.pack_records(pair_ids, out)), "imx_pack_records differs from the host statement of the record layout"
    n_matches = int((out["matches0"] > -1).sum().item())
    assert n_matches > 0
    # parity of what was just timed: this rank's sweep-seed pairs of the LAST timed step against the reference's committed outputs
    parity = parity_in_run(wl, pair_ids, out) if (not sp_only and rank == 0) else None
    strict = None
    if not sp_only and rank == 0 and wl["sweep"] and not args.no_extras:
        log("parity_in_run_strict: the same pairs with the 't' SuperGlue weight set (untimed)")
        strict = parity_in_run_strict(matching, wl, pair_ids, img0, img1, sd_sg)

    total_pairs = world * B * args.steps
    value = (2 * total_pairs if sp_only else total_pairs) / dt
    line = {
        "metric": {"c3": "image-pairs/sec (640x480, 1024 kpts, 30 Sinkhorn iters)",
                   "c5": "image-pairs/sec (1280x960, 2048 kpts, 100 Sinkhorn iters)",
                   "c2": "images/sec (SuperPoint-only, 640x480, NMS + top-1024 kpts)"}[args.workload],
        "value": round(value, 3), "unit": "images/s" if sp_only else "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "compute_pipe": "fp32 products as exact-split plane products on the 16-bit matrix pipes (two fp16 planes x three products: convolutions, attention, GNN tails; three bf16 planes x six products: the other linear layers), fp32 accumulation",
        "data": "synthetic",
        "config": {"workload": wl["name"], "pairs_per_gpu_per_step": B, "global_pairs_per_step": world * B,
                   "parallelism": f"pair-sharded x{world}" + ((" + RCCL gather of match records to rank 0" if backend == "nccl" else
                                                                f" + {backend} (host-side bring-up hook, NOT RCCL) gather of match records to rank 0") if use_pg else ""),
                   "weights": "synthetic, BN-calibrated (synth.py seeds 123/456)", "matches_per_pair": round(n_matches / B, 1),
                   "inputs": f"synth_pair(seed = {wl['seed0']} + pair index): the first pairs are the unselected sweep seeds of the parity tests"},
    }
    if parity is not None:
        line["parity_in_run"] = parity
    if strict is not None:
        line["parity_in_run_strict"] = strict
    line["step_ms"] = stats["step_ms"]
    line["clocks"] = sampler.report()
    # the box beside the number: boxes of this pool differ by +-4 % in sustained clock under this power-bound step, as
    # much as a round's gain; `value` is what was measured, `value_at_2400MHz` the same scaled to the clock the peaks are quoted at
    _clk = (line["clocks"].get("sclk_mhz") or {}).get("median")
    line["sclk_mhz_median"] = _clk
    line["socket_power_w_median"] = (line["clocks"].get("socket_power_w") or {}).get("median")
    line["value_at_2400MHz"] = round(value * 2400.0 / _clk, 3) if _clk else None
    line["value_at_2400MHz_note"] = ("value x 2400 / median sclk of rank 0's GPU inside the timed region: a normalisation for comparing runs on "
                                     "different boxes, not a measurement (HBM-bound kernels do not scale with sclk)")
    if use_pg:
        line["backend"] = {"process_group": dist.get_backend(), "collective": "RCCL gather (ncclSend/ncclRecv group) over xGMI" if backend == "nccl" else f"{backend} on the host (bring-up hook)"}
        line["world_checked"] = {"WORLD_SIZE": world, "--gpus": args.gpus, "process_group_size": dist.get_world_size(), "ranks_counted_by_all_reduce": world}
        line["compute_ms_per_step"] = stats["compute_ms_per_step"]
        line["gather_ms_per_step"] = stats["gather_ms_per_step"]
        line["per_step_split_note"] = "HIP events on the launch stream either side of the collective, mean over the K timed steps, max over ranks"

    # ---- roofline: second pass of the same K steps with per-launch HIP events on the launch stream
    # (every rank runs the steps -- step() contains the gather, a rank-0-only pass would hang the others;
    #  only rank 0 records events and reports)
    rows = None
    if not args.no_roofline_pass:
        eng = matching._shared.engine
        if rank == 0:
            log("roofline pass (per-launch HIP events)")
            eng.timing_reset()
            eng.set_timing(True)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if rank == 0:
            rows = eng.timing_report(forms=True)
            eng.set_timing(False)
            eng.timing_reset()
    if rows:
        line["roofline"] = roofline_block(rows, wl, B, args.steps, dt, matching._shared.engine.lib.imx_version().decode(), args.workload)
        clk = (line["clocks"].get("sclk_mhz") or {}).get("median")
        if clk and line["roofline"]["bound"] == "mfma":
            line["roofline"]["frac_at_clock"] = round(line["roofline"]["frac"] * 2400.0 / clk, 4)
            line["roofline"]["frac_at_clock_note"] = f"frac x 2400 / {clk} MHz: the peak is quoted at 2.4 GHz, the median sclk sampled inside the timed region was {clk} MHz"
    line["config"]["arithmetic"] = arithmetic_text(line.get("roofline"))
    if rank == 0 and world == 1 and not sp_only and not args.no_extras:
        log("single-pair latency (Matching.forward, B = 1)")
        line["latency_b1_ms"] = latency_b1(matching, wl, device)
        log("PCIe-inclusive rate (IngestPipeline)")
        pc = pcie_inclusive(matching, wl, B)
        line["pcie_inclusive_pairs_s"] = pc["value"]
        line["pcie_inclusive"] = pc
        if not use_pg:
            log("the gather at world 1 (RCCL)")
            line["gather_ms"], line["gather_note"] = gather_baseline(matching.pack_records(pair_ids_dev, out, pad_to=rows_per_rank))
        log("C2 leg (SuperPoint only)")
        line["c2"] = c2_leg(matching, torch.cat([img0, img1]), args.steps)
        if args.workload == "c3":
            log("C5 leg (1280x960, d=256, 2048 kpts, 100 iterations)")
            line["c5"] = c5_leg(device)
    if use_pg:
        barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not sp_only:
        log("cpu baseline (oracle on host cores)")
        line["cpu_baseline"] = cpu_baseline(args.workload)
    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
