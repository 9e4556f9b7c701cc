"""Does a SECOND HIP stream buy throughput?  The C3 step is one stream of ~600 kernels: MFMA-bound convs / attention / GNN tails,
then HBM-bound Sinkhorn and the latency-bound keypoint kernels.  Here the 64-pair batch is split over two library handles (own
workspaces) on two streams, so that one half's Sinkhorn / NMS can sit beside the other half's convs.

  a) one handle, 64 pairs, one stream                      (what bench.py times)
  b) one handle, 2 x 32 pairs back to back, one stream     (the cost of the smaller batch alone)
  c) two handles, 32 pairs each, two streams               (overlap)
  d) as c, the second stream started half a step late      (dephased)
usage: python tools/two_stream_probe.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = bench.WORKLOADS["c3"]
    B = 64
    m = [bench.build_matching(wl, dev)[0] for _ in range(2)]
    img0, img1 = bench.resident_inputs(wl, list(range(B)), dev)
    h = B // 2
    halves = [(img0[:h].contiguous(), img1[:h].contiguous()), (img0[h:].contiguous(), img1[h:].contiguous())]
    st = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

    def run(fn, n):
        for _ in range(2):
            fn(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def a(i):
        m[0].match_batch(img0, img1)

    def b(i):
        m[0].match_batch(*halves[0])
        m[0].match_batch(*halves[1])

    def c(i):
        for k in (0, 1):
            with torch.cuda.stream(st[k]):
                m[k].match_batch(*halves[k])

    sp_only = [None]

    def d(i):
        # stream 1 runs half a step behind: an extra SuperPoint-sized delay once, before its first batch
        if i == 0 and sp_only[0] is None:
            sp_only[0] = True
            with torch.cuda.stream(st[1]):
                m[1]._shared.get_engine([0, 1]).superpoint_batch(torch.cat(halves[1]))
        c(i)

    ref = m[0].match_batch(img0, img1)
    torch.cuda.synchronize()
    for k in (0, 1):
        with torch.cuda.stream(st[k]):
            o = m[k].match_batch(*halves[k])
        torch.cuda.synchronize()
        sl = slice(0, h) if k == 0 else slice(h, B)
        assert torch.equal(o["matches0"], ref["matches0"][sl]) and torch.equal(o["matching_scores0"], ref["matching_scores0"][sl]), "half-batch differs from the full batch"
    for name, fn in (("a one stream, 64 pairs", a), ("b one stream, 2 x 32", b), ("c two streams, 32 + 32", c), ("d two streams, dephased", d), ("a again", a)):
        ms = run(fn, steps)
        print(f"{name:28s} {ms:8.3f} ms / 64 pairs = {64e3 / ms:7.1f} pairs/s", flush=True)


if __name__ == "__main__":
    main()
