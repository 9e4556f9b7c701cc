#!/usr/bin/env python
"""rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, tools/gpu_profile_all.sh) -> profiles/<tag>_pmc_traffic.json,
the file bench.py's roofline.traffic is read from.  HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes): on gfx950
FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section).  The JSON records the library build id the
passes ran on; bench.py drops `traffic` when it times a different build.
usage: tools/pmc_traffic.py <fetch.txt> <write.txt> <pairs> <out.json>   (run on the GPU box, same build as the passes)"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# bench.py kernel name -> pattern on the rocprofv3 kernel name (tools/rocpd_pmc.py: short()).  The gemm_x3 instantiations are told
# apart by their template arguments <BN, RES, RELU>: mlp.0 is the only one with ReLU, mlp.3 the only 128-column one with a residual;
# <128, false, false> averages q|k|v (18 launches) with final_proj and convDb (1 each)
NAMES = {"conv1ab_pool": r"conv1ab_wino24", "attention": r"attention_h2q2_kernel|attention_x3_kernel|attention_kernel", "sinkhorn": r"sinkhorn_slab",
         "gnn_tail": r"gnn_tail_x3_kernel<3"}      # (round 4: the three gemm_x3 instantiations of a layer tail became gnn_tail_x3; the remaining
                                                   # gemm_x3 launches -- heads, keypoint encoder, layer 0's q|k|v -- share instantiations and are not split)


def table(path):
    rows = {}
    for ln in open(path).read().splitlines()[1:]:
        m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.eE+-]+)\s*$", ln)
        if m:
            rows[m.group(1).strip()] = float(m.group(4))
    return rows


def main(fetch, write, pairs, out):
    from image_matching_amd import _lib
    build = _lib.load_library().imx_version().decode()
    f, w = table(fetch), table(write)
    kernels = {}
    for name, pat in NAMES.items():
        fk = [v for k, v in f.items() if re.search(pat, k)]
        wk = [v for k, v in w.items() if re.search(pat, k)]
        if fk and wk:
            kernels[name] = {"fetch_kb": fk[0], "write_kb": wk[0], "traffic_bytes": (2 * fk[0] + wk[0]) * 1024}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KB per launch averaged over launches, "
                       "tools/run_pairs.py; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
               "build": build, "pairs_per_gpu": int(pairs), "kernels": kernels}, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main(*sys.argv[1:5])
