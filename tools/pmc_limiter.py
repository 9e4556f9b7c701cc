#!/usr/bin/env python
"""rocprofv3 PMC passes of tools/gpu_pmc_limiter.sh -> profiles/<tag>_pmc_limiter.json, the file bench.py's roofline.limiter is read
from (same build-id rule as profiles/*_pmc_traffic.json).  Per kernel, per launch (averages over the launches of two steps):
  mfma_busy         SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GPU cycles)        -- the matrix pipes
  valu / lds / vmem_active  4 x SQ_ACTIVE_INST_* / SIMD cycles
  wait_any          SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   -- the share of resident-wave time spent at an s_waitcnt
  l1_l2_read        TCP_TCC_READ_REQ_sum x 128 bytes per launch (and per CU cycle)     -- what the CUs pull from L2 (the U stream)
  l1_data_path_frac TCP_TOTAL_CACHE_ACCESSES_sum (64-byte accesses) against 64 bytes per clock and CU
  l2_read_latency   TCP_TCC_READ_REQ_LATENCY_sum / TCP_TCC_READ_REQ_sum (cycles)
  lds_conflict      SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
usage: tools/pmc_limiter.py <prefix of the pass tables> <pairs> <out.json> [build string] [workload note]
(run on the GPU box, same build as the passes; off the box, tables of another build / workload -- tools/gpu_pmc_c5.sh -- name their build)"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = {"conv1ab_pool": r"conv1ab_wino24", "conv3x3_pair": r"conv3x3_wino24p", "conv3x3_tile": r"conv3x3_wino24h", "attention": r"attention_h2q2_kernel|attention_x3p?_kernel|attention_kernel",
         "sinkhorn": r"sinkhorn_slab", "gnn_tail": r"gnn_tail_h2|gnn_tail_x3", "gemm_x3": r"gemm_x3<", "gemm_h2": r"gemm_h2(<|IL)"}
NSIMD, NCU, NXCD = 4, 256, 8


def table(path):
    """tools/rocpd_pmc.py output -> {kernel: {counter: avg, 'avg_us': .., 'n': ..}} (counter columns are right-truncated to 18 characters)"""
    lines = open(path).read().splitlines()
    if not lines:
        return {}, []
    cols = lines[0].split()[3:]
    rows = {}
    for ln in lines[1:]:
        parts = ln.rsplit(None, len(cols) + 2)
        if len(parts) != len(cols) + 3:
            continue
        try:
            vals = [float(x) for x in parts[3:]]
            rows[parts[0].strip()] = dict(zip(cols, vals), avg_us=float(parts[2]), n=int(parts[1]))
        except ValueError:
            continue
    return rows, cols


def pick(rows, pat):
    """launch-weighted average over the instantiations that match (the pair kernel has one per layer shape)"""
    sel = [v for k, v in rows.items() if re.search(pat, k)]
    if not sel:
        return None
    n = sum(v["n"] for v in sel)
    return {c: sum(v[c] * v["n"] for v in sel) / n for c in sel[0] if c != "n"}, n


def col(d, suffix):
    for k, v in d.items():
        if suffix.endswith(k) or k.endswith(suffix[-18:]):
            return v
    return None


def main(prefix, pairs, out, build=None, workload=None):
    if build is None:
        from image_matching_amd import _lib
        build = _lib.load_library().imx_version().decode()
    tabs = {}
    for tag in ("sqa", "sqb", "ta2", "tcp", "tcp2"):
        try:
            tabs[tag] = table(prefix + tag + ".txt")[0]
        except OSError:
            tabs[tag] = {}
    kernels = {}
    for name, pat in NAMES.items():
        got = {tag: pick(rows, pat) for tag, rows in tabs.items()}
        if not got["sqa"] or not got["tcp"]:
            continue
        a, n = got["sqa"]
        gui = col(a, "GRBM_GUI_ACTIVE") / NXCD                 # GRBM_GUI_ACTIVE is summed over the eight XCDs
        simd_cycles = NSIMD * NCU * gui
        k = {"launches_seen": n, "avg_launch_us_profiled": round(a["avg_us"], 1), "gpu_cycles": round(gui)}
        k["mfma_busy"] = round(col(a, "SQ_VALU_MFMA_BUSY_CYCLES") / simd_cycles, 4)
        # SQ_ACTIVE_INST_* / SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES count in units of four cycles
        k["valu_active_per_simd_cycle"] = round(4 * col(a, "SQ_ACTIVE_INST_VALU") / simd_cycles, 4)
        k["lds_active_per_simd_cycle"] = round(4 * col(a, "SQ_ACTIVE_INST_LDS") / simd_cycles, 4)
        k["vmem_active_per_simd_cycle"] = round(4 * col(a, "SQ_ACTIVE_INST_VMEM") / simd_cycles, 4)
        k["wait_any_of_wave_cycles"] = round(col(a, "SQ_WAIT_INST_ANY") / max(1.0, col(a, "SQ_WAVE_CYCLES")), 4)
        k["waves_resident_per_simd"] = round(4 * col(a, "SQ_WAVE_CYCLES") / simd_cycles, 3)
        if got["sqb"]:
            b = got["sqb"][0]
            for cn in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"):
                k[cn.lower()] = col(b, cn)
            idx = col(b, "SQ_LDS_IDX_ACTIVE")
            k["lds_bank_conflict_of_lds_active"] = round(col(b, "SQ_LDS_BANK_CONFLICT") / idx, 4) if idx else None
        c = got["tcp"][0]
        cg = col(c, "GRBM_GUI_ACTIVE") / NXCD
        req = col(c, "TCP_TCC_READ_REQ_sum")
        k["l1_l2_read_requests"] = req
        k["l1_l2_read_bytes"] = req * 128.0                     # 128 bytes per request: calibrated on the first layer's U stream (tiles x 393216 B)
        k["l1_l2_read_TBps"] = round(req * 128.0 / (c["avg_us"] * 1e-6) / 1e12, 3)
        k["l1_l2_read_bytes_per_cu_cycle"] = round(req * 128.0 / (NCU * cg), 2)
        k["l2_read_latency_cycles"] = round(col(c, "TCP_TCC_READ_REQ_LATENCY_sum") / max(1.0, req), 1)
        if got["tcp2"]:
            c2 = got["tcp2"][0]
            g2 = col(c2, "GRBM_GUI_ACTIVE") / NXCD
            acc = col(c2, "TCP_TOTAL_CACHE_ACCESSES_sum")
            k["l1_accesses"] = acc
            k["l1_data_path_frac"] = round(acc * 64.0 / (64.0 * NCU * g2), 4)      # 64-byte accesses against 64 B/clk per CU
            k["tcp_pending_stall_per_cu_cycle"] = round(col(c2, "TCP_PENDING_STALL_CYCLES_sum") / (NCU * g2), 4)
        if got["ta2"]:
            t = got["ta2"][0]
            tg = col(t, "GRBM_GUI_ACTIVE") / NXCD
            k["ta_addr_stalled_by_tc_per_cu_cycle"] = round(col(t, "TA_ADDR_STALLED_BY_TC_CYCLES_sum") / (NCU * tg), 4)
            k["ta_data_stalled_by_tc_per_cu_cycle"] = round(col(t, "TA_DATA_STALLED_BY_TC_CYCLES_sum") / (NCU * tg), 4)
        kernels[name] = k
    json.dump({"workload": workload or "c3", "note": "rocprofv3 --pmc (kernel trace only), tools/gpu_pmc_limiter.sh over tools/run_pairs.py --pairs 64 (or tools/gpu_pmc_c5.sh: --workload c5 --pairs 8): passes sqa, sqb (SQ), ta2 (TA stall "
                       "cycles), tcp, tcp2 (TCP); per-launch averages over the launches of two steps (the pair / tile Winograd entries average their "
                       "layers); GRBM_GUI_ACTIVE / 8 = GPU cycles; TA_BUSY / TA_BUFFER_* make rocprofv3 abort on this box and are not collected",
               "build": build, "pairs_per_gpu": int(pairs), "kernels": kernels}, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main(*sys.argv[1:6])
