#!/usr/bin/env python3
"""Turn gpurun_out/<tag>_<workload>/ (written by profiles/collect.sh) into the committed evidence:
   profiles/<tag>_<workload>_bench.json, _kernel_stats.csv, _hbm_traffic.json.

HBM bytes per call follow MI355X_MICROARCH.md's HBM section: FETCH_SIZE / WRITE_SIZE are in KiB and on gfx950
FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B.  The calibration is inside the same run:
k_compare reads 2 x n x B bytes with 16 B/lane loads, so factor = expected / reported.

usage: python3 profiles/summarize.py <tag> <workload>"""
import csv
import glob
import json
import os
import sys

tag, wl = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "%s_%s" % (tag, wl))
pre = os.path.join(root, "profiles", "%s_%s" % (tag, wl))
bench = json.loads([l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1])
KERNELS = {
    "lz4_decode": ("k_lz4_index", "k_lz4_dec_seq", "k_lz4_dec_ring"),
    "zstd_decode": ("k_zplan", "k_zhuf", "k_zmove", "k_zchain", "k_zmat", "k_zexec", "k_zstd_dec"),
    "lz4": ("k_lz4_index", "k_lz4_dec_seq", "k_lz4_dec_ring", "k_lz4_enc"),
    "zstd": ("k_zplan", "k_zhuf", "k_zmove", "k_zchain", "k_zmat", "k_zexec", "k_zstd_dec", "k_zstd_enc"),
}[wl]

stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
open(pre + "_kernel_stats.csv", "w").write(open(stats).read())


def counter(kind):
    f = glob.glob(os.path.join(src, kind, "**", "*counter_collection.csv"), recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return per


fetch, write = counter("fetch"), counter("write")
cfg = bench["config"]
n = cfg["blocks_per_gpu"]
B = cfg.get("block_size", 131072)
comp = round(n * B / cfg["compression_ratio"])
cmp_k = [k for k in fetch if "k_compare" in k][0]
cmp_kib = sum(fetch[cmp_k]) / len(fetch[cmp_k])
factor = (2.0 * n * B / 1024.0) / cmp_kib
calls = 3  # collect.sh runs the counter passes with --steps 2 --warmup 1
per_kernel = {}
f_tot = w_tot = 0.0
for k in fetch:
    if any(d in k for d in KERNELS):
        name = k.split("(")[0].split("::")[-1]
        fk = sum(fetch[k]) * 1024.0 / calls
        wk = sum(write.get(k, [0.0])) * 1024.0 / calls
        per_kernel[name] = {"launches_per_call": len(fetch[k]) / calls, "FETCH_SIZE_bytes_per_call_raw": fk, "WRITE_SIZE_bytes_per_call": wk}
        f_tot += fk
        w_tot += wk
algo = bench["roofline"]["algorithmic_bytes_per_launch"]
out = {
    "workload": {"method": (wl.split("_")[0] + ("_roundtrip" if "_" not in wl else "")), "param": cfg.get("param", 1), "blocks_per_gpu": n,
                 "block_size": B, "distribution": cfg["distribution"]},
    "kernels": per_kernel,
    "FETCH_SIZE_bytes_per_call_raw": f_tot, "WRITE_SIZE_bytes_per_call": w_tot,
    "calibration": {"kernel": "k_compare (2 x n x B bytes, 16 B/lane coalesced loads)",
                    "FETCH_SIZE_reported_KiB": cmp_kib, "expected_KiB": 2.0 * n * B / 1024.0, "factor": factor},
    "compressed_bytes": comp, "algorithmic_bytes_per_launch": algo,
}
if wl == "lz4_decode":
    # wide coalesced staging reads: the compressed input once by the index kernel (16 B/lane pieces of 128-B lines) and
    # once by the decoder (8 B/lane, 512 B per instruction); everything else (index rows, far-match read-back) is
    # narrow and counted at face value
    staged = comp * (2 if any("k_lz4_index" in k for k in per_kernel) else 1)
    wide = min(staged, f_tot * factor) if factor > 1.5 else 0.0
    traffic = w_tot + (f_tot - wide / factor) + wide if factor > 1.5 else w_tot + f_tot
    out["traffic_derivation"] = ("WRITE_SIZE + [FETCH_SIZE - staged/factor] (narrow reads at face value) + staged; staged = the "
                                 "compressed input, read once by k_lz4_index and once by k_lz4_dec_seq, under-counted by the calibrated factor")
elif wl == "zstd_decode":
    wide = min(comp, f_tot * factor) if factor > 1.5 else 0.0
    traffic = w_tot + (f_tot - wide / factor) + wide if factor > 1.5 else w_tot + f_tot
    out["traffic_derivation"] = "WRITE_SIZE + [FETCH_SIZE - csize/factor] + csize (compressed input staging under-counted by the calibrated factor)"
else:
    traffic = w_tot + f_tot * (factor if factor > 1.5 else 1.0)
    out["traffic_derivation"] = ("WRITE_SIZE + FETCH_SIZE x factor: an upper bound (every read counted as a wide coalesced one; the encoders' "
                                 "table probes are narrow and over-counted by this)")
out["traffic_bytes_per_launch"] = traffic
out["traffic_over_algorithmic"] = traffic / algo
json.dump(out, open(pre + "_hbm_traffic.json", "w"), indent=1)
bench["roofline"]["traffic"] = traffic
json.dump(bench, open(pre + "_bench.json", "w"), indent=1)
print(pre, "traffic/algorithmic = %.3f" % (traffic / algo), "factor %.2f" % factor)
for row in csv.DictReader(open(stats)):
    if any(d in row["Name"] for d in KERNELS):
        print("%-50s calls %4s avg %10.3f ms  %5s%%" % (row["Name"].split("(")[0][-50:], row["Calls"], float(row["AverageNs"]) / 1e6, row["Percentage"]))
