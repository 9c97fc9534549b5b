#!/usr/bin/env python3
"""Per-dispatch view of the decode that follows an encode: kernel durations and cache / translation
counters of the FIRST decode after k_lz4_enc2 against the SECOND one right behind it (CRYO_BENCH_DEC_AGAIN), same process.
usage (on the GPU box, from the repo root): python3 profiles/scripts/r05_after_encode.py"""
import csv, glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out", "r05_after_encode")
shutil.rmtree(OUT, ignore_errors=True)
os.makedirs(OUT)
env = dict(os.environ, TMPDIR="/tmp", CRYO_BENCH_DEC_AGAIN="1")
cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "lz4", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"]

def rows(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []

def split(vals):
    """decode dispatches come in pairs per step: first (after the encode), again"""
    return vals[0::2], vals[1::2]

avg = lambda v: sum(v) / max(1, len(v))
# 1. durations per dispatch
d = os.path.join(OUT, "trace")
subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "run", "--"] + cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=900)
tr = sorted(rows(d, "*kernel_trace.csv"), key=lambda r: int(r["Start_Timestamp"]))
for kern in ("k_lz4_index", "k_lz4_dec_seq"):
    v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr if kern in r["Kernel_Name"]]
    a, b = split(v)
    print("%-14s duration ms: first after encode %s | again %s" % (kern, " ".join("%.3f" % x for x in a), " ".join("%.3f" % x for x in b)))
# gaps between the encode's end and the first decode's start
enc = [r for r in tr if "k_lz4_enc" in r["Kernel_Name"]]
idx = [r for r in tr if "k_lz4_index" in r["Kernel_Name"]]
print("encode durations ms:", " ".join("%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in enc))
# 2. counters per dispatch, one group per pass
groups = [["TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"],
          ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_REQUEST_sum"],
          ["TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_RD_UNCACHED_32B_sum", "TCC_WRITEBACK_sum"],
          ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE"]]
for gi, g in enumerate(groups):
    d = os.path.join(OUT, "pmc%d" % gi)
    p = subprocess.run(["rocprofv3", "--pmc"] + g + ["--output-format", "csv", "-d", d, "-o", "run", "--"] + cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=900)
    rr = rows(d, "*counter_collection.csv")
    if not rr:
        print("counter group", g, "not collected:", (p.stderr or p.stdout)[-300:].replace("\n", " | "))
        continue
    for kern in ("k_lz4_index", "k_lz4_dec_seq"):
        for cname in g:
            v = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in rr if kern in r["Kernel_Name"] and r["Counter_Name"] == cname]
            v = [x for _, x in sorted(v)]
            if not v:
                continue
            a, b = split(v)
            print("%-14s %-34s first after encode %.4g | again %.4g | ratio %.3f" % (kern, cname, avg(a), avg(b), avg(a) / avg(b) if avg(b) else 0))
