#!/usr/bin/env python3
"""A/B runs of bench.py over variant builds (profiles/variants_<name>.so, see build_variant.sh; `prod` = the product
library):  python3 profiles/scripts/ab.py [--prof] [--reps N] [--args "<bench args>"] name1 name2 ...
Prints one line per variant: GB/s, ms per step and, with --prof, the per-kernel averages of a rocprofv3 --kernel-trace
--stats run of the same command (5 steps)."""
import argparse, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--prof", action="store_true")
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--args", default="")
ap.add_argument("--env", default="", help="K=V,K=V applied to every run")
ap.add_argument("names", nargs="+")
a = ap.parse_args()
extra = a.args.split()
for name in a.names:
    env = dict(os.environ, TMPDIR="/tmp")
    for kv in filter(None, a.env.split(",")):
        k, v = kv.split("=", 1); env[k] = v
    if name != "prod":
        env["CRYO_CODEC_LIB"] = os.path.join(ROOT, "profiles", "variants_%s.so" % name)
    else:
        env.pop("CRYO_CODEC_LIB", None)
    vals = []
    for r in range(a.reps):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", str(a.steps), "--warmup", "3"] + extra,
                           env=env, capture_output=True, text=True, timeout=900)
        try:
            d = json.loads(p.stdout.strip().splitlines()[-1])
            vals.append("%.1f GB/s %.3f ms frac %.4f" % (d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac", 0)))
        except Exception:
            vals.append("FAILED rc %d: %s" % (p.returncode, (p.stderr or p.stdout)[-400:].replace("\n", " | ")))
    line = "%-22s %s" % (name, " ; ".join(vals))
    if a.prof:
        out = "/tmp/ab_prof_%s" % name
        shutil.rmtree(out, ignore_errors=True)
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "run", "--",
                        sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "5", "--warmup", "1"] + extra,
                       env=env, capture_output=True, text=True, timeout=900, cwd="/tmp")
        f = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)
        if f:
            ks = []
            for r in csv.DictReader(open(f[0])):
                nm = r["Name"].split("(")[0].split("::")[-1]
                if any(x in nm for x in ("k_lz4", "k_z")) and "enc" not in nm:
                    ks.append("%s %.3f" % (nm.split("<")[0], float(r["AverageNs"]) / 1e6))
            line += " | " + ", ".join(ks)
    print(line, flush=True)
