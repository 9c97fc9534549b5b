#!/bin/bash
# PC sampling of the headline decode (rocprofv3 beta): where the waves of k_lz4_dec_seq / k_lz4_index spend their cycles
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/r04_pcs; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
cd /tmp
for cfg in "stochastic cycles 1048576" "stochastic cycles 262144" "host_trap time 100" "host_trap time 10"; do
  set -- $cfg
  d=$O/$1_$3; mkdir -p $d
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 --kernel-trace --output-format csv -d $d -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 --warmup 2 ${BENCH_ARGS} > $d/log.txt 2>&1
  echo "$cfg rc $? $(tail -1 $d/log.txt | cut -c1-200)"
  find $d -name "*.csv" | xargs ls -la | head
done
# keep the merged output small: compress the sample files
cd $O && for f in $(find . -name "*pc_sampling*.csv"); do python3 - "$f" <<'PY'
import sys, csv, collections
f = sys.argv[1]
rd = csv.DictReader(open(f))
cols = rd.fieldnames
print(f, cols)
cnt = collections.Counter()
n = 0
for r in rd:
    n += 1
    key = tuple(r.get(c, "") for c in cols if c.lower() in ("code_object_id", "code_object_offset", "instruction_type", "stall_reason", "wave_issued", "instruction", "instruction_comment", "exec_mask", "wave_count", "dual_issue_valu", "inst_type", "reason_not_issued", "arb_state_issue", "arb_state_stall"))
    cnt[key] += 1
open(f + ".hist", "w").write("\n".join("%d\t%s" % (v, "\t".join(k)) for k, v in cnt.most_common()))
print(n, "samples", len(cnt), "distinct")
PY
rm -f "$f"; done
du -sh $O
