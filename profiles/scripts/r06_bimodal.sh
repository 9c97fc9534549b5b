#!/bin/bash
# round 6: the index pass runs at 2.03 or 2.22 ms from process to process.  Ten processes: device addresses
# of the buffers (CRYO_BENCH_TRACE), the index pass's and the decoder's average time (rocprofv3 --kernel-trace --stats), then a
# TCC / TCP counter pass of one process (its speed class is read from the step times it prints).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r06_bimodal; rm -rf $O; mkdir -p $O
cd /tmp
for i in 1 2 3 4 5 6 7 8 9 10; do
  CRYO_BENCH_TRACE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$i -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $O/p$i.out 2> $O/p$i.err
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, json, re
for i in range(1, 11):
    d = 'gpurun_out/r06_bimodal/p%d' % i
    st = {}
    for f in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            for k in ('k_lz4_index', 'k_lz4_dec_seq'):
                if k in r['Name']: st[k] = float(r['AverageNs']) / 1e6
    ptr = [l.strip() for l in open(d + '.err') if 'device pointers' in l]
    val = None
    for l in open(d + '.out'):
        if l.startswith('{'): val = json.loads(l)['value']
    print(i, 'index %.3f ms  dec %.3f ms  value %s  %s' % (st.get('k_lz4_index', 0), st.get('k_lz4_dec_seq', 0), val, ptr[0][ptr[0].find('comp'):] if ptr else ''))
PY
