#!/bin/bash
# PMC passes of one layer on hx2p and hx2q (kbench binary directly behind `--`)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; K=$R/tools/kbench/conv_bench
cd /tmp && export TMPDIR=/tmp
export REPS=30
ARGS=${ARGS:-"32 64 64 0 0"}
rocprofv3 -L > $O/counters.txt 2>&1
for w in hx2p hx2q; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU" \
             "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"; do
    i=$((i+1))
    rm -rf /tmp/p_${w}_$i
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/p_${w}_$i -- $K $ARGS 512 $w > /dev/null 2>&1 || { echo "pass $w $i failed"; continue; }
    f=$(find /tmp/p_${w}_$i -name '*counter_collection.csv' | head -1)
    python3 - "$f" "$w" <<'PY'
import csv, sys, collections
f, w = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'hx2' not in k or 'pack' in k or 'scale' in k: continue
    acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(w, k)
    for c, v in sorted(d.items()):
        v = v[len(v)//2:]
        print('   %-28s %14.0f  (n=%d)' % (c, sum(v)/len(v), len(v)))
PY
  done
done
