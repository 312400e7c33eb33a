#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_conv -o c -- python $R/tools/conv_probe.py > $R/gpurun_out/pmc_conv.log 2>&1
cd $R
ls gpurun_out/pmc_conv
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_conv/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
by = collections.defaultdict(dict)
for r in rows:
    if 'conv_mfma' not in r['Kernel_Name']: continue
    key = (r['Dispatch_Id'], r['Kernel_Name'][38:78], r['Grid_Size'])
    by[key][r['Counter_Name']] = float(r['Counter_Value'])
    by[key]['t'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) if 'End_Timestamp' in r else 0
for k, v in sorted(by.items(), key=lambda kv: int(kv[0][0])):
    print(k, {a: round(b) for a, b in v.items()})
PY
