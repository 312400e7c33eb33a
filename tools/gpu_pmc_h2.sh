#!/bin/bash
# PMC counters of conv_h2_kernel on 64^2 256->256 x 18 for the product build and ablation builds
# (tools/ab_build.sh p2l_h2 -DP2L_H2_ABL=n -> tools/micro/libp2l_hip_h2abl<n>.so)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_h2
for a in product "$@"; do
  L=""; [ "$a" != product ] && L=$R/tools/micro/libp2l_hip_h2abl$a.so
  cd /tmp
  P2L_LIB_PATH=$L timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_h2/$a -o c -- python $R/tools/conv_probe_h2.py > /dev/null 2>&1
  cd $R
  python - $a <<'PY'
import csv, glob, sys, collections
a = sys.argv[1]
f = glob.glob('gpurun_out/pmc_h2/%s/**/*counter_collection.csv' % a, recursive=True)[0]
by = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'conv_h2_kernel' not in r['Kernel_Name']: continue
    by.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
v = list(by.values())[-1]
gui = v['GRBM_GUI_ACTIVE'] / 8; wc = v['SQ_WAVE_CYCLES']
print('%-8s mfma_busy %.1f%% parked %.1f%% issue_stall %.1f%% issuing %.1f%% valu %.1f%% lds_busy %.1f%% lds_conf %.1f%% gui_cycles %d' % (
    a, 100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024), 100 * v['SQ_WAIT_ANY'] / wc, 100 * v['SQ_WAIT_INST_ANY'] / wc,
    100 * v['SQ_ACTIVE_INST_ANY'] / wc, 100 * v.get('SQ_ACTIVE_INST_VALU', 0) / wc, 100 * v['SQ_LDS_IDX_ACTIVE'] / (gui * 256),
    100 * v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1), gui))
PY
done
