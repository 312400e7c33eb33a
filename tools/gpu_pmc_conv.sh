#!/bin/bash
# PMC view of the 3x3 kernels (direct bf16x3 / fp16x2, Winograd 16x16 bf16x3 / fp16x2, sub-pixel fp16x2) on the bench layers:
# rocprofv3 --pmc over tools/conv_probe.py, one line per dispatch
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_conv -o c -- python $R/tools/conv_probe.py > $R/gpurun_out/pmc_conv.log 2>&1
rocprofv3 --list-avail > $R/gpurun_out/pmc_avail.txt 2>&1
cd $R
python - <<'PY' | tee gpurun_out/pmc_conv.txt
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_conv/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if 'conv_mfma' not in n and 'wino' not in n and 'conv_h2' not in n: continue
    if 'pack' in n or 'wmax' in n: continue
    key = (int(r['Dispatch_Id']), ('amax' if 'wino_amax' in n else 'wino16-f16x2' if ('wino16s' in n and 'true' in n) else 'wino16-bf16x3' if 'wino16s' in n else 'wino8' if 'wino_conv' in n else 'subpix-f16x2' if 'conv_h2_kernel<4' in n else 'direct-f16x2' if 'conv_h2' in n else 'direct-bf16x3'), r['Grid_Size'])
    by.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
print('disp kind grid  mfma_busy parked issue_stall issuing valu_share lds_busy lds_conf clk(GUI cycles)')
for k, v in sorted(by.items()):
    gui = v['GRBM_GUI_ACTIVE'] / 8
    wc = v['SQ_WAVE_CYCLES']
    print(k[0], k[1], k[2], '%.1f%%' % (100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024)),
          '%.1f%%' % (100 * v['SQ_WAIT_ANY'] / wc), '%.1f%%' % (100 * v['SQ_WAIT_INST_ANY'] / wc),
          '%.1f%%' % (100 * v['SQ_ACTIVE_INST_ANY'] / wc), '%.1f%%' % (100 * v.get('SQ_ACTIVE_INST_VALU', 0) / wc),
          '%.1f%%' % (100 * v['SQ_LDS_IDX_ACTIVE'] / (gui * 256)),
          '%.1f%%' % (100 * v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1)), round(gui))
PY
