#!/bin/bash
# PMC view of the fused attention kernels (tools/bench_attn.py), one line per kernel
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_attn -o c -- python $R/tools/bench_attn.py > $R/gpurun_out/pmc_attn.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_attn2 -o c -- python $R/tools/bench_attn.py >> $R/gpurun_out/pmc_attn.log 2>&1
cd $R
python - <<'PY' | tee gpurun_out/pmc_attn.txt
import csv, glob, collections
for d in ('pmc_attn', 'pmc_attn2'):
    fs = glob.glob('gpurun_out/%s/**/*counter_collection.csv' % d, recursive=True)
    if not fs: print('no data', d); continue
    rows = list(csv.DictReader(open(fs[0])))
    by = collections.OrderedDict()
    for r in rows:
        n = r['Kernel_Name']
        if 'attn_' not in n: continue
        key = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        a = by.setdefault(key, collections.defaultdict(float))
        a[r['Counter_Name']] += float(r['Counter_Value']); a['_n' + r['Counter_Name']] += 1
    for k, v in by.items():
        c = {n: v[n] / v['_n' + n] for n in v if not n.startswith('_n')}
        gui = c['GRBM_GUI_ACTIVE'] / 8
        if 'SQ_WAVE_CYCLES' in c:
            wc = c['SQ_WAVE_CYCLES']
            print('%-28s clk %8d  mfma_busy %5.1f%%  parked %5.1f%%  issue_stall %5.1f%%  issuing %5.1f%%  valu %5.1f%%  lds_busy %5.1f%%  lds_conf %4.1f%%' % (
                k, gui, 100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024), 100 * c['SQ_WAIT_ANY'] / wc, 100 * c['SQ_WAIT_INST_ANY'] / wc,
                100 * c['SQ_ACTIVE_INST_ANY'] / wc, 100 * c['SQ_ACTIVE_INST_VALU'] / wc, 100 * c['SQ_LDS_IDX_ACTIVE'] / (gui * 256),
                100 * c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1)))
        else:
            print('%-28s clk %8d ' % (k, gui) + '  '.join('%s %.3g' % (n.replace('SQ_', ''), c[n] / gui) for n in sorted(c) if n != 'GRBM_GUI_ACTIVE'))
PY
