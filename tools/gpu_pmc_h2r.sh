#!/bin/bash
# PMC view of conv_h2r_kernel against the chunked conv_h2_kernel on 18 x 256^2 64 -> 64 (maxima handed in)
mkdir -p gpurun_out/pmc_h2r
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_h2r
cat > /tmp/h2r_probe.py <<'PY'
import math, os, sys
import torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from pix2latent_amd import ops as O, _native as N
dev = 'cuda'
g = torch.Generator().manual_seed(0)
for B, H in ((18, 256), (3, 1024)):
    x = torch.randn(B, H, H, 64, generator=g).to(dev)
    am = x.abs().reshape(B, 2048, -1).amax(dim=2).contiguous()
    wp = O.pack_conv_weight((torch.randn(64, 64, 3, 3, generator=g) / math.sqrt(576)).to(dev), 9, 64, 64, wfmt=2)
    bias = torch.randn(64, generator=g).to(dev)
    for form in (N.FORM_NO_WINO, N.FORM_NO_WINO | N.FORM_H2R_SEQ_EPI, N.FORM_NO_WINO | N.FORM_NO_H2R):
        for _ in range(4):
            O.conv(x, wp, B, H, H, 64, 64, 9, wfmt=2, amax_in=am, form=form, bias=bias, act=N.ACT_RELU)
torch.cuda.synchronize()
PY
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc -o c -- python /tmp/h2r_probe.py > $O/pmc.log 2>&1
cd $R
python - <<'PY' | tee gpurun_out/pmc_h2r/pmc_h2r.txt
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_h2r/pmc/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if 'conv_h2' not in n or 'pack' in n or 'wmax' in n: continue
    kind = ('resident' + n.split('conv_h2r_kernel')[1][:7]) if 'conv_h2r' in n else 'chunked'
    by.setdefault((int(r['Dispatch_Id']), kind, r['Grid_Size']), {})[r['Counter_Name']] = float(r['Counter_Value'])
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
rm -rf $O/pmc
