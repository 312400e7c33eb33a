#!/bin/bash
# HBM traffic of the full-tile pointwise kernel PER LAYER: FETCH_SIZE / WRITE_SIZE passes over single launches
mkdir -p gpurun_out/pmc_pw
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_pw
cat > /tmp/pw_probe.py <<'PY'
import math, os, sys
import torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from pix2latent_amd import ops as O, _native as N
dev = 'cuda'
B = 18
CASES = [(32, 1024, 256), (32, 256, 1024), (64, 512, 256), (64, 256, 512), (64, 512, 128), (64, 128, 512), (64, 512, 64),
         (64, 64, 512), (128, 256, 64), (128, 64, 256), (128, 256, 128), (128, 128, 256), (256, 128, 64), (256, 64, 128)]
g = torch.Generator().manual_seed(0)
for H, Cin, Cout in CASES:
    x = torch.randn(B, H, H, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)).to(dev)
    am = x.abs().amax(dim=(1, 2, 3)).view(B, 1).contiguous()
    wp = O.pack_conv_weight(w, 1, Cout, Cin, wfmt=3)
    for _ in range(3):
        O.conv(x, wp, B, H, H, Cin, Cout, 1, wfmt=3, amax_in=am)
    torch.cuda.synchronize()
PY
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $O/$c -o c -- python /tmp/pw_probe.py > $O/$c.log 2>&1
done
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o c -- python /tmp/pw_probe.py > $O/trace.log 2>&1
cd $R
python - <<'PY' | tee gpurun_out/pmc_pw/pmc_pw.txt
import csv, glob
CASES = [(32, 1024, 256), (32, 256, 1024), (64, 512, 256), (64, 256, 512), (64, 512, 128), (64, 128, 512), (64, 512, 64),
         (64, 64, 512), (128, 256, 64), (128, 64, 256), (128, 256, 128), (128, 128, 256), (256, 128, 64), (256, 64, 128)]
def vals(c):
    f = glob.glob('gpurun_out/pmc_pw/%s/**/*counter_collection.csv' % c, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if 'pw_h2_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return [float(r['Counter_Value']) for r in rows]
fe, wr = vals('FETCH_SIZE'), vals('WRITE_SIZE')
f = glob.glob('gpurun_out/pmc_pw/trace/**/*kernel_trace.csv', recursive=True)[0]
tr = [r for r in csv.DictReader(open(f)) if 'pw_h2_kernel' in r['Kernel_Name']]
tr.sort(key=lambda r: int(r['Dispatch_Id']))
us = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in tr]
print('pw_h2_kernel<0,false>, 18 candidates: HBM traffic per launch (FETCH_SIZE x 2 per the guide, KB -> MB) vs algorithmic')
print('  H   Cin  Cout |  us   | read MB: algo  fetched  x   | write MB: algo written  x   | fetched+written TB/s')
for i, (H, ci, co) in enumerate(CASES):
    j = 3 * i + 2
    ra, wa = 18 * H * H * ci * 4 / 1e6, 18 * H * H * co * 4 / 1e6
    F, W = fe[j] * 2 * 1024 / 1e6, wr[j] * 1024 / 1e6
    print('%4d %5d %5d | %5.1f | %7.1f %7.1f %5.2f | %7.1f %7.1f %5.2f | %5.2f' % (H, ci, co, us[j], ra, F, F / ra, wa, W, W / wa, (F + W) / us[j]))
PY
