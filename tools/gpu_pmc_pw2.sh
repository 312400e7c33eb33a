#!/bin/bash
# where the full-tile pointwise kernel waits: SQ / TCP / TCC counters per layer (single launches)
mkdir -p gpurun_out/pmc_pw2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_pw2
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
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TCR_TCP_STALL_CYCLES" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ TCC_TAG_STALL TCC_EA0_RDREQ_LEVEL" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/s$i -o c -- python /tmp/pw_probe.py > $O/s$i.log 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/pmc_pw2/pmc_pw2.txt
import csv, glob, collections
CASES = [(32, 1024, 256), (32, 256, 1024), (64, 512, 256), (64, 256, 512), (64, 512, 128), (64, 128, 512), (64, 512, 64),
         (64, 64, 512), (128, 256, 64), (128, 64, 256), (128, 256, 128), (128, 128, 256), (256, 128, 64), (256, 64, 128)]
tab = collections.defaultdict(dict)
for f in glob.glob('gpurun_out/pmc_pw2/s*/**/*counter_collection.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'pw_h2_kernel' in r['Kernel_Name']]
    ids = sorted({int(r['Dispatch_Id']) for r in rows})
    for r in rows:
        j = ids.index(int(r['Dispatch_Id']))
        if j % 3 == 2:
            tab[j // 3][r['Counter_Name']] = float(r['Counter_Value'])
names = sorted({n for t in tab.values() for n in t})
for i, c in enumerate(CASES):
    t = tab.get(i, {})
    print('%4d^2 %5d -> %-5d' % c, ' '.join('%s=%.4g' % (n, t[n]) for n in names if n in t))
PY
