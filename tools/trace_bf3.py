import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pix2latent_amd import ops as O, _native as N
dev='cuda'
B,H,Cin,Cout=18,64,256,256
g=torch.Generator().manual_seed(0)
x=torch.relu(torch.randn(B,H,H,Cin,generator=g)).to(dev)
w=(torch.randn(Cout,Cin,3,3,generator=g)/math.sqrt(Cin*9)).to(dev)
wp=O.pack_conv_weight(w,9,Cout,Cin,wfmt=1)
for _ in range(3): O.conv(x,wp,B,H,H,Cin,Cout,9,wfmt=1,splitk=1)
torch.cuda.synchronize()
O.conv(x,wp,B,H,H,Cin,Cout,9,wfmt=1,splitk=1)
torch.cuda.synchronize()
t=O.conv.last_ws.view(torch.int64).cpu()[:8*128].view(2,4,128)
import numpy as np
for blk in range(2):
    for wv in range(4):
        a=t[blk,wv].numpy()
        n=(a!=0).sum()
        a=a[:n]
        d=np.diff(a)
        # stamps per chunk: 6 (start, after issue loads, after mfma, after bar1, after write, after bar2)
        per=d[:(len(d)//6)*6].reshape(-1,6) if len(d)>=6 else d
        print('blk',blk,'wave',wv,'n',n,'first',a[0]-t[0,0,0].item())
        print('  mean per chunk [issue, mfma, bar1, write, bar2, loop]:', per[1:-1].mean(0).round(0))
