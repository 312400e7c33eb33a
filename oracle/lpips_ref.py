"""ORACLE (test infrastructure, not product code): CPU restatement of the loss
side of the pix2latent hot path.

* l1 / l2 / masked / ReconstructionLoss / ProjectionLoss follow the reference
  file pix2latent/loss_functions.py line by line (cited per function) and are
  PINNED by tests/golden/losses.npz, generated from the imported reference
  (tools/make_golden.py).
* The LPIPS network itself is PARITY UNPINNED: `lpips>=0.1`
  (requirements.txt:15; call sites loss_functions.py:15,131,142) and the
  torchvision VGG16 / AlexNet weights are absent here.  `lpips_spatial` restates the
  published algorithm of lpips.LPIPS(net='vgg'|'alex', version='0.1', spatial=True)
  from recall.  The network is chosen by the keys of the weight dict
  ('vgg.conv*', 'alex.conv*' or 'squeeze.conv0*'); 'alex' is the reference default
  (loss_functions.py:87 `lpips_net='alex'`).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import torch
import torch.nn.functional as F

# lpips.ScalingLayer constants [3P-recall]
LPIPS_SHIFT = (-.030, -.088, -.188)
LPIPS_SCALE = (.458, .448, .450)

# torchvision vgg16.features conv layout; taps after relu1_2, 2_2, 3_3, 4_3, 5_3
VGG_CFG = [(3, 64), (64, 64), 'M', (64, 128), (128, 128), 'M',
           (128, 256), (256, 256), (256, 256), 'M',
           (256, 512), (512, 512), (512, 512), 'M',
           (512, 512), (512, 512), (512, 512)]
VGG_TAPS_AFTER_CONV = (1, 3, 6, 9, 12)   # 0-based conv indices whose ReLU output is tapped
VGG_CHNS = (64, 128, 256, 512, 512)


# torchvision alexnet.features [3P-recall]: conv(3,64,k11,s4,p2) relu | maxpool(3,2)
# conv(64,192,k5,p2) relu | maxpool(3,2) conv(192,384,k3,p1) relu | conv(384,256,k3,p1)
# relu | conv(256,256,k3,p1) relu ; lpips taps relu1..relu5
ALEX_CFG = [(3, 64, 11, 4, 2), (64, 192, 5, 1, 2), (192, 384, 3, 1, 1), (384, 256, 3, 1, 1),
            (256, 256, 3, 1, 1)]
ALEX_POOL_BEFORE = (1, 2)      # 3x3 stride-2 max-pool in front of conv index 1 and 2
ALEX_CHNS = (64, 192, 384, 256, 256)


def l1_loss(out, target):                       # loss_functions.py:20-22
    return torch.abs(target - out)


def l2_loss(out, target):                       # loss_functions.py:25-27
    return (target - out) ** 2


def masked_l1_loss(out, target, mask):          # loss_functions.py:41-50
    if mask.size(0) == 1:
        mask = mask.repeat(out.size(0), 1, 1, 1)
    if target.size(0) == 1:
        target = target.repeat(out.size(0), 1, 1, 1)
    loss = l1_loss(out, target)
    return torch.sum(loss * mask, [1, 2, 3]) / torch.sum(mask, [1, 2, 3])


def masked_l2_loss(out, target, mask):          # loss_functions.py:53-61
    if mask.size(0) == 1:
        mask = mask.repeat(out.size(0), 1, 1, 1)
    if target.size(0) == 1:
        target = target.repeat(out.size(0), 1, 1, 1)
    loss = l2_loss(out, target)
    return torch.sum(loss * mask, [1, 2, 3]) / torch.sum(mask, [1, 2, 3])


def _weighted(loss, weight, loss_mask):         # loss_functions.py:119-123 / :143-147
    if weight is not None:
        _w = weight if loss_mask is None else (loss_mask * weight)
        loss = torch.sum(loss * _w, [1, 2, 3]) / torch.sum(_w, [1, 2, 3])
    return loss


def reconstruction_loss(output, target, weight=None, loss_mask=None, loss_type='l1', tape=None):
    """ReconstructionLoss.__call__ (loss_functions.py:117-124).  `tape` (oracle/masks.py): the
    sign of (output - target) per pixel is a discrete decision of the L1 term."""
    fn = l1_loss if loss_type in ['l1', 1] else l2_loss
    if tape is not None and fn is l1_loss:
        return _weighted(tape.abs_diff(output, target, 'l1'), weight, loss_mask)
    return _weighted(fn(output, target), weight, loss_mask)


def vgg_features(Wv, x, tape=None):
    """torchvision vgg16.features sliced as lpips.pretrained_networks.vgg16 does
    [3P-recall]: returns relu1_2, relu2_2, relu3_3, relu4_3, relu5_3.
    `tape` (oracle/masks.py): record / replay of the ReLU signs and max-pool winners."""
    from .masks import relu as _relu, maxpool2 as _maxpool2
    taps, ci = [], 0
    for item in VGG_CFG:
        if item == 'M':
            x = _maxpool2(x, tape, 'vgg.pool%d' % ci)
        else:
            x = _relu(F.conv2d(x, Wv['vgg.conv%d.weight' % ci], Wv['vgg.conv%d.bias' % ci], padding=1),
                      tape, 'vgg.conv%d' % ci)
            if ci in VGG_TAPS_AFTER_CONV:
                taps.append(x)
            ci += 1
    return taps


def alex_features(Wa, x):
    """torchvision alexnet.features sliced as lpips.pretrained_networks.alexnet does
    [3P-recall]: returns relu1 .. relu5."""
    taps = []
    for i, (_, _, k, s, p) in enumerate(ALEX_CFG):
        if i in ALEX_POOL_BEFORE:
            x = F.max_pool2d(x, 3, 2)
        x = F.relu(F.conv2d(x, Wa['alex.conv%d.weight' % i], Wa['alex.conv%d.bias' % i],
                            stride=s, padding=p))
        taps.append(x)
    return taps


# torchvision squeezenet1_1.features sliced as lpips.pretrained_networks.squeezenet does [3P-recall]:
# conv(3,64,k3,s2) relu | maxpool(3,2,ceil) fire(64,16,64,64) fire(128,16,64,64) | maxpool fire(128,32,128,128)
# fire(256,32,128,128) | maxpool fire(256,48,192,192) | fire(384,48,192,192) | fire(384,64,256,256) |
# fire(512,64,256,256); the 7 LPIPS taps are the outputs of the 7 slices.  Fire(in, s, e1, e3):
# x -> relu(conv1x1(in,s)) -> cat(relu(conv1x1(s,e1)), relu(conv3x3(s,e3,pad 1)))
SQZ_FIRES = [(64, 16, 64), (128, 16, 64), (128, 32, 128), (256, 32, 128), (256, 48, 192), (384, 48, 192),
             (384, 64, 256), (512, 64, 256)]          # (in, squeeze, expand per branch)
SQZ_POOL_BEFORE_FIRE = (0, 2, 4)                      # 3x3 stride-2 ceil-mode max-pool in front of these fires
SQZ_TAP_AFTER_FIRE = (1, 3, 4, 5, 6, 7)               # taps 1..6 (tap 0 = relu(conv0))
SQZ_CHNS = (64, 128, 256, 384, 384, 512, 512)


def squeeze_features(Ws, x, tape=None):
    """-> the 7 taps of lpips.pretrained_networks.squeezenet [3P-recall].  `tape` (oracle/masks.py): the ReLU
    signs and the winners of the three overlapping max-pools are the discrete decisions of the pass."""
    relu = (lambda v, n: F.relu(v)) if tape is None else tape.relu
    pool = (lambda v, n: F.max_pool2d(v, 3, 2, ceil_mode=True)) if tape is None else tape.maxpool3s2
    x = relu(F.conv2d(x, Ws['squeeze.conv0.weight'], Ws['squeeze.conv0.bias'], stride=2), 'squeeze.conv0')
    taps = [x]
    for i in range(len(SQZ_FIRES)):
        if i in SQZ_POOL_BEFORE_FIRE:
            x = pool(x, 'squeeze.pool%d' % i)
        p = 'squeeze.fire%d.' % i
        q = relu(F.conv2d(x, Ws[p + 'squeeze.weight'], Ws[p + 'squeeze.bias']), p + 'squeeze')
        x = torch.cat([relu(F.conv2d(q, Ws[p + 'expand1x1.weight'], Ws[p + 'expand1x1.bias']), p + 'expand1x1'),
                       relu(F.conv2d(q, Ws[p + 'expand3x3.weight'], Ws[p + 'expand3x3.bias'], padding=1),
                            p + 'expand3x3')], 1)
        if i in SQZ_TAP_AFTER_FIRE:
            taps.append(x)
    return taps


def normalize_tensor(f, eps=1e-10):
    """lpips.normalize_tensor [3P-recall]."""
    norm = torch.sqrt(torch.sum(f ** 2, dim=1, keepdim=True))
    return f / (norm + eps)


def lpips_spatial(Wv, in0, in1, tape=None):
    """lpips.LPIPS(net='vgg', spatial=True).forward(in0, in1) -> [B,1,H,W]  [3P-recall].
    `tape` applies to the features of in0 (the generated image: the differentiated pass)."""
    shift = torch.tensor(LPIPS_SHIFT, dtype=in0.dtype).view(1, 3, 1, 1)
    scale = torch.tensor(LPIPS_SCALE, dtype=in0.dtype).view(1, 3, 1, 1)
    feats = alex_features if 'alex.conv0.weight' in Wv else \
        squeeze_features if 'squeeze.conv0.weight' in Wv else vgg_features
    f0 = feats(Wv, (in0 - shift) / scale, tape) if tape is not None else feats(Wv, (in0 - shift) / scale)
    f1 = feats(Wv, (in1 - shift) / scale)
    val = None
    for kk in range(len(f0)):
        d = (normalize_tensor(f0[kk]) - normalize_tensor(f1[kk])) ** 2
        lin = F.conv2d(d, Wv['lpips.lin%d.weight' % kk])       # [1,C,1,1], no bias
        up = F.interpolate(lin, size=in0.shape[2:], mode='bilinear', align_corners=False)
        val = up if val is None else val + up
    return val


def perceptual_loss(Wv, output, target, weight=None, loss_mask=None, tape=None):
    """PerceptualLoss.__call__ (loss_functions.py:140-148)."""
    return _weighted(lpips_spatial(Wv, output, target, tape), weight, loss_mask)


def projection_loss(Wv, output, target, weight=None, loss_mask=None, beta=10, tape=None):
    """ProjectionLoss.__call__ (loss_functions.py:97-100): rec + beta * per."""
    rec = reconstruction_loss(output, target, weight, loss_mask, tape=tape)
    per = perceptual_loss(Wv, output, target, weight, loss_mask, tape)
    return rec + beta * per
