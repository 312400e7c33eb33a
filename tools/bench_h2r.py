"""conv_h2r_kernel (register-resident weights, csrc/p2l_h2r.hip) against the chunked direct fp16 x 2 kernel on the
64 -> 64 channel layers of the bench step, per launch (hipEvents over 20 launches, maxima handed in: no amax pass)."""
import os
import sys
import math

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402


def main():
    from pix2latent_amd import _native as N, ops as O
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1)
    w = torch.randn(64, 64, 3, 3, generator=g) / math.sqrt(576)
    wp = O.pack_conv_weight(w.to(dev), 9, 64, 64, wfmt=2)
    for B, H in ((18, 256), (9, 256), (3, 256), (2, 256), (18, 128), (9, 128), (2, 128), (3, 512), (3, 1024)):
        x = torch.randn(B, H, H, 64, generator=g).to(dev)
        am = x.abs().reshape(B, 2048, -1).amax(dim=2).contiguous()      # (as many partials as a 256^2 producer leaves)
        s = (0.5 + torch.rand(B, 64, generator=g)).to(dev)
        t = (0.1 * torch.randn(B, 64, generator=g)).to(dev)
        row = []
        for name, kw in (('plain', {}), ('pro+relu', dict(pro=N.PRO_AFFINE_RELU, pro_s=s, pro_t=t, pro_bstride=64, act=N.ACT_RELU))):
            ms = []
            for form in (N.FORM_NO_WINO, N.FORM_NO_WINO | N.FORM_NO_H2R):
                for _ in range(3):
                    O.conv(x, wp, B, H, H, 64, 64, 9, wfmt=2, amax_in=am, form=form, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                # (time the launches only: output buffers are allocated by the wrapper, outside the events would be
                #  better, but the allocator serves them from its cache without a device call)
                e0.record()
                for _ in range(20):
                    O.conv(x, wp, B, H, H, 64, 64, 9, wfmt=2, amax_in=am, form=form, **kw)
                e1.record()
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1) / 20)
            fl = 2.0 * B * H * H * 64 * 64 * 9
            by = 4.0 * B * H * H * 64 * 2
            row.append('%s: resident %.3f ms (%.0f TF, %.2f TB/s) chunked %.3f ms  %.2fx' % (
                name, ms[0], fl / ms[0] / 1e9, by / ms[0] / 1e9, ms[1], ms[1] / ms[0]))
        print('%2d x %4d^2 64->64  ' % (B, H) + ' | '.join(row), flush=True)


if __name__ == '__main__':
    main()
