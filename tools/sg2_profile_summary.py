"""profiles/round3_sg2_*: appends to the conv per-layer table of tools/step_sg2_one.py the HBM-bound
StyleGAN2 kernels of the same rocprofv3 run (blur, styled activation backward, rgb upsampling,
max-pool backward) with their aggregate GB/s: analytic bytes of one step / rocprofv3 time per step.
usage: sg2_profile_summary.py kernel_stats.csv layers.txt size batch steps_in_run out.txt"""
import csv, sys
stats, layers, size, B, nsteps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 512, 128: 256, 256: 128, 512: 64, 1024: 32}
res = [r for r in (8, 16, 32, 64, 128, 256, 512, 1024) if r <= size]
fam = {}
total = 0.0
for row in csv.DictReader(open(stats)):
    name, t, calls = row['Name'], float(row['TotalDurationNs']), int(row['Calls'])
    total += t
    for key in ('blur_fwd', 'blur_bwd', 'styled_act_bwd', 'rgb_up_fwd', 'rgb_up_bwd', 'maxpool2_bwd', 'wino16s', 'wino_conv',
                'conv_mfma_kernel<4', 'conv_mfma_kernel<9', 'conv_mfma_kernel<1', 'linear_fwd', 'lpips_tap', 'gconv'):
        if key in name:
            f = fam.setdefault(key, [0.0, 0])
            f[0] += t; f[1] += calls
            break
blur = sum(4.0 * B * ch[r] * ((r + 2) ** 2 + r * r) for r in res)
act = sum(3 * 4.0 * B * r * r * ch[r] * (2 if r > 4 else 1) for r in [4] + res)
rgb = sum(4.0 * B * 16 * (r * r + (r // 2) ** 2) for r in res)
vgg_pool = sum(4.0 * B * c * (3 * s * s // 4 + s * s) for s, c in ((size, 64), (size // 2, 128), (size // 4, 256), (size // 8, 512)))
bytes_per_step = {'blur_fwd': blur, 'blur_bwd': blur, 'styled_act_bwd': act, 'rgb_up_fwd': rgb, 'rgb_up_bwd': rgb,
                  'maxpool2_bwd': vgg_pool}
lines = open(layers).read().rstrip('\n').split('\n')
lines.append('')
lines.append('rocprofv3 --kernel-trace --stats of the same process (%d steps incl. warm-up; the model constructor'
             % nsteps)
lines.append("also runs: linear_fwd_kernel = 4096 mapping-network passes for the mean latent), per family:")
lines.append('%-22s %8s %10s %8s %10s' % ('kernel family', 'calls', 'ms / step', 'share', 'GB/s (analytic bytes of a step / time)'))
step_total = sum(v[0] for k, v in fam.items() if k != 'linear_fwd') / nsteps
for key, (t, calls) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    if key == 'linear_fwd':
        continue
    gbs = '%.0f' % (bytes_per_step[key] / (t / nsteps)) if key in bytes_per_step else ''
    lines.append('%-22s %8d %10.3f %7.1f%% %10s' % (key, calls, t / nsteps * 1e-6, 100 * t / nsteps / step_total, gbs))
dom = max((k for k in fam if k != 'linear_fwd'), key=lambda k: fam[k][0])
lines.append('dominant: %s, %.1f %% of the kernel time of a step' % (dom, 100 * fam[dom][0] / nsteps / step_total))
open(out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[-14:]))
