"""Per-layer roofline of the 1x1 convs of the bench step from a tools/prof_layers.py table:
floor = max(FLOPs / matrix-pipe ceiling, bytes / 6.3 TB/s) as VERDICT round 2 asked -- the ceiling of a
launch is the dense 16-bit MFMA peak / its products per fp32 product (column mm: 419.4 TFLOP/s for
bf16x3, 838.9 for fp16x2, 157.3 for the exact-fp32 MFMA) -- and the memory floor
with the stream rates measured on the part (tools/micro/mem_rate.hip: read 6.5, write 4.5 TB/s;
output bytes = B*H*W*Cout*4, the rest of the algorithmic bytes are reads).
usage: roofline_1x1.py layers.txt > profiles/round3_conv1x1_roofline.txt"""
import sys
rows = []
for l in open(sys.argv[1]):
    if '|' not in l or l.startswith('taps '):
        continue
    a, b = l.split('|')
    k = [int(x) for x in a.split()]
    v = b.split()
    if k[0] != 1:
        continue
    n, ms, tf, gbs = float(v[0]), float(v[1]), float(v[2]), float(v[3])
    per = ms / n                                     # ms per launch
    flops = tf * 1e12 * per * 1e-3
    byts = gbs * 1e9 * per * 1e-3
    B, H, W, Cin, Cout = k[1:6]
    pooled = k[8] == 2                               # arb with the 2x2-summed shortcut
    wr = 4.0 * B * H * W * Cout / (4 if pooled else 1)
    wr = min(wr, byts)
    mm = k[10] if len(k) > 10 else 6
    ceil = 2516.6e12 / mm
    f1 = max(flops / ceil, byts / 6.3e12) * 1e3
    f2 = max(flops / ceil, (byts - wr) / 6.5e12 + wr / 4.5e12) * 1e3
    rows.append((ms, k, n, per, tf, gbs / 1e3, f1, f2, mm, ceil))
print('1x1 conv launches of one BasinCMA inner step (BigGAN-256, 18 candidates): measured vs floors')
print('floor A = max(FLOPs / (2516.6 TFLOP/s / mm), bytes / 6.3 TB/s); floor B = the same with read 6.5 / write 4.5 TB/s;  mm = MFMA products per fp32 product (6 bf16x3, 3 fp16x2, 16 fp32 MFMA)')
print('   B    H   Cin  Cout pro arb sk mm | n/step ms/launch TFLOP/s  TB/s | floor A  x    | floor B  x    | bound')
ta = tb = tm = 0.0
for ms, k, n, per, tf, tbs, f1, f2, mm, ceil in sorted(rows, key=lambda r: -r[0]):
    bound = 'mfma' if (tf * 1e12 * per * 1e-3) / ceil * 1e3 >= f2 * 0.999 else 'hbm'
    print('%4d %4d %5d %5d %3d %3d %2d %2d | %5.1f %8.4f %7.1f %5.2f | %7.4f %5.2f | %7.4f %5.2f | %s' % (
        k[1], k[2], k[4], k[5], k[7], k[8], k[9], mm, n, per, tf, tbs, f1, per / f1, f2, per / f2, bound))
    ta += f1 * n; tb += f2 * n; tm += ms
print('sum: measured %.3f ms/step, floor A %.3f (x%.2f), floor B %.3f (x%.2f)' % (tm, ta, tm / ta, tb, tm / tb))
