"""LDS bank conflicts of the A-fragment reads of conv_h2_kernel (csrc/p2l_h2.hip) by patch-line pitch.

Rows are 64 B = 4 x 16-byte chunks, physical chunk = logical ^ ((row >> 2) & 3) ^ (((row >> 1) & 1) << 1)
(the last term makes the ds_write_b64 of the staging conflict free; it is constant over a read lane group's
rows with equal row & 3, so the read analysis is unchanged).  A ds_read_b128 is
served in four lane groups of 16 ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32,
/opt/skills/guides/MI355X_MICROARCH.md, LDS); a group takes one LDS cycle when its 16 addresses hit
16 different 16-byte bank windows.  The MFMA M index -> pixel map is the 2x2-quad order of the conv
kernels.  Prints LDS cycles per lane group (1.0 = conflict free) for every tile shape and pitch."""
import itertools

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
          [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def rows_of_wave(wave, tw_log, th_log, HP):
    TW, TH = 1 << tw_log, 1 << th_log
    HH = TH + 2
    out = []
    for l31 in range(32):
        i = wave * 32 + l31
        Q, s = i >> 2, i & 3
        qx = Q & ((TW >> 1) - 1)
        qy = (Q >> (tw_log - 1)) & ((TH >> 1) - 1)
        tb = Q >> (tw_log + th_log - 2)
        out.append((tb * HH + 2 * qy + (s >> 1)) * HP + 2 * qx + (s & 1))
    return out


def cycles(rows, chunk):
    worst = 0.0
    tot = 0
    for g in GROUPS:
        windows = {}
        for l in g:
            r = rows[l]
            w = ((r * 64 + ((chunk ^ ((r >> 2) & 3) ^ (((r >> 1) & 1) << 1)) * 16)) // 16) % 16     # 16-byte window of 64 banks x 4 B
            windows.setdefault(w, set()).add(r)
        tot += max(len(v) for v in windows.values())
    return tot / len(GROUPS)


for tw_log, th_log in ((4, 3), (3, 4), (3, 3), (2, 2), (2, 3)):
    TW = 1 << tw_log
    best = []
    for HP in range(TW + 2, TW + 2 + 17):
        c = 0.0
        for wave in range(4):
            rows = rows_of_wave(wave, tw_log, th_log, HP)
            for tap_off in (0, 1, 2, HP, HP + 1, 2 * HP + 2):
                c = max(c, cycles([r + tap_off for r in rows], 0), cycles([r + tap_off for r in rows], 1))
        best.append((c, HP))
    print('tile %2dx%-2d (WxH): ' % (TW, 1 << th_log) + '  '.join('hp %d: %.1f' % (hp, c) for c, hp in best))
