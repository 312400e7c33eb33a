"""C-ABI checks that need no GPU: the shared library loads, exports every symbol
include/p2l.h declares, pure-host entry points behave, ctypes structs match the
C layout, and the product path refuses to run without the HIP device."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    so = os.path.join(ROOT, 'pix2latent_amd', 'libp2l_hip.so')
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    from pix2latent_amd import _native as N
    return N.lib()


def header_symbols(names=('p2l.h', 'p2l_test.h')):
    """every function the boundary header (p2l.h) and the test-hook header (p2l_test.h) declare"""
    out = set()
    for name in names:
        txt = open(os.path.join(ROOT, 'include', name)).read()
        txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
        out |= set(re.findall(r'\b(p2l_[a-z0-9_]+)\s*\(', txt))
    return sorted(out)


def test_every_declared_symbol_is_exported(lib):
    from pix2latent_amd import _native as N
    syms = header_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(lib, s), 'libp2l_hip.so does not export %s' % s
    assert sorted(N.EXPORTS) == syms, set(N.EXPORTS) ^ set(syms)
    # the test hooks live in their own header: the drop-in boundary declares none of them
    hooks = header_symbols(('p2l_test.h',))
    assert hooks == sorted(['p2l_selftest_amaxreg', 'p2l_biggan_ws_lookup', 'p2l_projloss_ws_lookup', 'p2l_sg2_ws_lookup',
                            'p2l_sqzloss_ws_lookup',
                            'p2l_mfma_probe'])
    assert not set(hooks) & set(header_symbols(('p2l.h',)))
    # version 101: the sized-struct totals have their OWN name; the name that had carried two signatures is gone
    assert 'p2l_prof_totals' in syms and not [s for s in syms if s.startswith('p2l_prof_end')]


def test_version_and_errors(lib):
    from pix2latent_amd import _native as N
    assert lib.p2l_version() == N.ABI_VERSION == 101
    assert lib.p2l_strerror(0) == b'ok'
    assert b'workspace' in lib.p2l_strerror(-3)
    assert lib.p2l_affine_relu_bwd_nblk(65536) == 256
    assert lib.p2l_l1_loss_nblk(256, 256) == 256


def test_struct_layouts_match_the_compiled_header(tmp_path):
    """sizes and member offsets of the ctypes mirrors against include/p2l.h compiled by gcc (the
    structs that carry pointers next to 32-bit members: padding is where mirrors go wrong)"""
    import shutil
    import subprocess
    from pix2latent_amd import _native as N
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    members = {
        'P2LAmax': ['out', 'outp', 'in', 'in_n', 'next_s', 'next_t', 'next_bstride', 'in_applied'],
        'P2LConvExtra': ['oscale', 'oscale_bstride', 'noise', 'noise_w', 'amax'],
        'P2LArb': ['x', 'x_ld', 's', 't', 'st_bstride', 'skip', 'skip_ld', 'skip_C', 'skip_ups', 'ds', 'dt',
                   'dsdt_bstride', 'partial', 'nomask', 'amax'],
        'P2LConv': ['wfmt', 'form', 'algo_flops', 'w_floats'],
        'P2LProfTotals': ['size', 'count', 'flops', 'ms', 'bytes', 'exec_flops', 'mfma_flops', 'write_bytes',
                          'fam_count', 'fam_ms', 'fam_flops', 'fam_mfma_flops', 'fam_bytes'],
    }
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "p2l.h"', 'int main(void) {']
    for st, ms in members.items():
        src.append('printf("%s %%zu\\n", sizeof(%s));' % (st, st))
        for m in ms:
            src.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, m, st, m))
    src += ['return 0; }']
    c = tmp_path / 'layout.c'
    c.write_text('\n'.join(src))
    exe = tmp_path / 'layout'
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include')
    subprocess.check_call(['gcc', '-I', inc, str(c), '-o', str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for st, ms in members.items():
        ct = getattr(N, st)
        assert int(got[st]) == C.sizeof(ct), st
        for m in ms:
            f = getattr(ct, 'in_' if m == 'in' else m)
            assert int(got['%s.%s' % (st, m)]) == f.offset, (st, m)


def test_prof_totals_fills_what_fits(lib):
    """one p2l_prof_totals for every generation of callers from version 101 on: the library writes the leading `size` bytes of
    its totals and nothing behind them; a struct too short for the counts is refused"""
    from pix2latent_amd import _native as N
    t = N.P2LProfTotals()                            # (nothing was timed: no device needed)
    t.size = N.P2LProfTotals.bytes.offset            # a caller compiled before `bytes` existed
    t.bytes[0] = t.write_bytes[1] = -7.0
    assert lib.p2l_prof_totals(C.byref(t)) == 0
    assert t.size == N.P2LProfTotals.bytes.offset and t.count[0] == 0 and t.ms[1] == 0.0
    assert t.bytes[0] == -7.0 and t.write_bytes[1] == -7.0
    t.size = 8
    assert lib.p2l_prof_totals(C.byref(t)) == -1
    assert lib.p2l_prof_totals(None) == -1


def test_maxima_registry_rules(lib):
    """P2LAmax in the plans: an entry is found only for the same address AND extents, dies when the
    tensor is re-written or dropped or when its ring set is handed out again, and there are never
    more live entries than slot sets (host logic of csrc/p2l_plan.hip, run through a test hook)"""
    assert lib.p2l_selftest_amaxreg() == 0


def test_struct_layouts_match_c(lib):
    """sizes implied by include/p2l.h on LP64"""
    from pix2latent_amd import _native as N
    assert C.sizeof(N.P2LConv) == 23 * 4 + 4 + 8 + 8      # (+ form; padded to the double; + w_floats)
    assert C.sizeof(N.P2LGemm) == 7 * 4 + 4 + 3 * 8 + 4 * 4
    assert C.sizeof(N.P2LGenBlock) == 7 * 4 + 4 + 14 * 8
    assert C.sizeof(N.P2LVggLpips) == (13 * 3 + 5 + 2) * 8 + 8
    assert C.sizeof(N.P2LLossCache) == 11 * 8
    assert C.sizeof(N.P2LGConv) == 16 * 4
    assert C.sizeof(N.P2LAlexLpips) == (5 * 4 + 2) * 8


def test_host_side_planning_calls(lib):
    """shape validation / workspace sizing run on the host only"""
    from pix2latent_amd import _native as N
    d = N.P2LConv()
    d.B, d.H, d.W, d.Cin, d.Cout, d.taps = 9, 4, 4, 512, 512, 9
    d.splitk = 1
    s = lib.p2l_conv_suggest_splitk(C.byref(d))
    assert 1 < s <= 32                       # 4x4 layers split K (shape-only rule: 8 slices)
    d.splitk = s
    assert lib.p2l_conv_workspace_bytes(C.byref(d)) == s * 9 * 16 * 512 * 4
    d.H = d.W = 256
    d.Cin = d.Cout = 64
    assert lib.p2l_conv_suggest_splitk(C.byref(d)) == 1
    # the slice count follows the LAYER, never the batch (round 5: a candidate's bits must not depend on
    # how many others share its launch -- 1 GPU x 18 vs 8 GPUs x 2-3): every conv shape of BigGAN-deep-256,
    # VGG16 and StyleGAN2, both weight formats, forward and gradient orientation
    shapes = set()
    for H, cin, cout in ((4, 2048, 2048), (8, 2048, 2048), (16, 2048, 1024), (16, 1024, 1024),
                         (32, 1024, 1024), (64, 1024, 512), (64, 512, 512), (128, 512, 256),
                         (128, 256, 256), (256, 256, 128)):
        mid = cin // 4
        for hh in (H, max(4, H // 2)):
            shapes |= {(1, hh, cin, mid), (1, hh, mid, cin), (9, hh, mid, mid), (1, hh, mid, cout), (1, hh, cout, mid)}
    for H, c in ((256, 64), (128, 128), (64, 256), (32, 512), (16, 512), (4, 512), (8, 512)):
        shapes |= {(9, H, c, c), (9, H, c, max(c // 2, 32)), (9, H, max(c // 2, 32), c)}
    n_split = 0
    for taps, H, cin, cout in sorted(shapes):
        for ups in (0, 1):
            if ups and taps == 1:
                continue
            got = set()
            for B in (1, 2, 3, 5, 9, 18, 22, 32):
                d = N.P2LConv()
                d.B, d.H, d.W, d.Cin, d.Cout, d.taps, d.ups = B, H, H, cin, cout, taps, ups
                d.wfmt = 2 if taps == 9 else 3
                d.x_ld, d.n_store, d.y_ld = cin, cout, cout
                got.add(lib.p2l_conv_suggest_splitk(C.byref(d)))
            assert len(got) == 1, ((taps, H, cin, cout, ups), got)
            n_split += got.pop() > 1
    assert n_split >= 10
    # ... and the sub-pixel input-gradient form of StyleGAN2's transposed convs (ups 3, ext 1; round 6): slices from
    # the shape only, at 8^2 ... 32^2, none where the grid fills the chip; workspace and partial counts follow
    sliced = {}
    for H, c in ((8, 512), (16, 512), (32, 512), (64, 512), (128, 256), (512, 64), (1024, 32)):
        got = set()
        for B in (1, 3, 9, 22, 32):
            d = N.P2LConv()
            d.B, d.H, d.W, d.Cin, d.Cout, d.taps, d.ups, d.ext = B, H, H, c, c, 9, 3, 1
            d.wfmt, d.x_ld, d.n_store, d.y_ld = 2, c, c, c
            got.add(lib.p2l_conv_suggest_splitk(C.byref(d)))
        assert len(got) == 1, (H, c, got)
        sliced[H] = got.pop()
    assert sliced[8] > 1 and sliced[16] > 1 and sliced[32] > 1 and sliced[128] == sliced[512] == sliced[1024] == 1
    d = N.P2LConv()
    d.B, d.H, d.W, d.Cin, d.Cout, d.taps, d.ups, d.ext = 3, 16, 16, 512, 512, 9, 3, 1
    d.wfmt, d.x_ld, d.n_store, d.y_ld = 2, 512, 512, 512
    d.splitk = sliced[16]
    assert lib.p2l_conv_arb_nblk_ws(C.byref(d)) == 4 * 4          # quads of the LOW-res 8x8 grid the finish kernel walks
    assert lib.p2l_conv_arb_split_fusable(C.byref(d)) == 1
    assert lib.p2l_conv_amax_slots(C.byref(d)) == 0                # (a sliced sub-pixel launch leaves no maxima)
    assert lib.p2l_conv_workspace_bytes(C.byref(d)) >= 3 * 64 * 4 + sliced[16] * 3 * 8 * 8 * 512 * 4
    d.ext = 0                                                     # BigGAN's nearest-upsample gradients: never sliced
    assert lib.p2l_conv_suggest_splitk(C.byref(d)) == 1
    # the noise relayout validates its descriptor on the host
    m = N.P2LStyleGAN2()
    fake = C.c_void_p(4096)
    assert lib.p2l_sg2_noise_relayout(C.byref(m), fake, fake, 2, 1, None) == -1        # no layers
    assert lib.p2l_sg2_noise_relayout(None, fake, fake, 2, 1, None) == -1
    assert lib.p2l_projloss_ws_bytes(2, 256, 256) > 0
    assert lib.p2l_projloss_ws_bytes(2, 100, 100) == 0          # not a power of two
    # P2LConv.w_floats (version 101): a sub-pixel launch of a P2L_WFMT_BF16X3W model reads
    # the fp16 x 2 image BEHIND the bf16 x 3 one; a buffer of the bf16 x 3-only length is refused on the
    # host, before any launch (fake non-null device pointers: nothing is dereferenced)
    d = N.P2LConv()
    d.B, d.H, d.W, d.Cin, d.Cout, d.taps, d.ups = 2, 64, 64, 128, 128, 9, 2
    d.wfmt, d.x_ld, d.n_store, d.y_ld, d.yp_ld, d.alpha, d.splitk = 2, 128, 128, 128, 128, 1.0, 1
    short = lib.p2l_packed_subpix_weight_floats(128, 128, 1)      # what p2l_pack_conv_weight_subpix_bf3 fills
    full = lib.p2l_packed_subpix_weight_floats(128, 128, 2)
    assert short < full
    d.w_floats = short
    fake = C.c_void_p(4096)
    assert lib.p2l_conv_fwd(C.byref(d), fake, fake, None, None, None, None, None, fake, None,
                            None, C.c_size_t(0), None) == -1
    d.w_floats = -5
    assert lib.p2l_conv_fwd(C.byref(d), fake, fake, None, None, None, None, None, fake, None,
                            None, C.c_size_t(0), None) == -1
    # invalid arguments are rejected before any launch
    assert lib.p2l_conv_fwd(None, None, None, None, None, None, None, None, None, None,
                            None, C.c_size_t(0), None) == -1


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from pix2latent_amd import _native as N
    from pix2latent_amd.model.biggan import BigGAN
    from pix2latent_amd.utils import synthetic as S
    with pytest.raises(N.NativeError, match='no CPU fallback'):
        BigGAN(weights={'embeddings.weight': torch.zeros(128, 1000),
                        'generator.gen_z.weight': torch.zeros(8, 256),
                        'generator.gen_z.bias': torch.zeros(8)}, device='cpu')
    import pix2latent_amd.loss_functions as LF
    with pytest.raises(NotImplementedError):
        LF.ProjectionLoss(lpips_net='resnet')
    from pix2latent_amd.model.stylegan2 import StyleGAN2
    with pytest.raises(N.NativeError, match='no CPU fallback'):
        StyleGAN2(weights={}, size=64, device='cpu')


def test_library_has_no_switches_of_its_own():
    """re-entrant, no global mutable state.  Kernel-form choices travel per call
    (P2LConv.form, model descriptor flags), environment variables are read in Python only, the
    deferred-reduction / last-error state is per host thread, and the one process-wide object --
    the opt-in launch profiler, which must also see the launches of torch's autograd thread --
    sits behind a mutex (p2l_conv.hip prof())."""
    csrc = os.path.join(ROOT, 'pix2latent_amd', 'csrc')
    for f in sorted(os.listdir(csrc)):
        if not f.endswith(('.hip', '.cpp', '.h')):
            continue
        src = open(os.path.join(csrc, f)).read()
        src = re.sub(r'#ifdef P2L_LAB.*?#endif', '', src, flags=re.S)      # measurement build only
        assert 'getenv' not in src, f
        # namespace-scope definitions of g_* variables (declarations start in column 0)
        for m in re.finditer(r'^[A-Za-z_][A-Za-z0-9_:<>\s]*?\bg_[a-z0-9_]+\s*(=|;|\{)', src, flags=re.M):
            assert 'thread_local' in m.group(0), '%s: %s' % (f, m.group(0))
    hdr = open(os.path.join(ROOT, 'include', 'p2l.h')).read()
    assert 'p2l_set_' not in hdr


def test_oracle_not_imported_by_product():
    """the oracle is test infrastructure: nothing under pix2latent_amd/ or the
    drop-in alias may import it."""
    for base in ('pix2latent_amd', 'pix2latent'):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith('.py'):
                    src = open(os.path.join(dp, f)).read()
                    assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
