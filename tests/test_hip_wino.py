"""GPU tests of the Winograd F(2x2,3x3) kernels of the exact-f32 path (ctrlhair_amd/csrc/conv_wino.h): the ResBlock 3x3 convs
(/root/reference/sean_codes/models/networks/architecture.py:82-91) and the SPADE gamma/beta conv + style convs over the boundary
quads (normalization.py:117-153,172-173,249-257).

The golden / oracle tests of test_hip_sean_generator.py already run with them on (option sean.wino = 1 is the default of the
exact-f32 path); here the library is compared with ITSELF on the direct evaluation (sean.wino = 0) at a tolerance far inside the
parity bound, on the shapes that exercise the kernels' edge cases: a single 16-channel row tile (ngf = 16: the second row tile
of a pair does not exist), levels below 32 pixels (direct kernels), S < max_size, ragged batches, labels >= 19 at tile borders,
label maps with no interior pixel, the gather mode of the ACE kernel (the default) and both tile heights of its tile mode."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gen(sd, mb, ms, wino, extra=None):
    from ctrlhair_amd.sean.generator import SeanGenerator
    opts = {'sean.wino': wino}
    opts.update(extra or {})
    return SeanGenerator(0, f16x3=0, options=opts).load_state_dict(sd, max_batch=mb, max_size=ms)


def _run(gen, labels, codes, noise):
    dev = gen.device
    out = gen.generate(torch.from_numpy(labels).to(dev), torch.from_numpy(codes).to(dev), torch.from_numpy(noise).to(dev))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _label_sets(B, S):
    from ctrlhair_amd import procedural as P
    sets = {'blocky': P.blocky_labels(B, S, grid=8), 'face': np.stack([P.face_like_labels(S, 40 + b) for b in range(B)])}
    diag = (np.add.outer(np.arange(S), np.arange(S)) % 19).astype(np.uint8)          # no interior pixel at any resolution
    sets['diag'] = np.repeat(diag[None], B, 0)
    edge = P.blocky_labels(B, S, grid=4, seed=5).copy()                              # 'no class' labels across tile borders
    edge[:, 30:34, :] = 255
    edge[:, :, 62:66] = 19
    edge[:, S // 2:, : S // 2] = 200
    sets['noclass_at_tile_borders'] = edge
    sets['one_region'] = np.full((B, S, S), 13, np.uint8)                            # interior everywhere but the image frame
    return sets


@pytest.mark.parametrize('S,th', [(64, 0), (128, 16), (128, 32)])
def test_winograd_equals_direct_tiny(hip_lib, S, th):
    """ngf = 16, max_size 128: one row tile per ACE at the last level, levels of 2 ... 16 pixels on the direct kernels."""
    from ctrlhair_amd import procedural as P
    ngf, B = 16, 3
    sd = P.sean_state_dict(0, ngf)
    direct = _gen(sd, B, 128, 0)
    wino = _gen(sd, B, 128, 1, {'sean.wino_gather': 0, 'sean.wino_th': th} if th else None)      # th = 0: gather mode
    codes, noise = P.style_codes(B, seed=3), P.noise_planes(B, S, ngf, seed=4)
    for name, lab in _label_sets(B, S).items():
        a, b = _run(direct, lab, codes, noise), _run(wino, lab, codes, noise)
        assert np.isfinite(b).all()
        d = float(np.abs(a - b).max())
        print(f'ngf16 S={S} th={th} {name}: max |winograd - direct| = {d:.3e}')
        assert d <= 2e-5, name
        assert np.array_equal(b, _run(wino, lab, codes, noise)), 'repeated call differs'
    one = _run(wino, _label_sets(B, S)['face'][1:2], codes[1:2], noise[1:2])          # ragged batch: one sample of three
    assert np.abs(one[0] - _run(wino, _label_sets(B, S)['face'], codes, noise)[1]).max() <= 2e-5
    direct.handle.close()
    wino.handle.close()


def test_winograd_equals_direct_ngf64(hip_lib):
    """ngf = 64 at 256^2 (all four Winograd levels 32 ... 256, styled and unstyled ACEs, 64 row tiles at C = 1024) and against
    the oracle."""
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    ngf, S, B = 64, 256, 2
    sd = P.sean_state_dict(0, ngf)
    direct, wino, tile = _gen(sd, B, S, 0), _gen(sd, B, S, 1), _gen(sd, B, S, 1, {'sean.wino_gather': 0})
    codes, noise = P.style_codes(B, seed=8), P.noise_planes(B, S, ngf, seed=9)
    sets = _label_sets(B, S)
    for name in ('face', 'blocky', 'noclass_at_tile_borders', 'diag'):
        a, b = _run(direct, sets[name], codes, noise), _run(wino, sets[name], codes, noise)
        d = float(np.abs(a - b).max())
        print(f'ngf64 {name}: max |winograd - direct| = {d:.3e}')
        assert d <= 2e-5, name
        # gather mode (tasks of 64 consecutive boundary quads) against tile mode (tasks per tile): the same arithmetic per quad
        assert np.array_equal(b, _run(tile, sets[name], codes, noise)), f'{name}: gather and tile mode differ'
    tile.handle.close()
    ref = O.generator_forward(O.to_torch(sd), sets['face'][:1], codes[:1], noise[:1], ngf).numpy()
    assert np.abs(_run(wino, sets['face'][:1], codes[:1], noise[:1]) - ref).max() <= 1e-3
    # executed-FLOP accounting of the plain convs: 16 / 36 of the dense count on the Winograd levels
    wino.handle.profile_enable(True)
    _run(wino, sets['face'], codes, noise)
    wino.handle.profile_enable(False)
    plain = wino.handle.profile_read(0)
    wino.handle.profile_read(-1)
    assert 16 / 36 - 1e-9 <= plain['flops_executed'] / plain['flops'] <= 0.62       # (+ the direct 1x1 shortcuts and 8 / 16-pixel levels)
    direct.handle.close()
    wino.handle.close()


def test_winograd_f4x4_equals_direct(hip_lib):
    """Option sean.wino = 2 (the default): the ResBlock 3x3 convs from 32 x 32 pixels as Winograd F(4x4,3x3) (conv_wino4.h; architecture.py:82-91),
    everything else as with 1.  Its transforms multiply by 2, 4, 5, 8 and amplify rounding where F(2x2,3x3) only adds: measured here
    against the direct evaluation (sean.wino = 0), against F(2x2,3x3) and against the oracle -- ngf = 64 at 256^2 (levels 32 ... 256 of
    64 ... 1024 channels), ngf = 16 with a ragged row tile (Cout = 16 / 48), S = 96 (tiles of 32 on a 96-pixel level)."""
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    for ngf, S, B in ((64, 256, 2), (16, 128, 3), (16, 96, 2)):
        sd = P.sean_state_dict(0, ngf)
        # (sean.wino4_force: these small calls have fewer F(4x4) tasks than CUs, where the library would pick the F(2x2) kernels by itself)
        direct, f2, f4 = _gen(sd, B, S, 0), _gen(sd, B, S, 1), _gen(sd, B, S, 2, {'sean.wino4_force': 1})
        codes, noise = P.style_codes(B, seed=71), P.noise_planes(B, S, ngf, seed=72)
        sets = _label_sets(B, S)
        for name in ('face', 'blocky', 'diag'):
            a, b, c = _run(direct, sets[name], codes, noise), _run(f2, sets[name], codes, noise), _run(f4, sets[name], codes, noise)
            d4, d2 = float(np.abs(a - c).max()), float(np.abs(a - b).max())
            print(f'ngf{ngf} S={S} {name}: max |F(4x4) - direct| = {d4:.3e}, |F(2x2) - direct| = {d2:.3e}')
            assert np.isfinite(c).all() and d4 <= 2e-4, (ngf, S, name)
            assert np.array_equal(c, _run(f4, sets[name], codes, noise)), 'repeated call differs'
        if ngf == 64:
            ref = O.generator_forward(O.to_torch(sd), sets['face'][:1], codes[:1], noise[:1], ngf).numpy()
            d = float(np.abs(_run(f4, sets['face'][:1], codes[:1], noise[:1]) - ref).max())
            print(f'ngf64: max |F(4x4) - oracle| = {d:.3e}')
            assert d <= 1e-3
            f4.handle.profile_enable(True)
            _run(f4, sets['face'], codes, noise)
            f4.handle.profile_enable(False)
            plain = f4.handle.profile_read(0)
            f4.handle.profile_read(-1)
            assert 0.25 - 1e-9 <= plain['flops_executed'] / plain['flops'] <= 0.45      # 36 / 144 on the F(4x4) levels (+ 1x1 shortcuts, 8 / 16-pixel levels)
        for g in (direct, f2, f4):
            g.handle.close()


def test_f4x4_task_count_rule_only_changes_the_kernels(hip_lib):
    """Small calls (fewer tasks of 32 x 32 pixels than CUs) take the F(2x2,3x3) kernels on their own (wino4_pays, conv_wino4.h): the result of
    a single-image call at the default options equals the sean.wino = 1 result bit for bit where every layer falls back, and stays within the
    F(4x4) tolerance of the forced F(4x4) evaluation."""
    from ctrlhair_amd import procedural as P
    ngf, S, B = 64, 128, 1
    sd = P.sean_state_dict(0, ngf)
    auto, f2, forced = _gen(sd, B, S, 2), _gen(sd, B, S, 1), _gen(sd, B, S, 2, {'sean.wino4_force': 1})
    codes, noise = P.style_codes(B, seed=5), P.noise_planes(B, S, ngf, seed=6)
    lab = _label_sets(B, S)['face']
    a, b, c = _run(auto, lab, codes, noise), _run(f2, lab, codes, noise), _run(forced, lab, codes, noise)
    assert np.array_equal(a, b), 'a 128 x 128 single image has at most 16 x 2 ... 1 x 32 tasks per layer: every layer falls back'
    d = float(np.abs(a - c).max())
    print(f'B=1 S=128: max |auto - forced F(4x4)| = {d:.3e}')
    assert 0 < d <= 2e-4
    for g in (auto, f2, forced):
        g.handle.close()


@pytest.mark.parametrize('ngf,S,B', [(64, 256, 3), (64, 512, 2), (16, 128, 5)])
def test_pretransformed_input_route_equals_the_in_kernel_transform(hip_lib, ngf, S, B):
    """Option sean.wino4v (default 1; conv_wino4v.h): F(4x4,3x3) layers with >= 512 GEMM rows at <= 64 pixels read V = B^T d B from an
    extra pass instead of transforming their patches in every row tile -- the same wino4_in1d sequence, the same order of products, the
    same epilogues: the images must be IDENTICAL, with the route on the plain convs (G_middle / up_0 at ngf = 64), on the SPADE / style
    convs (every F(4x4) ACE level with C >= 256; ngf = 16 has none: the option must change nothing there) and with styled and unstyled
    ACEs, ragged batches and label maps with no interior pixel.  architecture.py:82-91, normalization.py:249-257."""
    from ctrlhair_amd import procedural as P
    sd = P.sean_state_dict(0, ngf)
    on, off = _gen(sd, B, S, 2, {'sean.wino4_force': 1, 'sean.wino4v': 1}), _gen(sd, B, S, 2, {'sean.wino4_force': 1, 'sean.wino4v': 0})
    codes, noise = P.style_codes(B, seed=81), P.noise_planes(B, S, ngf, seed=82)
    sets = _label_sets(B, S)
    for name in ('face', 'diag', 'noclass_at_tile_borders'):
        a, b = _run(on, sets[name], codes, noise), _run(off, sets[name], codes, noise)
        assert np.isfinite(a).all()
        assert np.array_equal(a, b), f'{name}: max |V route - in-kernel transform| = {np.abs(a - b).max():.3e}'
    one = _run(on, sets['face'][B - 1:], codes[B - 1:], noise[B - 1:])        # a smaller batch on the same handle (V image sized for max_batch)
    assert np.array_equal(one, _run(off, sets['face'][B - 1:], codes[B - 1:], noise[B - 1:]))
    if ngf == 64:       # the route is actually taken: its kernels show up in the executed-FLOP accounting unchanged, and the pass is timed with the conv
        on.handle.profile_enable(True)
        _run(on, sets['face'], codes, noise)
        on.handle.profile_enable(False)
        plain = on.handle.profile_read(0)
        on.handle.profile_read(-1)
        assert 0.25 - 1e-9 <= plain['flops_executed'] / plain['flops'] <= 0.45
    for g in (on, off):
        g.handle.close()


@pytest.mark.parametrize('ngf,S,B', [(64, 256, 3), (64, 512, 2), (16, 256, 4)])
def test_straight_edge_pixels_equal_the_boundary_conv(hip_lib, ngf, S, B):
    """Option sean.edge (default 1; ace_sparse.h): a boundary pixel whose 5x5 label neighbourhood is five uniform columns (rows) A^s B^(5-s)
    gets gamma / beta from the per-code row of its ACE's table (+ three column / row sums of the style LUT) in the interior pass instead of
    from the boundary conv -- on the levels of 128 pixels and more.  The same real number in another association: images against the
    conv evaluation of the same library (sean.edge = 0) at 1e-5 and against the direct evaluation (sean.wino = 0), on label maps with
    straight edges (blocky), curved ones (face-like), 'no class' labels along tile borders, no straight edge at all (diag) and a single
    region; the reduction must actually apply (fewer executed FLOPs) where straight edges exist.  normalization.py:117-153,172-187,249-257."""
    from ctrlhair_amd import procedural as P
    sd = P.sean_state_dict(0, ngf)
    on, off, direct = _gen(sd, B, S, 2, {'sean.edge': 1}), _gen(sd, B, S, 2, {'sean.edge': 0}), _gen(sd, B, S, 0)
    codes, noise = P.style_codes(B, seed=91), P.noise_planes(B, S, ngf, seed=92)
    sets = _label_sets(B, S)
    for name in ('blocky', 'face', 'noclass_at_tile_borders', 'diag', 'one_region'):
        a, b, c = _run(on, sets[name], codes, noise), _run(off, sets[name], codes, noise), _run(direct, sets[name], codes, noise)
        d, dd = float(np.abs(a - b).max()), float(np.abs(a - c).max())
        print(f'ngf{ngf} S={S} {name}: max |edge rows - boundary conv| = {d:.3e}, |edge rows - direct| = {dd:.3e}')
        assert np.isfinite(a).all() and d <= 1e-5 and dd <= 2e-4, (name, d, dd)
        assert np.array_equal(a, _run(on, sets[name], codes, noise)), 'repeated call differs'
    ex = {}
    for g, key in ((on, 1), (off, 0)):
        g.handle.profile_enable(True)
        _run(g, sets['blocky'], codes, noise)
        g.handle.profile_enable(False)
        ex[key] = g.handle.profile_read(1)['flops_executed']
        g.handle.profile_read(-1)
    print(f'SPADE conv FLOPs executed on blocky labels: {ex[0]:.3e} -> {ex[1]:.3e}')
    assert ex[1] < 0.8 * ex[0]
    for g in (on, off, direct):
        g.handle.close()


@pytest.mark.parametrize('S,mb', [(64, 8), (256, 9)])
def test_grouped_style_luts_equal_per_ace_launches(hip_lib, S, mb):
    """Exact-f32 path, more than 64 (sample, label) columns: the style LUTs of all styled ACEs from ONE grouped GEMM launch
    (conv_pw.h, operands swapped; option sean.lut_grouped, the default) against one launch of the generic 1x1 kernel per ACE
    (normalization.py:117-153,172-173 conv_gamma / conv_beta of the projected codes).  S = 64 runs in the full run-ahead mode of
    small jobs, 9 x 256^2 in the large-job mode; ragged batches exercise the partial row groups (N = 95: 3 of 4 row tiles) and the
    fall-back for N <= 64, and stale projections of a larger earlier call must not leak into a smaller one."""
    from ctrlhair_amd import procedural as P
    ngf = 64
    sd = P.sean_state_dict(0, ngf)
    grouped, single = _gen(sd, mb, S, 1), _gen(sd, mb, S, 1, {'sean.lut_grouped': 0})
    codes, noise = P.style_codes(mb, seed=21), P.noise_planes(mb, S, ngf, seed=22)
    lab = np.stack([P.face_like_labels(S, 60 + b) for b in range(mb)])
    full = _run(grouped, lab, codes, noise)
    d = float(np.abs(full - _run(single, lab, codes, noise)).max())
    print(f'S={S} B={mb}: max |grouped - per-ACE| = {d:.3e}')
    assert np.isfinite(full).all() and d <= 2e-6
    for B in (5, 3):
        a, b = _run(grouped, lab[:B], codes[:B], noise[:B]), _run(single, lab[:B], codes[:B], noise[:B])
        assert float(np.abs(a - b).max()) <= 2e-6, B
        assert float(np.abs(a - full[:B]).max()) <= 2e-5, B          # (batch composition changes kernel choices, not the maths)
    assert np.array_equal(full, _run(grouped, lab, codes, noise)), 'repeated call differs'
    grouped.handle.close()
    single.handle.close()


def test_size_with_mixed_levels_against_the_oracle(hip_lib):
    """S = 96 at ngf = 64: resolution levels of 3 ... 96 pixels, of which only the last fits the Winograd tiles (the others run on the
    direct kernels, the 48-pixel level with the SPADE-interior reduction), five samples = 95 (sample, label) columns through the
    grouped LUT build (three of four row tiles in its one row group) -- against the oracle."""
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    ngf, S, B = 64, 96, 5
    sd = P.sean_state_dict(0, ngf)
    g = _gen(sd, B, S, 1)
    codes, noise = P.style_codes(B, seed=31), P.noise_planes(B, S, ngf, seed=32)
    lab = np.stack([P.face_like_labels(S, 90 + b) for b in range(B)])
    out = _run(g, lab, codes, noise)
    assert np.isfinite(out).all()
    ref = O.generator_forward(O.to_torch(sd), lab[3:5], codes[3:5], noise[3:5], ngf).numpy()
    d = float(np.abs(out[3:5] - ref).max())
    print(f'S=96 ngf=64: max |hip - oracle| = {d:.3e}')
    assert d <= 1e-3
    g.handle.close()


def test_hidden_activations_on_boundary_quad_patches_only(hip_lib):
    """Option sean.hidden_wq (default 1): from 128 pixels the SPADE hidden activations (normalization.py:249-251) and the one-hot
    planes come from one persistent kernel that writes only the 64-byte pixel groups some boundary quad's 4 x 4 patch touches;
    0 = the label-table kernel over every pixel.  Same sums in the same order, and no unwritten pixel is ever read: the images
    must be bit-identical -- also when a call with other labels ran in between (stale hidden activations in the skipped pixels)."""
    from ctrlhair_amd import procedural as P
    ngf, B, S = 16, 3, 256
    sd = P.sean_state_dict(0, ngf)
    g = _gen(sd, B, S, 1)
    codes, noise = P.style_codes(B, seed=51), P.noise_planes(B, S, ngf, seed=52)
    sets = _label_sets(B, S)
    outs = {}
    for name in ('face', 'blocky', 'one_region', 'noclass_at_tile_borders', 'diag'):
        outs[name] = _run(g, sets[name], codes, noise)
    g.handle.set_option('sean.hidden_wq', 0)
    for name in ('diag', 'noclass_at_tile_borders', 'one_region', 'blocky', 'face'):
        ref = _run(g, sets[name], codes, noise)
        assert np.isfinite(ref).all()
        assert np.array_equal(ref, outs[name]), name
    g.handle.close()


def test_overlap_mode_equals_the_serial_schedule(hip_lib):
    """Option sean.overlap (CUs of the side streams; default 0 = off; handles beyond the run-ahead sizes --
    forced here with sean.ahead = 0): label tables of all
    ACEs ahead on a CU-masked side stream, interior passes on a second one beside the boundary convs (disjoint pixels, no filling),
    everything on the library's own main stream, forked from and joined to the caller's.  Scheduling only: the images must be
    bit-identical to the serial schedule (sean.overlap = 0), call after call, also with the caller on a non-default stream."""
    from ctrlhair_amd import procedural as P
    ngf, B, S = 16, 3, 256
    sd = P.sean_state_dict(0, ngf)
    # (sean.edge = 0 on the serial handle: the overlap schedule's quad-only interior pass does not take the straight-edge pixels, and the
    #  table rows are another association of the same sums -- 1e-6, not bit-identical)
    ov, serial = _gen(sd, B, S, 1, {'sean.ahead': 0, 'sean.overlap': 64}), _gen(sd, B, S, 1, {'sean.ahead': 0, 'sean.overlap': 0, 'sean.edge': 0})
    codes, noise = P.style_codes(B, seed=61), P.noise_planes(B, S, ngf, seed=62)
    sets = _label_sets(B, S)
    for name in ('face', 'blocky', 'one_region', 'noclass_at_tile_borders', 'diag'):
        ref = _run(serial, sets[name], codes, noise)
        for rep in range(2):
            got = _run(ov, sets[name], codes, noise)
            assert np.isfinite(got).all()
            assert np.array_equal(ref, got), (name, rep)
    with torch.cuda.stream(torch.cuda.Stream()):
        got = _run(ov, sets['face'][:2], codes[:2], noise[:2])
    assert np.array_equal(got, _run(serial, sets['face'][:2], codes[:2], noise[:2]))
    ov.handle.close()
    serial.handle.close()


def test_runtime_size_reaches_a_level_max_size_does_not_have(hip_lib):
    """max_size = 96, S = 64, B = 2: the 32-pixel level of S = 64 sits on the Winograd grid, the same level of the
    handle (48 pixels) does not, so no quad lists exist for it -- the ACE must take the direct kernels AND write the hidden
    activations in their layout (one predicate for both).  Samples b >= 1 are the ones a layout mismatch corrupts."""
    from ctrlhair_amd import procedural as P
    ngf, B, S = 16, 2, 64
    sd = P.sean_state_dict(0, ngf)
    direct, wino = _gen(sd, B, 96, 0), _gen(sd, B, 96, 1)
    codes, noise = P.style_codes(B, seed=41), P.noise_planes(B, S, ngf, seed=42)
    lab = np.stack([P.face_like_labels(S, 70 + b) for b in range(B)])
    a, b = _run(direct, lab, codes, noise), _run(wino, lab, codes, noise)
    d = [float(np.abs(a[i] - b[i]).max()) for i in range(B)]
    print(f'max_size=96 S=64: max |winograd - direct| per sample = {d}')
    assert np.isfinite(b).all() and max(d) <= 2e-5
    direct.handle.close()
    wino.handle.close()


def test_option_must_precede_finalize(hip_lib):
    from ctrlhair_amd import procedural as P
    g = _gen(P.sean_state_dict(0, 16), 1, 64, 1)
    with pytest.raises(RuntimeError):
        g.handle.set_option('sean.wino', 0)
    g.handle.close()
