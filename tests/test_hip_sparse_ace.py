"""GPU tests of the exact SPADE-interior reduction (ctrlhair_amd/csrc/ace_sparse.h): gamma/beta of SPADE.forward
(/root/reference/sean_codes/models/networks/normalization.py:249-257) depend on the 5x5 label neighbourhood only, so pixels
with a uniform neighbourhood take per-label constants and only the compacted boundary pixels go through the conv.

The golden / oracle tests of test_hip_sean_generator.py already run with the reduction on (it is the default); here the
sparse path is compared with the dense evaluation of the SAME library (option sean.sparse = 0) on label maps chosen to hit
its edge cases, and the executed-FLOP accounting is checked."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
MODE = {'f32': 0, 'f16x3': 1}


def _gen(sd, mb, ms, path, sparse, extra=None):
    from ctrlhair_amd.sean.generator import SeanGenerator
    opts = {'sean.sparse': sparse}
    opts.update(extra or {})
    g = SeanGenerator(0, f16x3=MODE[path], options=opts).load_state_dict(sd, max_batch=mb, max_size=ms)
    if path == 'f16x3':
        # the f16x3 path serves the reduction in tile-skip mode of its wave-specialised kernel, which small jobs do not select
        # by themselves: force it (sean.dbg bit 64) so that these small cases exercise the mode
        g.handle.set_option('sean.dbg', 64)
    return g


def _run(gen, labels, codes, noise):
    dev = gen.device
    out = gen.generate(torch.from_numpy(labels).to(dev), torch.from_numpy(codes).to(dev), torch.from_numpy(noise).to(dev))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _label_sets(B, S):
    from ctrlhair_amd import procedural as P
    sets = {}
    sets['blocky'] = P.blocky_labels(B, S, grid=8)
    sets['face'] = np.stack([P.face_like_labels(S, 40 + b) for b in range(B)])
    sets['one_region'] = np.full((B, S, S), 13, np.uint8)                      # everything interior except the image frame
    diag = (np.add.outer(np.arange(S), np.arange(S)) % 19).astype(np.uint8)    # no interior pixel at any resolution (a
    sets['diag'] = np.repeat(diag[None], B, 0)                                 # checkerboard would turn uniform when down-sampled)
    noclass = P.blocky_labels(B, S, grid=4, seed=77).copy()                    # labels >= 19 ("no class") never count as interior
    noclass[:, : S // 2, : S // 2] = 255
    noclass[:, S // 2:, S // 2:] = 19
    sets['noclass'] = noclass
    stripes = np.zeros((B, S, S), np.uint8)                                    # 5-pixel stripes: interior = exactly the centre line
    stripes[:] = ((np.arange(S) // 5) % 19)[None, None, :]
    sets['stripes5'] = stripes
    return sets


@pytest.mark.parametrize('path', ['f32', 'f16x3'])
@pytest.mark.parametrize('S', [64, 160])
def test_sparse_equals_dense_tiny(hip_lib, path, S):
    """ngf=16 (a single 64-row tile per ACE: partially filled blocks), S=160 -> ACE resolutions 160 / 80 (ragged tiles)."""
    from ctrlhair_amd import procedural as P
    ngf, B = 16, 3
    sd = P.sean_state_dict(0, ngf)
    dense = _gen(sd, B, S, path, 0)
    sparse = _gen(sd, B, S, path, 1, {'sean.sparse_min': 32})
    codes, noise = P.style_codes(B), P.noise_planes(B, S, ngf)
    for name, lab in _label_sets(B, S).items():
        a, b = _run(dense, lab, codes, noise), _run(sparse, lab, codes, noise)
        d = float(np.abs(a - b).max())
        print(f'{path} S={S} {name}: max |sparse - dense| = {d:.3e}')
        assert np.isfinite(b).all() and d <= 2e-5, name
    dense.handle.close()
    sparse.handle.close()


@pytest.mark.parametrize('path', ['f32', 'f16x3'])
def test_sparse_equals_dense_ngf64(hip_lib, path):
    """ngf=64 at 256x256 (row tiles 2 ... 32), against the dense evaluation and against the oracle on face-like labels."""
    from ctrlhair_amd import procedural as P
    from oracle import sean_oracle as O
    ngf, S, B = 64, 256, 2
    sd = P.sean_state_dict(0, ngf)
    dense, sparse = _gen(sd, B, S, path, 0), _gen(sd, B, S, path, 1)
    codes, noise = P.style_codes(B), P.noise_planes(B, S, ngf)
    sets = _label_sets(B, S)
    for name in ('blocky', 'face', 'noclass'):
        a, b = _run(dense, sets[name], codes, noise), _run(sparse, sets[name], codes, noise)
        d = float(np.abs(a - b).max())
        print(f'{path} ngf64 {name}: max |sparse - dense| = {d:.3e}')
        assert d <= 2e-5, name
    ref = O.generator_forward(O.to_torch(sd), sets['face'][:1], codes[:1], noise[:1], ngf).numpy()
    got = _run(sparse, sets['face'][:1], codes[:1], noise[:1])
    assert np.abs(got - ref).max() <= 1e-3
    dense.handle.close()
    sparse.handle.close()


@pytest.mark.parametrize('path', ['f32', 'f16x3'])
def test_executed_flops_accounting(hip_lib, path):
    """ch_profile_read_ex: executed <= dense FLOPs; equal for a label map without interior pixels; the executed fraction of a
    single-region map is the 2-pixel image frame (rounded up to 32-pixel sub-tiles)."""
    from ctrlhair_amd import procedural as P
    ngf, S, B = 16, 128, 2
    sd = P.sean_state_dict(0, ngf)
    gen = _gen(sd, B, S, path, 1, {'sean.sparse_min': 64})
    codes, noise = P.style_codes(B), P.noise_planes(B, S, ngf)
    sets = _label_sets(B, S)
    frac = {}
    for name in ('diag', 'one_region', 'face'):
        gen.handle.profile_enable(True)
        _run(gen, sets[name], codes, noise)
        gen.handle.profile_enable(False)
        ace = gen.handle.profile_read(1)
        gen.handle.profile_read(-1)
        assert ace['launches'] > 0 and 0 < ace['flops_executed'] <= ace['flops'] * 1.2     # (C = 16 layers run a padded 64-row tile)
        frac[name] = ace['flops_executed'] / ace['flops']
    print('executed / dense SPADE-conv FLOPs:', frac)
    if path == 'f32':
        # Winograd F(2x2,3x3) on the exact-f32 path: 16 products per quad and channel instead of 36, plus (styled ACEs) 20
        # one-hot planes behind the 128 hidden channels: 16 / 36 * 148 / 128 = 0.514 when every quad is a boundary quad
        # (levels below 32 pixels and 16-channel layers run padded tiles)
        # -- and, with sean.wino = 2, the levels of 32 / 64 pixels as F(4x4,3x3) over every tile: 36 / 144 * 152 / 128 = 0.30
        assert 0.30 - 1e-9 <= frac['diag'] <= 0.62
        assert frac['one_region'] < 0.45 * frac['diag'] / 0.5 and frac['face'] < 0.9 * frac['diag']
    else:
        assert 1.0 - 1e-9 <= frac['diag'] <= 1.2
        # tile granularity (tiles of 32 x 16 without any boundary pixel are skipped)
        assert frac['one_region'] < 0.95 and frac['face'] <= frac['diag'] + 1e-9      # (a 128-pixel face has no boundary-free tile)
    gen.handle.close()


@pytest.mark.parametrize('path', ['f32', 'f16x3'])
def test_interior_and_label_table_kernels_match_their_first_versions(hip_lib, path):
    """The interior pass on blocks of 32 x 8 pixels (mostly-interior blocks write every pixel, the boundary conv overwrites the
    rest) and the compacting label-table kernel of the 512^2 level against the row-shaped kernels they replaced (sean.dbg bits
    2097152 / 33554432): the same arithmetic in the same order, so the images must be bit-identical."""
    from ctrlhair_amd import procedural as P
    ngf, S, B = 16, 512, 2
    sd = P.sean_state_dict(0, ngf)
    codes, noise = P.style_codes(B), P.noise_planes(B, S, ngf)
    new = _gen(sd, B, S, path, 1)
    old = _gen(sd, B, S, path, 1)
    try:
        old.handle.set_option('sean.dbg', (64 if path == 'f16x3' else 0) | 2097152 | 33554432)
    except RuntimeError as e:        # the superseded kernels only exist in A/B builds (make -C ctrlhair_amd/csrc ABLATE=1)
        new.handle.close()
        old.handle.close()
        assert 'CH_ABLATE' in str(e)
        pytest.skip('library built without -DCH_ABLATE: the first-version kernels are not in it')
    sets = _label_sets(B, S)
    for name in ('face', 'blocky', 'noclass', 'stripes5'):
        a, b = _run(new, sets[name], codes, noise), _run(old, sets[name], codes, noise)
        assert np.isfinite(a).all()
        assert np.array_equal(a, b), (name, float(np.abs(a - b).max()))
    new.handle.close()
    old.handle.close()


@pytest.mark.parametrize('mode,ngf,S,B,tol', [(1, 64, 256, 3, 2e-5), (1, 16, 512, 2, 2e-5), (1, 16, 416, 3, 2e-5), (2, 64, 256, 2, 2e-2), (3, 16, 512, 2, 1e-1)])
def test_straight_edge_pixels_on_the_f16_paths(hip_lib, mode, ngf, S, B, tol):
    """Option sean.edge on the f16x3 (mode 1) / single-term f16 (2) / bf16 (3) paths: with pixel-level compaction the 32 x 8 interior pass
    (ace_interior_sh16_tile_kernel) serves the straight-edge pixels from a per-block table of their codes' rows -- E[code] of the ACE plus
    the three style-LUT column / row sums -- instead of the boundary conv.  Against the conv evaluation of the same library
    (sean.edge = 0): f16x3 at 2e-5 (two f32-class evaluations of the same sums), the reduced-precision paths at their own tolerance
    (the table rows are exact f32, the conv they replace is not; bf16: two evaluations, each inside 5e-2 of the exact image); S = 416: levels of
    208 and 416 pixels, i.e. partial 32 x 8 blocks and partial classification tiles; fewer boundary tiles must run where straight edges exist.
    normalization.py:117-153,172-187,249-257."""
    from ctrlhair_amd import procedural as P
    from ctrlhair_amd.sean.generator import SeanGenerator
    sd = P.sean_state_dict(0, ngf)

    def gen(edge):
        g = SeanGenerator(0, f16x3=mode, options={'sean.edge': edge}).load_state_dict(sd, max_batch=B, max_size=S)
        g.handle.set_option('sean.dbg', 64)
        return g
    on, off = gen(1), gen(0)
    codes, noise = P.style_codes(B, seed=91), P.noise_planes(B, S, ngf, seed=92)
    sets = _label_sets(B, S)
    for name in ('blocky', 'face', 'noclass', 'stripes5', 'diag', 'one_region'):
        a, b = _run(on, sets[name], codes, noise), _run(off, sets[name], codes, noise)
        d = float(np.abs(a - b).max())
        print(f'mode {mode} ngf{ngf} S={S} {name}: max |edge rows - boundary conv| = {d:.3e}')
        assert np.isfinite(a).all() and d <= tol, (name, d)
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
    on.handle.close()
    off.handle.close()


def test_pair_and_quad_entries_match_single_row_tile_entries(hip_lib):
    """f16x3 compacting kernel: tiles with at most four sub-tiles of boundary pixels are served two or four row tiles per
    iteration, with the A fragments streamed by the consumer waves (conv_sh16_ws_kernel<..., CP = 2 / 3>); option
    sean.sh16_compact = 2 serves every tile one row tile at a time through the loaders' A DMA.  Same products added in the same
    order: bit-identical images.  ngf = 64 at 256^2 gives 2 ... 32 row tiles per layer (quad entries need at least four)."""
    from ctrlhair_amd import procedural as P
    ngf, S, B = 64, 256, 2
    sd = P.sean_state_dict(0, ngf)
    codes, noise = P.style_codes(B), P.noise_planes(B, S, ngf)
    a_gen = _gen(sd, B, S, 'f16x3', 1)
    b_gen = _gen(sd, B, S, 'f16x3', 1, {'sean.sh16_compact': 2})
    sets = _label_sets(B, S)
    for name in ('face', 'blocky', 'one_region', 'stripes5'):
        a, b = _run(a_gen, sets[name], codes, noise), _run(b_gen, sets[name], codes, noise)
        assert np.isfinite(a).all()
        assert np.array_equal(a, b), (name, float(np.abs(a - b).max()))
    a_gen.handle.close()
    b_gen.handle.close()
