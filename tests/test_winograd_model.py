"""numpy model of the Winograd F(2x2,3x3) evaluation the exact-f32 HIP path uses for its 3x3 convs (csrc/conv_wino.h):

    Y = A^T [ (G g G^T) .* (B^T d B) ] A

* the identity is exact in real arithmetic (checked in float64);
* carried out in float32 -- transformed weights rounded once from double, f32 input / output transforms, f32 products and
  sums -- it stays at the rounding level of the direct f32 evaluation (this is what makes it an f32 path, unlike a
  reduced-precision operand format);
* the style term of an ACE, sum_t P[label(p + t), t]  (normalization.py:117-153,172-173 on the piecewise-constant style map),
  equals the same Winograd conv of the ONE-HOT label planes with the per-sample weights P: what the HIP kernel runs as five
  extra k-steps on the matrix cores.
CPU only (no GPU, no library)."""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def direct_conv(x, w, dtype):
    """x [C,H,W], w [K,C,3,3], zero padding 1; accumulation in `dtype` in (c, tap) order."""
    C, H, W = x.shape
    xp = np.zeros((C, H + 2, W + 2), dtype)
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], H, W), dtype)
    for c in range(C):
        for t in range(9):
            out += w[:, c, t // 3, t % 3].astype(dtype)[:, None, None] * xp[c, t // 3:t // 3 + H, t % 3:t % 3 + W][None]
    return out


def winograd_conv(x, w, dtype):
    """The kernel's order of operations: U = G g G^T in double then rounded to `dtype`; V = B^T d B in `dtype` (adds only);
    M[xi] accumulated over channels in `dtype`; Y = A^T M A in `dtype`."""
    C, H, W = x.shape
    K = w.shape[0]
    U = np.einsum('ia,kcab,jb->ijkc', G, w.astype(np.float64), G).astype(dtype)
    xp = np.zeros((C, H + 2, W + 2), dtype)
    xp[:, 1:-1, 1:-1] = x
    d = np.empty((4, 4, C, H // 2, W // 2), dtype)
    for i in range(4):
        for j in range(4):
            d[i, j] = xp[:, i:i + H:2, j:j + W:2][:, :H // 2, :W // 2]
    t = np.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]])               # rows:    B^T d
    V = np.stack([t[:, 0] - t[:, 2], t[:, 1] + t[:, 2], t[:, 2] - t[:, 1], t[:, 1] - t[:, 3]], axis=1)   # columns: (.) B
    M = np.zeros((4, 4, K, H // 2, W // 2), dtype)
    for c in range(C):                                                                # the MFMA's k-ordered f32 fma chain
        M += U[:, :, :, c][:, :, :, None, None] * V[:, :, c][:, :, None]
    s = np.stack([M[:, 0] + M[:, 1] + M[:, 2], M[:, 1] - M[:, 2] - M[:, 3]], axis=1)  # [4, 2, ...]   (.) A
    Y = np.stack([s[0] + s[1] + s[2], s[1] - s[2] - s[3]])                             # [2, 2, K, h, w]  A^T (.)
    out = np.empty((K, H, W), dtype)
    for a in range(2):
        for b in range(2):
            out[:, a::2, b::2] = Y[a, b]
    return out


def test_transform_matrices():
    """A^T [(G g G^T) .* (B^T d B)] A == the 2x2 valid correlation of a 4x4 patch with a 3x3 kernel, for every basis pair."""
    for t in range(9):
        g = np.zeros((3, 3))
        g[t // 3, t % 3] = 1
        for e in range(16):
            d = np.zeros((4, 4))
            d[e // 4, e % 4] = 1
            y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
            ref = np.array([[(d[a:a + 3, b:b + 3] * g).sum() for b in range(2)] for a in range(2)])
            assert np.array_equal(y, ref)


def test_identity_in_double_and_error_in_float():
    rng = np.random.default_rng(0)
    C, K, H, W = 96, 24, 16, 32
    x = np.maximum(rng.standard_normal((C, H, W)), 0.2 * rng.standard_normal((C, H, W)))      # lrelu-like activations
    w = rng.standard_normal((K, C, 3, 3)) / np.sqrt(9 * C)
    ref = direct_conv(x, w, np.float64)
    assert np.abs(winograd_conv(x, w, np.float64) - ref).max() <= 1e-13
    d32 = np.abs(direct_conv(x.astype(np.float32), w.astype(np.float32), np.float32) - ref).max()
    w32 = np.abs(winograd_conv(x.astype(np.float32), w.astype(np.float32), np.float32) - ref).max()
    print(f'f32 error vs double: direct {d32:.2e}, Winograd F(2x2,3x3) {w32:.2e} (|ref| max {np.abs(ref).max():.2f})')
    assert w32 <= 4e-6 and w32 <= 6 * d32        # the same rounding class (measured: 1.3x ... 3x the direct evaluation)


def test_style_term_as_one_hot_conv():
    """sum_t P[label(p + t), t, c] (taps outside the image and labels >= 19 contribute nothing) == Winograd conv of the one-hot
    planes with weights P -- exact in double, f32-rounding level in float32."""
    rng = np.random.default_rng(1)
    H, W, Cc = 16, 32, 8
    lab = rng.integers(0, 21, size=(H, W))
    lab[lab == 20] = 255                                    # 'no class'
    lab[:, :11] = 3                                         # some uniform areas and boundaries
    lab[5:, 20:] = 7
    P = rng.standard_normal((19, 9, Cc))
    ref = np.zeros((Cc, H, W))
    for y in range(H):
        for x in range(W):
            for t in range(9):
                yy, xx = y + t // 3 - 1, x + t % 3 - 1
                if 0 <= yy < H and 0 <= xx < W and lab[yy, xx] < 19:
                    ref[:, y, x] += P[lab[yy, xx], t]
    onehot = np.stack([(lab == j) for j in range(19)]).astype(np.float64)
    w = P.transpose(2, 0, 1).reshape(Cc, 19, 3, 3)          # [c][j][tap]
    assert np.abs(winograd_conv(onehot, w, np.float64) - ref).max() <= 1e-12
    assert np.abs(winograd_conv(onehot.astype(np.float32), w.astype(np.float32), np.float32) - ref).max() <= 5e-6


def test_centre_tap_kernel_is_a_1x1_conv():
    """A 1x1 conv is the 3x3 conv with a centre tap only; its Winograd weights are non-zero at the four central positions."""
    g = np.zeros((3, 3))
    g[1, 1] = 2.0
    U = G @ g @ G.T
    nz = np.argwhere(U != 0)
    assert sorted(map(tuple, nz)) == [(1, 1), (1, 2), (2, 1), (2, 2)] and np.allclose(np.abs(U[1:3, 1:3]), 0.5)


# ---- F(4x4, 3x3): the ResBlock convs and the low-resolution SPADE convs of the exact-f32 path (csrc/conv_wino4.h) -------------------
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
              dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def winograd4_conv(x, w, dtype):
    """conv_wino4.h's order of operations: U = G g G^T in double, rounded once to `dtype`; V = B^T d B, M accumulated over the channels and
    Y = A^T M A all in `dtype` (the kernel's transforms are fma chains with these small-integer coefficients: one rounding per operation;
    here one rounding per matrix product entry -- the same error level, not the same bits)."""
    C, H, W = x.shape
    K = w.shape[0]
    U = np.einsum('ia,kcab,jb->ijkc', G4, w.astype(np.float64), G4).astype(dtype)
    xp = np.zeros((C, H + 2, W + 2), dtype)
    xp[:, 1:-1, 1:-1] = x
    out = np.empty((K, H, W), dtype)
    bt, at = BT4.astype(dtype), AT4.astype(dtype)
    for ty in range(H // 4):
        for tx in range(W // 4):
            d = xp[:, 4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6]                            # [C, 6, 6]
            V = np.einsum('ia,cab,jb->ijc', bt, d, bt).astype(dtype)
            M = np.zeros((6, 6, K), dtype)
            for c in range(C):
                M += U[:, :, :, c] * V[:, :, c][:, :, None]
            out[:, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4] = np.einsum('ia,abk,jb->kij', at, M, at).astype(dtype)
    return out


def test_f4_transform_matrices():
    """A^T [(G g G^T) .* (B^T d B)] A == the 4x4 valid correlation of a 6x6 patch with a 3x3 kernel, for every basis pair (float64)."""
    for t in range(9):
        g = np.zeros((3, 3))
        g[t // 3, t % 3] = 1.0
        for e in range(36):
            d = np.zeros((6, 6))
            d[e // 6, e % 6] = 1.0
            y = AT4 @ ((G4 @ g @ G4.T) * (BT4 @ d @ BT4.T)) @ AT4.T
            ref = np.array([[(d[a:a + 3, b:b + 3] * g).sum() for b in range(4)] for a in range(4)])
            assert np.abs(y - ref).max() < 1e-12, (t, e)


def test_f4_conv_is_exact_in_double_and_f32_class_in_float():
    """Whole conv, zero padding: identical to the direct sum in float64; in float32 the error is a small multiple of the direct f32
    evaluation's own (the transformed operands span a wider range than F(2x2)'s: DESIGN.md section 2, item 10), far inside the 1e-3
    the contract allows and inside the 5e-4 budget the HIP tests give a whole generator."""
    rng = np.random.default_rng(7)
    C, K, H = 24, 8, 16
    x = rng.standard_normal((C, H, H))
    w = rng.standard_normal((K, C, 3, 3)) / np.sqrt(9 * C)
    ref = direct_conv(x, w, np.float64)
    assert np.abs(winograd4_conv(x, w, np.float64) - ref).max() < 1e-12
    e_dir = np.abs(direct_conv(x.astype(np.float32), w.astype(np.float32), np.float32) - ref).max()
    e_w4 = np.abs(winograd4_conv(x.astype(np.float32), w.astype(np.float32), np.float32) - ref).max()
    e_w2 = np.abs(winograd_conv(x.astype(np.float32), w.astype(np.float32), np.float32) - ref).max()
    print(f'max abs error vs float64: direct f32 {e_dir:.2e}, F(2x2,3x3) f32 {e_w2:.2e}, F(4x4,3x3) f32 {e_w4:.2e} (max |ref| {np.abs(ref).max():.2f})')
    assert e_w4 < 2e-5 and e_w4 < 40 * max(e_dir, 1e-7)


def test_f4_position_layout_of_the_weight_image():
    """pack_wino4_A's fragment order: a = 36 m + 6 i + j for the 16-row half m (conv_wino4.h) -- every (half, position) once, nine
    16-byte units per half."""
    seen = set()
    for m in range(2):
        for i in range(6):
            for j in range(6):
                a = m * 36 + 6 * i + j
                seen.add((a >> 2, a & 3))
    assert len(seen) == 72 and max(u for u, _ in seen) == 17
