"""CPU oracle of the post-generator blending step (TEST INFRASTRUCTURE ONLY: imported by tests/ and smoke(); never by the
product path).

  poisson_blending()  restates /root/reference/poisson_blending.py:29-87: the same linear system (5-point Laplacian with
                      the reference's border rows; identity rows for interior pixels outside the mask -- note the
                      reference leaves BORDER pixels outside the mask as Laplacian rows with the target value on the right
                      hand side, poisson_blending.py:48-56 loops over the interior only), assembled without the Python
                      double loop, solved with the same scipy spsolve, same gamma 2.2 round trip and uint8 truncation.
                      Pinned: tests/golden/poisson_golden.npz holds outputs of the imported reference function
                      (tests/golden/make_poisson_golden.py); test_oracle_golden checks equality.
  blend_mask()        restates hair_editor.py:297-305: hair-mask union, cv2.dilate with 13x13 / 5x5 MORPH_ELLIPSE kernels,
                      background-dependent choice.  cv2 is not installed in this image, so the structuring elements follow
                      OpenCV's published construction (getStructuringElement, MORPH_ELLIPSE: row i covers
                      |j - c| <= round(c * sqrt((r^2 - (i-r)^2) / r^2))); pinned only against the well-known 5x5 ellipse.
                      The 13x13 case is "parity unpinned" against cv2 itself.
"""
import numpy as np
import scipy.sparse
from scipy.sparse.linalg import spsolve

HAIR_IDX, BACKGROUND_IDX = 13, 0


def system_matrix(mask01, H, W):
    """Rows: Laplacian (4 on the diagonal, -1 for every in-image 4-neighbour) except interior pixels with mask == 0,
    which are identity rows (poisson_blending.py:14-26, 48-57)."""
    idx = np.arange(H * W).reshape(H, W)
    interior = np.zeros((H, W), bool)
    interior[1:-1, 1:-1] = True
    ident = interior & (mask01 == 0)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [np.where(ident, 1.0, 4.0).ravel()]
    for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
        ys, xs = np.mgrid[0:H, 0:W]
        ny, nx = ys + dy, xs + dx
        ok = (ny >= 0) & (ny < H) & (nx >= 0) & (nx < W) & ~ident
        rows.append(idx[ok])
        cols.append(idx[ny[ok], nx[ok]])
        vals.append(np.full(ok.sum(), -1.0))
    A = scipy.sparse.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(H * W, H * W))
    return A


def laplacian_apply(img):
    """laplacian.dot(flat) of poisson_blending.py:44,70 for one channel [H,W]."""
    out = 4.0 * img
    out[:, 1:] -= img[:, :-1]
    out[:, :-1] -= img[:, 1:]
    out[1:, :] -= img[:-1, :]
    out[:-1, :] -= img[1:, :]
    return out


def poisson_blending(source, target, mask, with_gamma=True, return_float=False):
    """source, target: [H,W,3] uint8-valued; mask [H,W] or [H,W,1], non-zero = solve (keep source gradients), zero = keep
    target.  Returns uint8 [H,W,3] (and the float solution in gamma space when return_float)."""
    g = 2.2 if with_gamma else 1.0
    src = np.power(np.asarray(source).astype('float'), 1 / g)
    tgt = np.power(np.asarray(target).astype('float'), 1 / g)
    H, W = src.shape[:2]
    m = (np.asarray(mask).reshape(H, W) != 0).astype(np.uint8)
    A = system_matrix(m, H, W)
    sol = np.empty_like(tgt)
    for c in range(src.shape[2]):
        b = laplacian_apply(src[:, :, c].copy()).ravel()
        b[m.ravel() == 0] = tgt[:, :, c].ravel()[m.ravel() == 0]
        sol[:, :, c] = spsolve(A, b).reshape(H, W)
    with np.errstate(invalid='ignore'):
        res = np.power(sol, g)              # reference: plain power (a negative solution would give NaN -> 0 after the cast)
    res[res > 255] = 255
    res[res < 0] = 0
    out = np.nan_to_num(res, nan=0.0).astype('uint8')
    return (out, sol) if return_float else out


def ellipse_kernel(k):
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k)) for odd k."""
    r = c = k // 2
    K = np.zeros((k, k), np.uint8)
    for i in range(k):
        dy = i - r
        dx = int(np.rint(c * np.sqrt((r * r - dy * dy) / float(r * r)))) if r else 0
        K[i, max(c - dx, 0):min(c + dx + 1, k)] = 1
    return K


def dilate(img, K):
    """cv2.dilate(img, K, iterations=1): max over the kernel footprint, anchor at the centre, pixels outside the image
    ignored (default border value of dilation)."""
    H, W = img.shape
    r = K.shape[0] // 2
    out = np.zeros_like(img)
    pad = np.zeros((H + 2 * r, W + 2 * r), img.dtype)
    pad[r:r + H, r:r + W] = img
    for i in range(K.shape[0]):
        for j in range(K.shape[1]):
            if K[i, j]:
                out = np.maximum(out, pad[i:i + H, j:j + W])
    return out


def blend_mask(target_parsing, face_parsing):
    """hair_editor.py:297-305 -> res_mask_dilated [H,W] uint8 (1 = take the generated image)."""
    t, f = np.asarray(target_parsing), np.asarray(face_parsing)
    res_mask = np.logical_or(t == HAIR_IDX, f == HAIR_IDX).astype('uint8')
    d13, d5 = dilate(res_mask, ellipse_kernel(13)), dilate(res_mask, ellipse_kernel(5))
    bg = (t == BACKGROUND_IDX)
    return (d13 * (1 - bg) + d5 * bg).astype('uint8')
