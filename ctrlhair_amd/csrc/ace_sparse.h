// ace_sparse.h -- the exact SPADE-interior reduction of the ACE layers (hard part (iii)).
//
// SPADE.forward (/root/reference/sean_codes/models/networks/normalization.py:249-257) is
//      actv = relu(conv3x3_{19->128}(onehot));  gamma = conv3x3_{128->C}(actv);  beta = conv3x3_{128->C}(actv)
// so gamma/beta at pixel p depend on the labels of the 5x5 neighbourhood of p only.  Where that neighbourhood is uniform
// (one label j, entirely inside the image -- both convs zero-pad) every tap sees the same hidden vector
//      a_j = relu(b_shared + sum_t' W_shared[:, j, t'])
// and gamma/beta are per-label constants  G_j = b + (sum_t W[:, :, t]) a_j  (tables built once at ch_finalize, in double).
// The style term of a styled ACE (normalization.py:117-153,172-173: conv_gamma/conv_beta on the piecewise-constant style map)
// is likewise sum_t P[(sample, j), t] there.  So:
//   * INTERIOR pixels (5x5-uniform): no convolution at all -- an elementwise pass (ace_interior) modulates x with the
//     per-(sample, label) table row;
//   * BOUNDARY pixels (everything else, including the 2-pixel frame of the image and labels >= 19) are compacted per spatial
//     tile into 32-pixel MFMA sub-tiles and go through the dense SPADE conv + fused ACE epilogue (conv_ace_sparse.h /
//     conv_sh16.h), whose B-fragment addresses are per lane anyway.
// Identities in real arithmetic; in fp32 the interior value is the same sum in another association (1e-6 level).
//
// Round 6 -- STRAIGHT-EDGE pixels (exact-f32 Winograd path, levels of 128 pixels and more; option "sean.edge").  Most boundary pixels of a
// label map sit next to ONE straight, axis-aligned piece of a region border: their 5x5 neighbourhood is five uniform columns (or rows)
// A..A B..B with s = 1..4 columns of label A.  The hidden activation then depends on the column only (three vectors out of
// {a_A, h_AAB, h_ABB, a_B}, h_XYZ = relu(b + sum_tx Tcol[X|Y|Z][tx]) with the shared conv's taps summed over rows), and
//      gamma = b + sum_dx (sum_dy W[:, :, (dy, dx)]) h_dx(A, B, s)
// is a per-(orientation, A, B, s) constant: 2 x 19 x 19 x 4 = 2888 table rows E[code][gamma|beta][C] per ACE, built once at
// ch_finalize in double (ace_edge_table).  The style term (normalization.py:117-153,172-173) of such a pixel is three column (row)
// sums of the style LUT, P6[(sample, label)][3 columns + 3 rows][gamma|beta][C], built per call (ace_p6table).  Such pixels are marked
// u5 = 253 with their code in e16 and are modulated by the interior pass -- no convolution; only corners, curved pieces and the image
// frame are left to the boundary conv.  Same real number, another association of the f32 sum (as the interior pixels).
//
// Per resolution level and generate() chunk, ace_classify builds
//      u5   [B][H][W]  uint8   label if the pixel is interior, else 255
//      need [B][H][W]  uint8   1 where the boundary conv reads the SPADE hidden activations (the label-table kernel skips the rest)
//      cnt  [ntiles]   int     boundary pixels of the tile (tiles of 32 x TH pixels, one sample each)
//      list [ntiles][32*TH] uint16  their in-tile offsets ty*32+tx in raster order
// and ace_worklist turns the counts into the list of block tasks of one conv launch (depends on the row tiles of the layer):
// a tile's NS = ceil(cnt/32) sub-tiles are split into `ng` groups of at most 4 (one group = the N extent of one wave), each
// group meets every 64-row wave tile: ng * mtiles wave tasks, 4 per block (group-major, so that the 4 waves of a block share
// the group when mtiles % 4 == 0 and only the A rows differ).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace chk {

// sub-tile groups of a tile with NS sub-tiles for a layer with `mtiles` 64-row wave tiles: `ng` groups of `per` (last: rest)
__host__ __device__ inline void sparse_groups(int NS, int mtiles, int& ng, int& per) {
    if (NS <= 0) {
        ng = 0;
        per = 0;
        return;
    }
    int g = (NS + 3) >> 2;
    while (((g * mtiles) & 3) && g < NS) ++g;      // few row tiles: more, smaller groups so that all 4 waves of a block work
    per = (NS + g - 1) / g;
    ng = (NS + per - 1) / per;
}

// block tasks of a tile of 32 x TH pixels in the dense case (every pixel a boundary pixel): sizes work lists and grids
__host__ __device__ inline int sparse_max_tasks(int TH, int mtiles) {
    int ng, per;
    sparse_groups(TH, mtiles, ng, per);
    return (ng * mtiles + 3) / 4;
}

struct SparseLevel {            // device buffers of one resolution level (sean_model.cpp allocates them at build())
    uint8_t* u5 = nullptr;
    uint16_t* e16 = nullptr;    // [B][H][W] code of a straight-edge pixel (u5 == 253): ((orientation * 19 + A) * 19 + B) * 4 + (s - 1); or null
    uint8_t* need = nullptr;    // [B][H][W] 1 where the boundary conv reads the SPADE hidden activations (3x3 around a boundary pixel)
    uint16_t* list = nullptr;
    int* cnt = nullptr;
    int TH = 8;                 // tile height the level was classified with (tiles are 32 wide)
    int cap_tiles = 0;
};
struct SparseWork {             // block tasks of one (level, mtiles) pair
    unsigned* work = nullptr;   // tile | (block task within the tile << 20)
    unsigned* work2 = nullptr;  // mode 3: tile | (pair of row tiles << 20) for the tiles with at most four sub-tiles
    int* total2 = nullptr;      // [0] entries of work2, [1] entries of work3
    unsigned* work3 = nullptr;  // mode 3, at least four row tiles: tile | (group of four row tiles << 20) for the tiles with one or two sub-tiles
    int* total = nullptr;       // [0] number of block tasks, [1] boundary pixels, [2] sub-tiles (x32 = pixels the MFMAs run over)
                                // [3] wave tasks x sub-tiles (x 32 x 64 rows = accumulators computed)
    int mtiles = 0;
    int TH = 8;                 // tile height of the classification this list was built from
    int mode = 0;               // ace_worklist mode: 0 = block tasks of conv_ace_sparse_kernel, 1 = f16x3 tile-skip entries,
                                // 2 = entries of the compacting f16x3 kernel, 3 = the same with pair entries (ace_sparse.hip)
    long long cap = 0;
};

// lab: [B][H][W] labels of the level.  Tiles of 32 x TH pixels (TH = 8 or 16).
// e16 (may be null): also recognise the straight-edge pixels (u5 = 253, e16 = code); they are NOT boundary pixels of the lists
constexpr int ACE_EDGE = 253, ACE_EDGE_CODES = 2 * 19 * 19 * 4;
hipError_t ace_classify(const uint8_t* lab, uint8_t* u5, uint8_t* need, uint16_t* list, int* cnt, int B, int H, int W, int TH,
                        hipStream_t s, uint16_t* e16 = nullptr);
// E[code][gamma|beta][C] (float) from the column / row sums of the SPADE gamma/beta weights W6[gb][k][6][C] (double; 0..2: sum over dy of
// tap (dy, dx = -1, 0, 1), 3..5: sum over dx of tap (dy = -1, 0, 1)) and the hidden vectors hv[orientation][741][128] (double; index 0..18:
// a_j, 19 + (A * 19 + B) * 2 + (0: window X X Y = AAB | 1: window X Y Y = ABB)); scale_g / scale_b: the SPADE share of the blend;
// bias_g / bias_b are added (the table rows are complete gamma / beta of an unstyled ACE)
hipError_t ace_edge_table(const double* W6, const double* hv, const float* bias_g, const float* bias_b, float scale_g, float scale_b,
                          float* E, int C, hipStream_t s);
// P6[(b, j)][k = 0..5][gamma|beta][C] = lut_mul * sum of the three style-LUT taps of column k (k < 3: taps (dy, dx = k - 1)) or row k - 3
hipError_t ace_p6table(const float* lut, int lut_rs, int lut_ns, int lut_bs, float lut_mul, float* p6, int B, int C, hipStream_t s);
// mode 0: block tasks of conv_ace_sparse_kernel; mode 1: (tile, row tile) pairs of the tiles with a boundary pixel (f16x3 tile-skip)
hipError_t ace_worklist(const int* cnt, int ntiles, int mtiles, unsigned* work, int* total, hipStream_t s, int mode = 0,
                        int tile_px = 512, unsigned* work2 = nullptr, int* total2 = nullptr,    // mode 3: second list (pair entries)
                        unsigned* work3 = nullptr);                                              // and third (quad entries; may be null)
// gtab[b][j][gamma|beta][C] = bias + gconst[j] + sum_t lut[(t, gamma|beta, c)][(b, j)]   (lut may be null: unstyled ACE)
// lut element (row = (t*2+gb)*C + c, n = b*lut_bs + j) at lut[row*lut_rs + n*lut_ns]; lut_mul undoes a pre-multiplied LUT
hipError_t ace_gtable(const float* bias_g, const float* bias_b, const float* gconst, const float* lut, int lut_rs, int lut_ns,
                      int lut_bs, float lut_mul, float* gtab, int B, int C, hipStream_t s);

struct AceInteriorParams {
    const float* x;             // exact-f32 path: NCHW [B][C][H>>x_up][W>>x_up]; f16x3 path: C4 [B][C/4][h][w][4]
    void* out;                  // NCHW f32 / SH16
    const uint8_t* u5;
    const float* gtab;          // [B][19][2][C]
    const float *bn_a, *bn_d, *nv;
    const float* noise;         // plane base of this ACE, sample stride noise_bstride, layout [W][H]
    long long noise_bstride;
    int B, C, H, W, x_up, act;
    // f16x3 path (SH16 output): first-pass scale, the producer's slot, pass (0 record max / 1 rewrite if the max left the window)
    float out_scale;
    unsigned* out_amax;
    int pass, bf16;
    int single;                 // f16x3 kernels: 1 = single-term operands (f16 / bf16 legs): the low halves are never read, their plane is not written
    const int* cnt;             // f16x3 path: boundary-pixel count per tile of 32 x 16 (tiles_x = ceil(W / 32))
    int impl;                   // 0 = blocks of 32 x 8 pixels (default), 1 = blocks of 256 consecutive pixels (first version, A/B),
                                // 2 = exact-f32 kernel only: four pixels per thread, blocks of 128 x 8 (W >= 128)
    int fill_min;               // f16x3 kernel, variant 1: a block with at least this many interior pixels (of 256) writes ALL its
                                // pixels -- the boundary conv, launched after this pass, overwrites the others; 0 = 128
    int quad_only;              // exact-f32 tile kernels: 1 = write an interior pixel only when its whole 2 x 2 quad is interior (the Winograd
                                // boundary conv writes all four pixels of a boundary quad; needed when the two run concurrently)
    const uint16_t* e16;        // exact-f32 tile4 kernel: codes of the straight-edge pixels (u5 == 253), or null
    const float* etab;          //   E[2888][2][C] of this ACE (bias included)
    const float* p6;            //   P6[(b, j)][6][2][C] of this call, or null (unstyled ACE)
    int variant;                // exact-f32 kernel: 0 = one pixel per thread (default), 1 = four pixels per thread (16-byte
                                // accesses), 2 = one pixel per thread writing whole 32-byte sectors (A/B measurements)
};
hipError_t ace_interior_f32(const AceInteriorParams& q, hipStream_t s);
// f16x3 path (tile-skip mode): the pixels of the tiles of 32 x 16 WITHOUT a boundary pixel (q.cnt[tile] == 0); x in the C4
// layout, out in the SH16 layout, scale protocol of sh16.h (q.pass 0: write at out_scale and record the maximum in q.out_amax;
// pass 1: return at once unless the recorded maximum left the f16 window, else rewrite at the corrected scale)
hipError_t ace_interior_sh16(const AceInteriorParams& q, hipStream_t s);

}  // namespace chk
