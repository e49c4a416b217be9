// sean_model.h -- host side of the SEAN generator path: weight folding/packing at load, workspace arena,
// and the launch sequence of one batched forward.  No torch types; plain HIP runtime.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "ace_sparse.h"
#include "net_common.h"

namespace chk {

struct ConvW {
    float* wpk = nullptr;
    float* bias = nullptr;
    float* wscale = nullptr;                    // f16x3 path: per-row 2^-k undoing the weight scaling (sh16.h)
    float* wino = nullptr;                      // exact-f32 path, 3x3: Winograd F(2x2,3x3) image U = G g G^T (conv_wino.h)
    float* wino4 = nullptr;                     // exact-f32 path, 3x3, sean.wino = 2: Winograd F(4x4,3x3) image (conv_wino4.h)
    float* pw = nullptr;                        // exact-f32 path, 1x1: pack_pw_A image (conv_pw.h)
    int Cout = 0, Cin = 0, KS = 0;
};

struct AceW {
    std::string name;
    int C = 0, res_div = 1, index = 0;
    bool styled = false;
    float* spade_wpk = nullptr;                 // (gamma|beta) 64-row tiles, K = 128*9
    float *bias_g = nullptr, *bias_b = nullptr; // blended biases
    float *bn_a = nullptr, *bn_d = nullptr, *nv = nullptr;
    float *actv_table = nullptr, *actv_bias = nullptr;   // mlp_shared as label LUT [19*9][128]
    float *fcmu_w = nullptr, *fcmu_b = nullptr;          // [19][512][512], [19][512]
    float* lut_wpk = nullptr;                   // rows (tap, gamma|beta, c) x K=512
    float* lut_rows = nullptr;                  // same rows, plain [18C][512] (GEMV path for batches <= 3)
    float* lut_wt = nullptr;                    // the same matrix transposed, [512][18C]: the "image" of the grouped LUT build (exact f32)
    float *spade_wscale = nullptr, *lut_wscale = nullptr;   // f16x3 path: per-row 2^-k of the packed rows (sh16.h)
    float out_scale = 8.f;                      // f16x3 path: first-pass SH16 scale of this ACE's output (SH16_ACT_SCALE; the
                                                //   shortcut's ace_s carries the 2^D aligning conv_s with conv_1, sean_model.cpp)
    float actv_scale = 1.f;                     // f16x3 path: SH16 scale of the SPADE hidden activations (from a table bound)
    float* spade_wino = nullptr;                // exact-f32 Winograd path: pack_wino_A image of the (gamma | beta) rows, 16-channel row tiles
    float* edge_tab = nullptr;                  // sean.edge: E[2888][gamma|beta][C] of the straight-edge pixels (ace_sparse.h), ACEs with res_div <= 4
    float* spade_wino4 = nullptr;               // sean.wino = 2: the same rows as an F(4x4,3x3) image (conv_wino4.h wino4_ace_row), levels <= wino4_ace_max_r
    float* gconst = nullptr;                    // [19][gamma|beta][C]: SPADE gamma/beta of a pixel whose 5x5 label neighbourhood
                                                //   is uniformly j (blend factor folded in, biases not) -- ace_sparse.h
};

struct BlockW {
    std::string name;
    int fin = 0, fout = 0, fmid = 0, res_div = 1;
    bool up_before = false, styled = false, learned = false;
    ConvW conv_0, conv_1, conv_s;
    AceW ace_0, ace_1, ace_s;
};

struct ProfRec {
    hipEvent_t e0, e1;
    int kind;       // 0 plain conv, 1 ACE conv, 2 LUT gemm, 3 interior pass of a sparse ACE (table build + elementwise kernel)
    double flops, bytes;            // flops: the dense evaluation of the layer (every pixel through the conv)
    double flops_exec = -1.0;       // FLOPs the matrix cores ran when they differ from `flops` (Winograd convs: 16 / 36); < 0: = flops
    const int* sp_stat = nullptr;   // sparse ACE launch: snapshot of the statistics of its work list (read back at ch_profile_read)
    double sp_flops_unit = 0.0;     //   executed FLOPs = sp_stat[3] * sp_flops_unit
    double sp_bytes_px = 0.0, sp_bytes_fixed = 0.0, sp_npix = 0.0;   // algorithmic bytes = fixed + per-pixel x (boundary pixels
                                    //   sp_stat[1] for kind 1, interior pixels sp_npix - sp_stat[1] for kind 3)
};

struct SeanModel {
    int ngf = 0, max_batch = 0, max_size = 0;
    float* gb_small = nullptr;                  // exact-f32 path: gamma | beta sums of a tiny ACE level whose SPADE conv runs as a plain split-K conv (ace())
    long long gb_small_cap = 0;
    float* splitk_ws = nullptr;
    long long splitk_cap = 0;
    int dbg = 0;               // perf experiments (conv_mfma.h ConvParams::dbg)
    int dbg_sel = 16;          // dbg bit 256: index of the ACE launch whose tiles are cycle-stamped
    int terms = 3;             // f16 MFMA path: 3 = split operands (f32-class), 1 = f16 operands (BASELINE configs[4] class)
    bool use_sh16 = false;     // generator convs on the f16x3 split-operand MFMA path (conv_sh16.h)
    std::vector<BlockW> blocks;
    float *fc_table = nullptr, *fc_bias = nullptr;     // fc conv as label LUT [19*9][16ngf]
    float *img_w = nullptr, *img_b = nullptr;          // conv_img raw [3][ngf][3][3]
    float* img_w4 = nullptr;                           // the same, [ngf/4][tap][co][4] (conv_img_c4_kernel's scalar loads)
    bool has_zencoder = false;
    float* z1_w = nullptr;                             // Zencoder stem weights, unpacked (direct VALU conv)
    ConvLayer z1, z4, z7, z10, z14;                    // architecture.py:158-176
    float *z14_sh = nullptr, *z10_sh = nullptr;        // z14 / the ConvTranspose packed for the f16x3 kernel
    float* z10_wino = nullptr;                         // exact-f32 path: the ConvTranspose as four Winograd phase convs of the input grid (rows 4 co + phase)
    float* z10_pw[4] = {};                             // option "sean.convt_gemm": the same ConvTranspose as four phase GEMMs over shifted views (conv_pw.h; phase = 2 py + px)
    int convt_gemm = 1;
    float* z14_wino = nullptr;                         // exact-f32 path: z14 as Winograd A images (conv_wino.h, reflection padding)
    float* z14_wino4 = nullptr;                        // ... and as F(4x4,3x3) images (conv_wino4.h), sean.wino = 2
    float *z14_ws = nullptr, *z10_ws = nullptr;        // their per-row inverse weight scales
    float *z10_d2s = nullptr, *z10_d2s_ws = nullptr;   // the ConvTranspose in its 2x2-tap depth-to-space form (rows = phase * 256 + co)
    ConvLayer z4_s2d, z7_s2d;                          // the two stride-2 convs in the space-to-depth form (conv_sh16.h S2D)
    std::vector<void*> allocs;                         // everything to hipFree
    // workspace
    uint8_t* lab_r[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // res_div 32,16,8,4,2 (index by log2) ; [0] unused
    unsigned* amax_slots = nullptr;            // f16x3 path: recorded maxima of the dynamically scaled SH16 tensors (sh16.h)
    float *noise_ws = nullptr, *mu_img = nullptr, *lut = nullptr, *actv = nullptr;
    // f16x3 path, batches above the GEMV threshold: the style projections of ALL styled ACEs come from one launch
    // (fc_mu_batched) into per-ACE images mu_all + index * mu_stride; device arrays of the per-ACE weight / bias pointers
    bool fcmu_batched = false;
    float* mu_all = nullptr;
    long long mu_stride = 0;
    const float** fcmu_w_ptrs = nullptr;
    const float** fcmu_b_ptrs = nullptr;
    float *h0 = nullptr, *hs = nullptr, *dx = nullptr, *h1 = nullptr, *xs = nullptr, *xa = nullptr, *xb = nullptr;
    // run-ahead mode of interactive-size jobs: label / style-only kernels of every ACE on a side stream (sean_model.cpp)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_join;
    std::vector<float*> actv_ahead, lut_ahead;
    // exact-f32 path, more than 64 (sample, label) columns: the style LUTs of all styled ACEs from ONE grouped GEMM launch at the
    // start of a chunk (conv_pw.h: operands swapped -- the projected codes are the A operand, packed by fc_mu_batched) into lut_ahead
    void* lut_groups = nullptr;                 // device array of PwGroup
    int lut_ngroups = 0, lut_group_tiles = 0;   // total pixel tiles (tasks = tiles x row groups of the call's batch)
    double lut_group_rows = 0.0;                // sum of 18 C over the groups (profiling figures)
    bool ahead_full = false;                   // handle sized for the full run-ahead mode (else: style LUTs only)
    float* splitk_side = nullptr;
    long long ahead_pixels = -1;               // largest B*S*S served in run-ahead mode (-1: default 8 x 512^2)
    int n_aces = 0;
    // exact SPADE-interior reduction (ace_sparse.h): per resolution level (index = log2(res_div)) the classification buffers
    // and one work list per distinct number of 64-row tiles among the level's ACEs; gtab: [max_batch][19][2][C max]
    int wino = 2;                              // option "sean.wino": exact-f32 path, 3x3 convs as Winograd on the f32 matrix cores:
                                               //   1 = F(2x2,3x3) everywhere (conv_wino.h); 2 = the ResBlock convs from 32 x 32 pixels as
                                               //   F(4x4,3x3) (conv_wino4.h), the rest F(2x2,3x3); 0 = direct evaluation (conv_mfma.h)
    float* zero_page = nullptr;                // 256 bytes of zeros
    // Winograd ACE path (conv_wino.h wino_ace_kernel): per resolution level the boundary quads of every 32 x 16 tile and one task
    // list per distinct row-tile count; per-sample style images of the ACE being run
    struct WinoWork { int nrt = 0; unsigned* work = nullptr; int* total = nullptr; };
    struct WinoLevel {
        uint8_t* qlist = nullptr; int* qcnt = nullptr; int* pcnt = nullptr; int cap_tiles = 0, TH = 16; std::vector<WinoWork> works;
        // gather mode (conv_wino.h): one list of boundary quads per sample, tasks of 64 consecutive entries
        unsigned* gq = nullptr; int* gq_n = nullptr; int* qoff = nullptr; int gq_cap = 0;
        int* chunk_base = nullptr;        // [33] first chunk of 64 quads of every sample (patch source of the gather kernel, conv_wino.h)
        int* patch_mode = nullptr;        // device flag: 1 = this chunk's patches come pre-gathered from SeanModel::patchbuf
    };
    float* patchbuf = nullptr;                 // pre-gathered hidden-activation patches of the ACE being run (few, scattered boundary quads)
    int patch_cap_chunks = 0;
    int patch = 1;                             // option "sean.patch": 0 = the gather kernel always fetches from the hidden planes
    // Overlap mode (round 5; option "sean.overlap" = CUs of the side streams, 0 = off; exact-f32 Winograd path, jobs beyond the
    // run-ahead sizes): the HBM-bound kernels -- label tables of all ACEs, interior passes -- run on streams whose CU mask holds
    // `overlap` CUs (hipExtStreamCreateWithCUMask: every XCD gives overlap / 8 of its CUs), beside the matrix-bound convs on the
    // internal main stream, which claim their tasks dynamically (conv_wino.h) and so lose only the CUs, not the time, the side
    // kernels hold.  CU-masked streams are blocking streams (they synchronise with the NULL stream), so the whole generate() runs
    // on `main_i`, forked from and joined to the caller's stream by events.
    int overlap = 0;                           // (measured, DESIGN.md section 7: the side kernels are CU-bound, not HBM-bound, once confined -- no gain; default off)
    bool overlap_on = false;                   // streams / buffers of the mode exist (build())
    hipStream_t main_i = nullptr, side_int = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    std::vector<hipEvent_t> ev_x, ev_int;      // per ACE: its input is ready (main -> side_int), its interior pass is done (side_int -> main)
    unsigned* claim_pool = nullptr;            // dynamic task claiming: CLAIM_SLOTS launches x CLAIM_WORDS words (8 counters + mailboxes), zeroed per chunk
    static constexpr int CLAIM_SLOTS = 96, CLAIM_WORDS = 1024;
    int hidden_wq = 1;                         // option "sean.hidden_wq": Winograd ACE path from 128 pixels, SPADE hidden activations + one-hot planes
                                               //   from one persistent kernel, only where a boundary quad's patch reads them (0: every pixel, two kernels)
    int lut_grouped = 1;                       // option "sean.lut_grouped": exact-f32 path, style LUTs of all styled ACEs from one grouped GEMM launch
    int wino_gather = 1;                       // option "sean.wino_gather": 1 = gather mode of the Winograd ACE kernel (default), 0 = tile mode
    int wino_th = 0;                           // option "sean.wino_th": tile height 16 / 32 of the Winograd ACE kernel (0 = by level)
    // measured per level at B = 16, 512^2 on the benchmark labels (ms for the level's three ACEs, tiles of 32 x 16 / 32 x 32):
    // 512^2 13.8 / 9.6, 256^2 8.1 / 9.6, 128^2 7.9 / 7.1, 64^2 4.1 / 4.8
    int wino_tile_h(int r) const { return (wino_th == 16 || wino_th == 32) ? (r % wino_th ? 16 : wino_th) : ((r >= 512 || r == 128) ? 32 : 16); }
    WinoLevel wq_level[6];
    float* actv_lvl[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // per Winograd ACE level: hidden activations in the padded layout (conv_wino.h WINO_AXOFF)
    std::map<const float*, long long> pad_state;   // geometry (size, planes) a padded buffer's zero columns were last cleared for
    float* wsty = nullptr;
    float* wsty4 = nullptr;                    // per-sample F(4x4,3x3) style images of the ACE being run (conv_wino4.h)
    int num_cus = 256;                         // compute units of the handle's device (build())
    int wino4_force = 0;                       // option "sean.wino4_force": 1 = F(4x4,3x3) wherever the shape allows, whatever the task count (tests)
    int wino4_ace_max_r = 64;                  // option "sean.wino4_ace": largest level whose SPADE convs run as F(4x4,3x3) over EVERY tile (0 = none)
    int edge = 1;                              // option "sean.edge": 1 = straight-edge pixels of the levels >= 128 pixels are modulated by the interior pass from
                                               //   per-code table rows instead of going through the boundary conv (exact-f32 Winograd path, f16 paths in compaction mode; ace_sparse.h)
    float* p6 = nullptr;                       // per-call column / row sums of the style LUT of the ACE being run: [mb][19][6][2][C]
    int batch_inv = 0;                         // option "sean.batch_invariant": 1 = every choice that follows the number of tasks of a call (F(4x4) vs F(2x2),
                                               //   split-K, sample-pair tiles at 16 pixels, the small-batch LUT / tiny-level routes) is made as for a large
                                               //   batch: sample i alone == sample i in any batch, bit for bit (exact-f32 path)
    int wino4v = 1;                            // option "sean.wino4v": 1 = F(4x4,3x3) layers with >= 512 GEMM rows at <= 64 pixels (and the Zencoder's 256 -> 512
                                               //   conv) take their input pre-transformed by one extra pass (conv_wino4v.h; bit-identical results)
    float* vbuf = nullptr;                     // the pre-transformed input V of the layer being run (conv_wino4v.h)
    size_t vbuf_bytes = 0;
    bool wino4v_fits(int Bn, int r, int nks) const;      // sean_model.cpp
    int* prof_stats = nullptr;                 // profiling: snapshots of the work-list statistics of sparse launches (16 B each)
    int prof_stats_cap = 0, prof_stats_used = 0;
    int sparse = 1;                            // option "sean.sparse" (0 = every pixel through the conv)
    int sparse_min_r = 64;                     // option "sean.sparse_min": smallest resolution served by the sparse path
    int sh16_compact = 1;                      // option "sean.sh16_compact": f16x3 path, 1 = pixel-level compaction inside the
                                               //   wave-specialised ACE kernel (3-term path) with pair / quad entries for
                                               //   sparse tiles, 2 = without those entries (A/B), 0 = tile skipping only
    int sparse_th = 0;                         // option "sean.sparse_th": tile height 8 / 16 (0 = by the layer's row tiles)
    SparseLevel sp_level[6][2];                // [level][0: tiles of 32 x 8 | 1: tiles of 32 x 16]
    std::vector<SparseWork> sp_work[6];
    // Taller tiles where a tile of 32 x 8 carries few sub-tiles: few row tiles (C <= 64: the four waves of a block split the
    // sub-tiles), and the full-resolution level of large images (measured at 512^2, B = 16: 5.26 vs 5.64 ms per C = 128 ACE;
    // at 256^2 and below the 32 x 8 tiles are faster: 5.44 vs 5.62, 4.13 vs 4.27, 2.69 vs 2.80 ms)
    int sparse_tile_h(int mtiles, int r) const {
        return sparse_th == 8 || sparse_th == 16 ? sparse_th : ((mtiles <= 2 || r >= 512) ? 16 : 8);
    }
    float* gtab = nullptr;
    float* gtab_side = nullptr;                // overlap mode: the table of the interior passes on the side stream
    std::map<std::string, float*> taps;
    // profiling
    bool prof_on = false;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;

    size_t noise_floats(int S) const;
    // returns empty string on success, else error message
    std::string build(const TensorStore& ts, int max_batch, int max_size);
    std::string generate(const uint8_t* labels, const float* codes, const float* noise, uint64_t seed, float* out,
                         int B, int S, hipStream_t stream);
    // the planes generate() draws on device when it is given noise == nullptr, written out explicitly
    std::string draw_noise(uint64_t seed, float* out, int B, int S, hipStream_t stream);
    // Zencoder (style encoder + region average pooling), architecture.py:177-207
    std::string encode(const float* img, const uint8_t* labels, float* codes_out, int B, int S, hipStream_t stream, int phase = 0);
    int enc_pending_B = 0, enc_pending_S = 0;   // split encode: set by phase 1, consumed by phase 2
    void destroy();
};

}  // namespace chk
