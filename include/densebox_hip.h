/*
 * densebox_hip.h -- C ABI of libdensebox_hip.so (MI355X / gfx950 only).
 *
 * The reference (CaptainEven/DenseBox) has NO native/FFI interface: its only
 * boundary is the Python surface of DenseBox.py.  This header is
 * therefore the boundary the build defines for the hot path; each entry point
 * names the reference lines whose arithmetic it replaces.  The Python front end
 * (densebox_amd/) binds these with ctypes.
 *
 * Conventions
 *  - every function returns 0 on success, a negative dbx_status otherwise, and
 *    never throws across the ABI; dbx_last_error() gives a thread-local message.
 *  - all pointers are DEVICE pointers owned by the caller (torch allocates);
 *    the library owns nothing, keeps no state between calls, and is re-entrant.
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *    every call is asynchronous with respect to the host.
 *  - activations are "framed NHWC": [N][H+2p][W+2p][ld] elements of the compute
 *    dtype, the p-pixel frame is zero and is never written by any kernel.
 */
#ifndef DENSEBOX_HIP_H
#define DENSEBOX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum dbx_status {
    DBX_OK = 0,
    DBX_ERR_ARG = -1,      /* bad descriptor / unsupported shape */
    DBX_ERR_HIP = -2,      /* a HIP runtime call failed (launch, memset) */
    DBX_ERR_DTYPE = -3
};

enum dbx_dtype { DBX_F16 = 0, DBX_BF16 = 1, DBX_F32 = 2 };

const char* dbx_last_error(void);
/* ABI version of the header a caller was compiled against; dbx_version() returns the library's.  Bumped whenever a struct layout,
 * an argument list or a scratch-size contract changes (a binding must refuse a library whose version differs):
 *   2  (round 3) dbx_loss_forward_backward's scratch grew from n doubles to dbx_loss_scratch_bytes(n) (mask planes + partial sums);
 *      the dbx_pack_multi job record's former pad field became rows_lim
 *   3  (round 4) fused entry points added (see the round-4 section below); nothing removed
 *   4  (round 4) dbx_head2_backward_up takes a d_hid view with a NULL ptr ("do not store it"); heads-gen entry points added
 *   5  (round 4) dbx_sgd_pack_step and dbx_heads_forward_fused_heads added; nothing changed
 *   6  (round 5) dbx_conv_plan may name DBX_K_P8 (the 8-phase kernel: plain packed weights); dbx_heads_forward_fusable returns WHICH kernel
 *      takes the fused heads forward (1 = ws / fragment-order weights as before, 2 = 8-phase / plain weights + plain second-weight image)
 *   7  (round 6) additions only: dbx_grad_guard, dbx_sgd_step_guarded, dbx_sgd_pack_step_guarded (f16 overflow guard), dbx_conv_wgrad_pool_dz; the heads-gen
 *      entry points accept DBX_F32 (reference instantiations for the parity suite) */
#define DBX_ABI_VERSION 7
int dbx_version(void);
/* device sanity: returns gfx arch number (950) of `device`, or <0 */
int dbx_device_arch(int device);

/* ------------------------------------------------------------------ framed NHWC tensor view */
typedef struct dbx_view {
    void*   ptr;      /* element (n=0, y=-pad, x=-pad, c=0) of the frame */
    int32_t n, h, w;  /* logical (unframed) extent */
    int32_t pad;      /* frame width in pixels (0, 1, or kh-1 for the un-padded refine convs' gradients) */
    int32_t ld;       /* elements per pixel (>= c_off + c) */
    int32_t c_off;    /* first channel this view addresses */
    int32_t c;        /* channels in this view */
} dbx_view;

/* ------------------------------------------------------------------ convolution (implicit GEMM on MFMA)
 * y[n,oy,ox,co] = epi( sum_{ky,kx,ci} x[n, oy+ky-cpad, ox+kx-cpad, ci] * w[co][ky][kx][ci] + bias[co] )
 * Replaces nn.Conv2d(+ReLU) at DenseBox.py:185-209 (3x3 backbone), :223-224/:455-461/:717-726
 * (1x1 heads), :466-471 (refine branch), and -- with transposed/flipped packed weights --
 * autograd's dgrad of the same layers (DenseBox.py:2186 loss.backward()).
 */
enum dbx_epilogue {
    DBX_EPI_BIAS     = 1,   /* + bias[co] (fp32) */
    DBX_EPI_RELU     = 2,   /* max(.,0) */
    DBX_EPI_GATE     = 4,   /* zero where gate[n,oy,ox,co] <= 0 (ReLU backward; gate = forward activation) */
    DBX_EPI_DROPMASK = 8,   /* * 2 * mask[m][co] (uint8 {0,1}, unframed [M][c]) -- nn.Dropout(0.5) with a caller-supplied mask */
    DBX_EPI_ACCUM    = 16,  /* y += result (dtype of y) */
    DBX_EPI_F32_NCHW = 32,  /* write fp32 NCHW [N][cv][Ho][Wo] (cv = y->c valid channels) instead of framed NHWC */
    DBX_EPI_DROPHASH = 64,  /* nn.Dropout(0.5) with the keep bit of element (m, co) = dbx_drop_keep(desc.drop_seed, m, co): no mask
                               buffer; dbx_head2_dgrad regenerates the same bits in backward */
    DBX_CONV_WFRAG   = 128  /* not an epilogue: w_packed is in MFMA-fragment order (dbx_pack_weight modes 4/5), the layout
                               dbx_conv_plan() asks for when it selects the register-streamed-weights kernel */
};

typedef struct dbx_conv_desc {
    int32_t dtype;         /* dbx_dtype of x, w and (unless F32_NCHW) y */
    int32_t kh, kw;        /* 1x1, 3x3, 5x5 */
    int32_t cpad;          /* conv padding (0 or 1); x->pad >= cpad */
    int32_t cin_pad;       /* channels per tap in the packed weight (multiple of 16 bytes worth) */
    int32_t cout_pad;      /* rows of the packed weight (multiple of 64) */
    int32_t epilogue;      /* OR of dbx_epilogue */
    uint32_t drop_seed;    /* DBX_EPI_DROPHASH: per-step seed of the counter-based keep mask */
} dbx_conv_desc;

/* packed weight: [cout_pad][ktot] elements, ktot = roundup(kh*kw*cin_pad*esize, 128 B)/esize, k = tap*cin_pad + ci */
int64_t dbx_conv_packed_elems(const dbx_conv_desc* d);
int dbx_conv_forward(const dbx_conv_desc* d, const dbx_view* x, const void* w_packed, const float* bias,
                     const dbx_view* y, const dbx_view* gate, const uint8_t* dropmask, int32_t dropmask_ld,
                     void* stream);
/* Which kernel dbx_conv_forward would run for (d, x, y), and the weight layout it wants.  The caller packs the weights
 * accordingly (w_frag != 0: dbx_pack_weight mode 4 (forward) / 5 (dgrad) and DBX_CONV_WFRAG in d->epilogue; else modes 0 / 1).
 * `name` is the kernel family and tile as it appears in a rocprofv3 trace (bench.py labels its roofline with it). */
enum dbx_conv_kernel { DBX_K_IGEMM = 1, DBX_K_DMA = 2, DBX_K_BAND = 3, DBX_K_C64 = 4, DBX_K_C8 = 5, DBX_K_WS = 6, DBX_K_P8 = 7 };
typedef struct dbx_conv_plan_t {
    int32_t kernel;        /* dbx_conv_kernel */
    int32_t tile_m, tile_n;
    int32_t w_frag;        /* 1: fragment-order weights wanted */
    char    name[64];
} dbx_conv_plan_t;
int dbx_conv_plan(const dbx_conv_desc* d, const dbx_view* x, const dbx_view* y, dbx_conv_plan_t* out);

/* Heads forward, BOTH 1x1 convs of every head in one pass over the pixels (DenseBox.py:158-162 x nh heads; round 4):
 *   hid = Dropout(Conv1x1(768 -> 512)(x))  for the nh heads side by side (d: 1x1, cin_pad 768, cout_pad 512 nh, epilogue
 *   DBX_EPI_BIAS | DBX_EPI_DROPHASH | DBX_CONV_WFRAG, w1_frag = dbx_pack_weight mode 4) -- written to `hid` as dbx_conv_forward would,
 *   out[n][off_i + m][y][x] = bias2[off_i + m] + sum_c W2_i[m][c] hid[n][y][x][512 i + c]   (fp32 NCHW, [N][sum k][H][W]; off_i = k_0 + .. + k_{i-1})
 * from the tile while it is in registers (the rounded, dropped values that land in `hid`; fp32 accumulation; fixed summation order).
 * w2_frag: the nh second weights as ONE dbx_pack_weight mode-4 image of 256 rows x (512 nh) columns, head i's k_i rows at rows
 * 0..k_i-1 (row_off 0) and columns 512 i.. (k_off 512 i).  scratch: dbx_heads_forward_fused_scratch_bytes(nh, N H W).
 * dbx_heads_forward_fusable() names the kernel that takes the call (16-bit types, k_i <= 8), 0 = none (run dbx_conv_forward twice):
 *   1 = the 1x1 register-streamed-weights kernel: d->epilogue carries DBX_CONV_WFRAG, w1_frag / w2_frag are the mode-4 images above;
 *   2 = the 8-phase kernel (round 5; ABI version 6): NO DBX_CONV_WFRAG, w1_frag = the plain dbx_pack_weight mode-0 image [512 nh][768],
 *       w2_frag = ONE plain mode-0 image of 64 rows x (512 nh) columns, head i's k_i rows at rows 0..k_i-1 (row_off 0) and columns 512 i..
 *       (k_off 512 i), zero elsewhere.  Same scratch, same outputs, same dropout masks (dbx_drop_hash32 of seed, pixel, channel / 32).
 * A caller that passes DBX_CONV_WFRAG always gets the ws kernel where that one can run the problem. */
int dbx_heads_forward_fusable(const dbx_conv_desc* d, const dbx_view* x, const dbx_view* hid, const int32_t* k, int32_t nh);
int64_t dbx_heads_forward_fused_scratch_bytes(int32_t nh, int64_t pixels);
int dbx_heads_forward_fused(const dbx_conv_desc* d, const dbx_view* x, const void* w1_frag, const float* bias1, const dbx_view* hid,
                            const void* w2_frag, const float* bias2, const int32_t* k, int32_t nh, float* out_nchw, void* scratch,
                            void* stream);
/* The same with one destination per head, as the reference's forward returns them (DenseBox.py:223-224): outs[i] = head i's own
 * contiguous fp32 [N][k_i][H][W] tensor (host array of nh device pointers).  Saves the caller the nh slice copies out of [N][sum k][H][W]. */
int dbx_heads_forward_fused_heads(const dbx_conv_desc* d, const dbx_view* x, const void* w1_frag, const float* bias1, const dbx_view* hid,
                                  const void* w2_frag, const float* bias2, const int32_t* k, int32_t nh, float* const* outs, void* scratch,
                                  void* stream);

/* 1x1 GEMM (16-bit types) with a split destination: couts [0, split_c) go to y with d->epilogue / gate, couts
 * [split_c, split_c + y2->c) to y2 with epilogue2 (plain, GATE and/or ACCUM) / gate2.  split_c = y->c, a multiple of 256.
 * Used for the data gradient of the fusion concat (torch.cat, DenseBox.py:219): one pass over the 2048-channel hidden
 * gradient feeds both the up-sampled conv4_4 branch and the conv3_4 branch. */
int dbx_conv_forward_split(const dbx_conv_desc* d, const dbx_view* x, const void* w_packed, const float* bias,
                           const dbx_view* y, const dbx_view* gate, const dbx_view* y2, const dbx_view* gate2,
                           int32_t split_c, int32_t epilogue2, void* stream);

/* dbx_conv_forward with the 2x2 / stride 2 max pooling of its output done in the epilogue (conv1_2 -> pool1,
 * DenseBox.py:186-187): `ypool` (N x H/2 x W/2 x 64) receives exactly what dbx_maxpool2x2 would make of y; with
 * write_full == 0 only the pooled map is written (inference: nothing re-reads the full-resolution map).  Exists for the
 * problems dbx_conv_pool_fusable() returns 1 for -- 16-bit 3x3 / pad 1 on congruent frames, even H and W: 64 -> 64 channels with an
 * epilogue within BIAS | RELU (the halo-tile kernel), or a layer the 8-phase kernels take (DBX_K_P8: conv2_2 -> pool2, conv3_4 ->
 * pool3, DenseBox.py:191, :204) with a (BIAS |) RELU epilogue, `ypool` N x H/2 x W/2 x y->c; anything else is DBX_ERR_ARG and the
 * caller runs the two calls.  dbx_conv_pool_fusable() answers for (d, x, y) only: a 1 holds for every `ypool` that is a frame of exactly
 * N x H/2 x W/2 pixels with c == y->c channels whose base, ld and c_off are multiples of 16 bytes (any pad >= 0) -- the call checks those
 * on the real view (DBX_ERR_ARG otherwise) -- and a 0 leaves dbx_last_error() as it was. */
int dbx_conv_pool_fusable(const dbx_conv_desc* d, const dbx_view* x, const dbx_view* y);
int dbx_conv_forward_pool(const dbx_conv_desc* d, const dbx_view* x, const void* w_packed, const float* bias,
                          const dbx_view* y, const dbx_view* ypool, int32_t write_full, void* stream);
/* ... and the arg-max nibbles of the pooled map in `idx` (layout of dbx_maxpool2x2_idx; null: not written).  With them a training
 * step needs the full-resolution conv1_2 output for nothing -- its only other reader was the pooling backward -- so write_full = 0
 * saves the 472 MB store (batch 64) and dbx_maxpool2x2_bwd_idx the 472 MB re-read.  With a ReLU epilogue (conv1_2) the nibbles are
 * bitwise those dbx_maxpool2x2_idx takes from the full map (window logic on the rounded values); without one they are taken from
 * the fp32 values before rounding (two window elements that round to the same number: the one the fp32 computation picks). */
int dbx_conv_forward_pool_idx(const dbx_conv_desc* d, const dbx_view* x, const void* w_packed, const float* bias,
                              const dbx_view* y, const dbx_view* ypool, int32_t write_full, void* idx, void* stream);

/* fp32 OIHW [co][ci][kh][kw] -> packed compute-dtype weight.
 * mode 0: forward            wp[co][tap][ci]            = w[co][ci][tap]
 * mode 1: dgrad (transposed) wp[ci][taps-1-tap][co]     = w[co][ci][tap]   (rows = ci, "cin" = co)
 * mode 4 / 5: the same two matrices in MFMA-fragment order for the register-streamed-weights 3x3 kernel (rows_pad a
 *   multiple of 128, cin_pad of 64): element (row, tap = 3 ky + kx, k) at
 *   [row / BN][3 * (k / 64) + ky][kx * 4 + (k % 64) / 16][(row % BN) / 32][32 * ((k % 16) / 8) + row % 32][k % 8],
 *   BN = 256 if rows_pad % 256 == 0 else 128, KC = cin_pad / 64 -- one 1-KiB block is the A operand of one
 *   v_mfma_f32_32x32x16 for all 64 lanes.  Same size as modes 0 / 1.
 * row_off / k_off place the tensor inside a larger packed matrix (several heads' 1x1 weights side by side); elements whose
 * destination row / column falls outside [0, rows_pad) x [0, cin_pad) are skipped, so a NEGATIVE offset packs a channel range of
 * a wider tensor (the conv4_4 / conv3_4 parts of the 768-channel head weights). */
int dbx_pack_weight(int32_t dtype, int32_t mode, const float* w_oihw, int32_t co, int32_t ci, int32_t kh, int32_t kw,
                    void* w_packed, int32_t rows_pad, int32_t cin_pad, int32_t row_off, int32_t k_off, void* stream);

/* All parameters in ONE launch (after an optimizer step).  `jobs` is a device array of
 *   struct { const float* src; void* dst; int32 co, ci, taps, mode; int64 ktot; int32 cin_pad, row_off, k_off; }
 * followed by int32 rows_lim (rows of dst for modes 0/1, 0 = unchecked; the record is 56 bytes).
 * mode 0/1/4/5 as dbx_pack_weight (4/5: ktot carries rows_pad; out-of-range destinations are skipped), mode 2 = fp32 bias copy
 * into dst[row_off ...] (co = length, ci = taps = 1). */
int dbx_pack_multi(int32_t dtype, const void* jobs, int32_t count, int64_t max_elems, void* stream);

/* The optimizer step (dbx_sgd_step, below) AND the re-packing in one launch (round 4): one job per parameter updates it -- the bits of
 * dbx_sgd_step -- while its tiles are staged and emits ALL of its packed images from them (the parameter, its gradient and its
 * momentum buffer are read once; replaces dbx_sgd_step + dbx_pack_multi after a training step).  `jobs` is a device array of 192-byte records
 *   struct { float* p; int32 pidx, co, ci, taps, ndst, tiled;
 *            struct { void* dst; int64 ktot; int32 mode, cin_pad, row_off, k_off, rows_lim, pad; } d[4]; }
 * p = the fp32 OIHW parameter; pidx = its index in `ptrs` (the table dbx_sgd_step takes: [param, grad, momentum] x count; pidx < 0:
 * no gradient this step, the job only re-packs); d[0 .. ndst) = its destinations, fields as in dbx_pack_multi's record (ndst = 0: plain
 * update); tiled = 1 where every destination is a 16-bit image with 8-aligned offsets / paddings, its K index over a multiple of 8 source
 * channels and taps <= 25 (the staged-tile path; 0 = element-wise).  max_elems = the largest co ci taps. */
int dbx_sgd_pack_step(int32_t dtype, const void* jobs, int32_t count, int64_t max_elems, float* const* ptrs, float lr, float momentum,
                      float weight_decay, int32_t first_step, void* stream);

/* heads: data gradient of the nh (<= 4) Conv1x1(512->k_h) layers behind Dropout in one rank-k streaming pass:
 * d_hid[m, 512h+c] = keep(m, 512h+c) * sum_{j<k_h} d_out[m, slot*h + j] * w2[h][j][c];  keep = 2*mask[..] (dropmask buffer),
 * 2*hash bit (use_hash, same bits as DBX_EPI_DROPHASH with drop_seed) or 1 (neither)
 * d_out: nh equal channel slots (>= 8 each); w2[h]: fp32 [k_h][512]; w2 / k are HOST arrays (DenseBox.py:158-162) */
int dbx_head2_dgrad(int32_t dtype, const dbx_view* d_out, const float* const* w2, const int32_t* k, int32_t nh,
                    const dbx_view* d_hid, const uint8_t* dropmask, int32_t dropmask_ld, int32_t use_hash, uint32_t drop_seed,
                    void* stream);
/* Weight/bias gradients of the same stage-2 head convs in one streaming pass over the hidden map:
 * dw[h] fp32 [k[h]][512] (OIHW of the 1x1 conv), db[h] fp32 [k[h]] (may be NULL).  scratch: dbx_head2_wgrad_scratch_bytes
 * (rows = N * H of the maps).  Replaces autograd's weight gradient of `nn.Conv2d(512, k, 1)` at DenseBox.py:161,168,
 * :458-461, :720-726. */
int64_t dbx_head2_wgrad_scratch_bytes(int32_t nh, int32_t rows);
int dbx_head2_wgrad(int32_t dtype, const dbx_view* d_out, const dbx_view* hid, const int32_t* k, int32_t nh,
                    float* const* dw, float* const* db, void* scratch, void* stream);
/* Both of the above in ONE pass over the pixels (the d_hid write overlaps the hid read); results are bitwise those of
 * dbx_head2_wgrad followed by dbx_head2_dgrad.  scratch as for dbx_head2_wgrad. */
int dbx_head2_backward(int32_t dtype, const dbx_view* d_out, const dbx_view* hid, const float* const* w2, const int32_t* k,
                       int32_t nh, const dbx_view* d_hid, const uint8_t* dropmask, int32_t dropmask_ld, int32_t use_hash,
                       uint32_t drop_seed, float* const* dw, float* const* db, void* scratch, void* stream);
/* The same plus d_g44 = up^T(d_hid), the transposed bilinear up-sampling (align_corners, DenseBox.py:446-449) of the hidden gradient
 * onto conv4_4's grid -- dbx_upsample_bilinear_bwd(d_hid, d_g44, no gate) -- in the same pass: the 2048-channel gradient is not read
 * back.  d_hid is bitwise dbx_head2_dgrad's and d_g44 bitwise dbx_upsample_bilinear_bwd's; dw/db are summed per image (fixed order).
 * 16-bit types run fused when d_hid is up to 64 and d_g44 up to 32 pixels wide (up-sampling by ~2); everything else runs the two passes. */
int dbx_head2_backward_up(int32_t dtype, const dbx_view* d_out, const dbx_view* hid, const float* const* w2, const int32_t* k,
                          int32_t nh, const dbx_view* d_hid, const uint8_t* dropmask, int32_t dropmask_ld, int32_t use_hash,
                          uint32_t drop_seed, float* const* dw, float* const* db, void* scratch, const dbx_view* d_g44, void* stream);
/* Round 6: a weight gradient whose dz is the backward of a 2x2 / stride 2 max pooling need not see that map in memory either:
 * dbx_conv_wgrad_pool_dz == dbx_maxpool2x2_bwd_idx(idx, dy, dz, accumulate 0, relu_gate 1) followed by dbx_conv_wgrad(dz, x, ...), bit for bit,
 * without dz (conv1_2's weight gradient behind pool1, DenseBox.py:186-187 backwards: 148 MB of dy + nibbles read instead of the 472 MB map).
 * dy: the pooled gradient (N x H/2 x W/2, same channels / channel offset as dz); idx: the arg-max nibbles of the WHOLE pooled layer
 * (dbx_maxpool2x2_idx layout, idx_channels channels per pooled pixel); dz: the shape of the un-pooled gradient on its frame (congruent with
 * x; ptr is not read -- unless write_dz != 0: then the kernel ALSO writes the un-pooled gradient map there, every pixel of the frame incl. its
 * zero halo, bit for bit dbx_maxpool2x2_bwd_idx's, for a consumer that still wants it in memory: the pooling backward's own launch goes
 * away).  Exists where dbx_conv_wgrad_pool_dz_ok() returns 1 (16-bit, even H / W, the 3x3 column-strip kernel's shapes);
 * scratch as dbx_conv_wgrad_scratch_bytes(dz, x). */
int dbx_conv_wgrad_pool_dz_ok(int32_t dtype, const dbx_view* dz, const dbx_view* x, int32_t kh, int32_t kw);
int dbx_conv_wgrad_pool_dz(int32_t dtype, const dbx_view* dy, const void* idx, int32_t idx_channels, const dbx_view* dz, const dbx_view* x,
                           int32_t kh, int32_t kw, int32_t cpad, int32_t co, int32_t ci, float* dw_oihw, float* db, void* scratch,
                           int32_t accumulate, int32_t write_dz, void* stream);
/* Round 4: the hidden gradient need not exist in memory.  d_hid = keep * scale * (d_out W2) has <= 8 input channels per head, so
 * its consumers GENERATE it, 32 channels x 32 pixels per MFMA (W2 and d_out in the compute dtype, keep bits from the forward's hash):
 *   dbx_head2_backward_up with d_hid->ptr == NULL (the view's shape still describes the map) does not store it (one-pass form
 *     only: dbx_head2_backward_up_fused() == 1; hash dropout or none; it rounds W2 to the compute dtype like the generators),
 *   dbx_heads1_wgrad_gen = dbx_conv_wgrad_slice(d_hid, x, 1x1, ...) (x on a padded frame of >= 32 columns: dbx_heads1_wgrad_gen_ok()),
 *   dbx_heads1_dgrad_gen = dbx_conv_forward(d_hid, W1^T, DBX_EPI_GATE) -> y (256 channels; w1t_frag = dbx_pack_weight mode 5 image,
 *     rows_pad 256, cin_pad 512 nh).
 * d_out: the compact [N, H, W] map (pad 0) with one slot of >= 8 channels per head, channels >= k[i] zero; w2[i]: fp32 [k[i]][512].
 * Results equal the in-memory forms up to fp32 summation order (W2 representable in the compute dtype) / its rounding (otherwise).
 * DBX_F32 (round 6): plain one-thread-per-output reference instantiations of the two generators (csrc/heads_ref_f32.hip) so that the
 * exact-fp32 path can run the 16-bit step's call structure against the reference-captured gradients; there d_out slots may be any
 * width >= k, x / y / gate any frames, w1t_frag is the PLAIN dbx_pack_weight mode 1 image (rows of 512 nh floats) and scratch is unused.
 * (Heads: DenseBox.py:158-162, :174-178; their backward is autograd's in the reference.) */
int dbx_head2_backward_up_fused(int32_t dtype, const dbx_view* hid, const dbx_view* d_g44);
int dbx_heads1_wgrad_gen_ok(int32_t dtype, const dbx_view* x, int32_t nh);
int dbx_heads1_wgrad_gen(int32_t dtype, const dbx_view* d_out, const dbx_view* x, const float* const* w2, const int32_t* k, int32_t nh,
                         int32_t use_hash, uint32_t drop_seed, int32_t ci, float* dw_oihw, int32_t dw_ci_total, int32_t dw_ci_off,
                         float* db, void* scratch, void* stream);
int dbx_heads1_dgrad_gen(int32_t dtype, const dbx_view* d_out, const float* const* w2, const int32_t* k, int32_t nh, int32_t use_hash,
                         uint32_t drop_seed, const void* w1t_frag, const dbx_view* y, const dbx_view* gate, void* stream);

/* eval-mode folding of one head, Conv1x1(768->512) -> Dropout(identity) -> Conv1x1(512->k), into a single 768->k map
 * (no non-linearity in between, DenseBox.py:158-162): w_out[k][768] = w2 w1, b_out[k] = w2 b1 + b2 (all fp32) */
int dbx_fold_heads(const float* w2, const float* b2, const float* w1, const float* b1, int32_t k, float* w_out,
                   float* b_out, void* stream);
/* Eval mode, refine branch (pool4 -> conv6_1 3x3 -> conv6_2 5x5 -> bilinear up -> conv6_3 1x1, DenseBox.py:464-471 / :729-736): nothing
 * after the pooling is non-linear, so the three convs fold into ONE un-padded 7x7 conv ci -> 1 (w_out [1][ci][7][7], b_out [1]) whose
 * single map is then up-sampled (dbx_upsample_bilinear_nchw_f32: fp32 NCHW planes, align_corners=True, ATen's arithmetic).  w1
 * [cm][ci][3][3], b1 [cm], w2 [cm][cm][5][5], b2 [cm], w3 [1][cm][1][1], b3 [1]; cm <= 64.  v_out [cm][5][5] = sum_n w3[n] w2[n][m] (the
 * 64 -> 1 fold of conv6_3 into conv6_2, which dbx_refine_backward needs as well). */
int dbx_fold_refine(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                    int32_t ci, int32_t cm, float* w_out, float* b_out, float* v_out, void* stream);
/* ... and the folded branch itself in one fp32 kernel: cat(landmarks [n][4][h][w], score [n][1][h][w]) (fp32 NCHW, the heads' outputs) ->
 * MaxPool2d(2, 2) -> that 7x7 conv -> out_small [n][1][h/2 - 6][w/2 - 6]; dbx_upsample_bilinear_nchw_f32 then gives the refined score. */
int dbx_refine_eval(const float* landmark_nchw, const float* score_nchw, int32_t n, int32_t h, int32_t w, const float* w_fold,
                    const float* b_fold, float* out_small, void* stream);
int dbx_upsample_bilinear_nchw_f32(const float* x, int32_t planes, int32_t hi, int32_t wi, float* y, int32_t ho, int32_t wo,
                                   void* stream);
/* Training: the backward pass of the same branch by the same linear structure (csrc/refine_ops.hip has the algebra).  With g = the
 * transposed up-sampling of d_refine and G1 = the folded conv's weight gradient (245 numbers + sum g), every parameter gradient of
 * conv6_1 / conv6_2 / conv6_3 is a small contraction of G1 with the weights, and the gradient of cat(landmarks, score) is the
 * transposed folded conv of g routed through the pooling arg-max.  d_refine [n][1][h][w]; landmark [n][4][h][w] / score [n][1][h][w] =
 * the heads' fp32 NCHW outputs of the forward pass; w_fold / v_fold from dbx_fold_refine; g_landmark / g_score = the incoming gradients of
 * those two heads (null = zero), out_landmark / out_score = incoming + the branch's contribution; dw* / db* in the parameters' own
 * layouts (fp32, overwritten); scratch of dbx_refine_backward_scratch_bytes(n, h, w).  All sums in a fixed order (bitwise repeatable).
 * Reference: loss.backward() through DenseBox.py:464-471 (:2186 / :2731). */
int64_t dbx_refine_backward_scratch_bytes(int32_t n, int32_t h, int32_t w);
int dbx_refine_backward(const float* d_refine, const float* landmark, const float* score, int32_t n, int32_t h, int32_t w,
                        const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                        int32_t cm, const float* w_fold, const float* v_fold, const float* g_landmark, const float* g_score, float* out_landmark,
                        float* out_score, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, void* scratch,
                        void* stream);

/* ------------------------------------------------------------------ weight gradient
 * dw[co][ci][ky][kx] (+)= sum_{n,y,x} dz[n,y,x,co] * x[n,y+ky-cpad,x+kx-cpad,ci]   (fp32 OIHW, DenseBox.py:2186)
 * db[co] = sum dz.   dz and x must live in frames of identical geometry (same n,h,w,pad).
 * `partial` is a scratch buffer of dbx_conv_wgrad_scratch_bytes(); the reduction over it is deterministic.
 */
int64_t dbx_conv_wgrad_scratch_bytes(int32_t dtype, const dbx_view* dz, const dbx_view* x, int32_t kh, int32_t kw);
/* co/ci: real channel counts of dw_oihw [co][ci][kh][kw] (the views may be wider: padded channels are dropped).
 * db may be NULL.  accumulate != 0 adds into dw/db instead of overwriting. */
int dbx_conv_wgrad(int32_t dtype, const dbx_view* dz, const dbx_view* x, int32_t kh, int32_t kw, int32_t cpad,
                   int32_t co, int32_t ci, float* dw_oihw, float* db, void* scratch, int32_t accumulate, void* stream);
/* The same with dw a column slice of a wider tensor: dw_oihw is [co][dw_ci_total][kh][kw] and the ci input channels of x go to
 * columns [dw_ci_off, dw_ci_off + ci) -- the gradient of a convolution over a channel concat (torch.cat, DenseBox.py:219),
 * computed per concatenated tensor (the two may then live on different grids: see dbx_upsample_bilinear_bwd). */
int dbx_conv_wgrad_slice(int32_t dtype, const dbx_view* dz, const dbx_view* x, int32_t kh, int32_t kw, int32_t cpad,
                         int32_t co, int32_t ci, float* dw_oihw, int32_t dw_ci_total, int32_t dw_ci_off, float* db, void* scratch,
                         int32_t accumulate, void* stream);
/* The kernel dbx_conv_wgrad runs for this problem ("wgrad_row3_kernel<bf16>" ...) and its split-K factor: the library's own
 * selection, for profile labels (no caller-side copy of the rule).  name_len >= 32. */
int dbx_conv_wgrad_plan(int32_t dtype, const dbx_view* dz, const dbx_view* x, int32_t kh, int32_t kw, char* name, int32_t name_len,
                        int32_t* splits);

/* conv1_2's data gradient with conv1_1's weight/bias gradient folded into its epilogue (DenseBox.py:185-186 backwards):
 * d = conv_transpose(dz, w) * (gate > 0) is the dz operand of the first layer's weight gradient and of nothing else, so it is
 * contracted against the 8-channel framed network input x0 inside the kernel and never written:
 *   dw[co][c][tap] (+)= sum_px d[px][co] * x0[px + tap][c]  (c < ci <= 8),  db[co] (+)= sum_px d[px][co]
 * -- what dbx_conv_forward(GATE) into a d map followed by dbx_conv_wgrad(d, x0) produce, up to fp32 summation order.
 * d: the dgrad descriptor (3x3, cpad 1, 64 -> 64, epilogue DBX_EPI_GATE, weights packed with mode 1); exists where
 * dbx_conv_dgrad_wgrad1_fusable() returns 1.  scratch: dbx_conv_dgrad_wgrad1_scratch_bytes(). */
int64_t dbx_conv_dgrad_wgrad1_scratch_bytes(void);
int dbx_conv_dgrad_wgrad1_fusable(const dbx_conv_desc* d, const dbx_view* dz, const dbx_view* gate, const dbx_view* x0);
int dbx_conv_dgrad_wgrad1(const dbx_conv_desc* d, const dbx_view* dz, const void* w_packed, const dbx_view* gate, const dbx_view* x0,
                          int32_t ci, float* dw_oihw, float* db, void* scratch, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------ layout / pooling / resampling
 * nchw_to_framed: network input X (DenseBox.py:185) fp32 NCHW -> framed NHWC compute dtype (channels padded with 0).
 * maxpool2x2:     nn.MaxPool2d(2,2) floor mode (DenseBox.py:187,191,204,465), first-max-wins like ATen.
 * maxpool2x2_bwd: gradient routed to the arg-max, times ReLU gate of the pre-pool activation.
 * upsample:       nn.Upsample(size, bilinear, align_corners=True) (DenseBox.py:213-216, :468-470).
 */
/* x_nchw has c_src channels; channels c_src..y->c-1 of the framed view are written as 0 (also used for dL/dout) */
int dbx_nchw_to_framed(int32_t dtype, const float* x_nchw, int32_t c_src, const dbx_view* y, void* stream);
/* uint8 [N][H][W][3] images -> framed network input: ((u8/255) - mean[c]) / std[c] (fp32, true divisions), i.e. torchvision
 * ToTensor + Normalize of the reference datasets (DenseBox.py:766-772) fused with the layout change.  mean3/std3: HOST floats */
int dbx_u8hwc_to_framed(int32_t dtype, const uint8_t* x_nhwc, const dbx_view* y, const float* mean3, const float* std3,
                        void* stream);
/* scatter c_src fp32 NCHW planes into channels [c_dst_off, c_dst_off+c_src) of y; other channels untouched
 * (builds cat(landmarks, score), DenseBox.py:464 / :729) */
int dbx_nchw_to_framed_ch(int32_t dtype, const float* x_nchw, int32_t c_src, const dbx_view* y, int32_t c_dst_off,
                          void* stream);
/* nslots (<= 4) fp32 NCHW tensors x_nchw[i] (k[i] planes; a null pointer = zeros) into consecutive `slot`-channel ranges of ONE framed
 * view y (y->c == nslots * slot): channels k[i] .. slot-1 of a range are zeroed.  dL/d(head outputs) of all heads in one launch. */
int dbx_nchw_to_framed_slots(int32_t dtype, const float* const* x_nchw, const int32_t* k, int32_t nslots, int32_t slot,
                             const dbx_view* y, void* stream);
int dbx_framed_to_nchw_f32(int32_t dtype, const dbx_view* x, float* y_nchw, void* stream);
/* dst[..., c_dst_off+j] += src[..., c_src_off+j], j < n_ch (gradient of the channel concat at DenseBox.py:464) */
int dbx_framed_add_ch(int32_t dtype, const dbx_view* src, int32_t c_src_off, int32_t n_ch, const dbx_view* dst,
                      int32_t c_dst_off, void* stream);
int dbx_maxpool2x2(int32_t dtype, const dbx_view* x, const dbx_view* y, void* stream);
/* x = pre-pool activation, dy = grad of the pooled map, dx = grad of the pre-pool map (same frame as x) */
int dbx_maxpool2x2_bwd(int32_t dtype, const dbx_view* x, const dbx_view* dy, const dbx_view* dx,
                       int32_t accumulate, int32_t relu_gate, void* stream);
/* Training: the forward pooling also records WHERE each maximum came from, so that the backward pass reads the pooled gradient and
 * half a byte per pooled element instead of the whole un-pooled activation (nn.MaxPool2d backward, DenseBox.py:187 / :191 / :204
 * under loss.backward(), :2186).  idx: device buffer of dbx_maxpool_idx_bytes(n, h, w, c) bytes (h, w, c of the UN-pooled map),
 * 4-byte aligned, dense [n][h/2][w/2][c/2]: one nibble per pooled element, channel c in byte c/2 (even channel = low nibble);
 * bits 0..1 = window position of the first maximum in (0,0),(0,1),(1,0),(1,1) order (ATen's tie rule), bit 2 = (maximum > 0).
 * dbx_maxpool2x2_bwd_idx gives exactly dbx_maxpool2x2_bwd's result for the map the nibbles were taken from; relu_gate uses bit 2. */
int64_t dbx_maxpool_idx_bytes(int32_t n, int32_t h, int32_t w, int32_t c);
int dbx_maxpool2x2_idx(int32_t dtype, const dbx_view* x, const dbx_view* y, void* idx, void* stream);
int dbx_maxpool2x2_bwd_idx(int32_t dtype, const void* idx, const dbx_view* dy, const dbx_view* dx,
                           int32_t accumulate, int32_t relu_gate, void* stream);
int dbx_upsample_bilinear(int32_t dtype, const dbx_view* x, const dbx_view* y, void* stream);
/* gate (optional): forward activation of dx's tensor; dx is zeroed where gate <= 0 (ReLU backward) */
int dbx_upsample_bilinear_bwd(int32_t dtype, const dbx_view* dy, const dbx_view* dx, const dbx_view* gate, void* stream);

/* ------------------------------------------------------------------ dense per-pixel loss (DenseBox.py:2023-2180 etc.)
 * One call builds the label maps (a5-a8), element-wise L2, hard-negative mining (top-k per sample),
 * mask fill + gray zones (a11-a13), the weighted sums (a14/a15) and dL/d(out) for every head.
 */
typedef struct dbx_loss_desc {
    int32_t kind;            /* 0 DenseBox (train_online), 1 DenseBoxLM (train_LM_online), 2 DenseBoxLMLOC (train_densebox_online) */
    int32_t n;               /* local batch */
    int32_t half_neg;        /* hard negatives per sample = random negatives per sample (DenseBox.py:2081) */
    float   lambda_loc, lambda_det, lambda_lm;
    int32_t use_labels;      /* _pn variants: skip label==0 patches (kind 2) */
} dbx_loss_desc;

typedef struct dbx_loss_io {
    const float* bbox;       /* [n,4] 60-space corners */
    const float* vertices;   /* [n,8] or NULL */
    const float* labels;     /* [n,1] or NULL */
    const int64_t* rand_neg; /* [n,half_neg] */
    const int64_t* lm_rand_neg; /* [4,n,1] or NULL */
    const float* score;  const float* loc;  const float* lm;  const float* rf;  const float* lmloc;   /* fp32 NCHW outputs */
    float* d_score; float* d_loc; float* d_lm; float* d_rf; float* d_lmloc;                           /* fp32 NCHW grads (may be NULL) */
    float* loss;             /* [1] */
    float* mask_cls;         /* [n,1,60,60] out (debug/parity) or NULL */
    float* mask_lm;          /* [n,4,60,60] out or NULL */
    int64_t* neg_idx;        /* [n,2*half_neg] out or NULL */
    int64_t* lm_neg_idx;     /* [4,n,2] out or NULL */
    int32_t* pos_count;      /* [n] positives per sample out or NULL */
} dbx_loss_io;

/* scratch: dbx_loss_scratch_bytes(d->n) bytes, 8-byte aligned (per-workgroup partial sums, reduced in a fixed order, and the mask planes
 * the mining kernel hands to the gradient kernel) */
int64_t dbx_loss_scratch_bytes(int32_t n);
int dbx_loss_forward_backward(const dbx_loss_desc* d, const dbx_loss_io* io, void* scratch, void* stream);
/* positives per sample from the boxes alone (a5) -- lets the host derive half_neg without reading maps back */
int dbx_count_positives(const float* bbox, const float* labels, int32_t n, int32_t* count_per_sample, void* stream);

/* label-map generators as stand-alone ops (reference function names in DenseBox.py:1556-1914) */
int dbx_init_score_map(const float* bbox, const float* labels, int32_t n, float* out, void* stream);
int dbx_init_offset_map(const float* coords, const float* labels, int32_t n, int32_t c, float* out, void* stream);
int dbx_init_lm_heatmap(const float* vertices, const float* labels, int32_t n, int32_t clamp, float* out, void* stream);
int dbx_mask_by_sel(float* mask, int32_t n, const int64_t* pos_idx, int64_t n_pos, const int64_t* neg_idx, int32_t n_neg, void* stream);
int dbx_mask_gray_zone_cls(float* mask, const float* bbox, const float* labels, int32_t n, void* stream);
int dbx_mask_gray_zone_lm(float* mask, int32_t n, const int64_t* pos_idx, int64_t n_pos, void* stream);

/* ------------------------------------------------------------------ SGD (DenseBox.py:2001-2004, torch semantics)
 * for each tensor t: g = grad + wd*p; buf = first ? g : mu*buf + g; p -= lr*buf
 * ptrs: device array of 3*count pointers {p, grad, buf}; sizes: device array of count element counts.
 */
int dbx_sgd_step(float* const* ptrs, const int64_t* sizes, int32_t count, int64_t max_size,
                 float lr, float momentum, float weight_decay, int32_t first_step, void* stream);
/* Overflow guard of 16-bit training (round 6; no reference counterpart: the reference trains in fp32, DenseBox.py:2186-2187).  The f16 step
 * keeps dL/d(pre-activation) in f16 frames and the loss is an un-normalised sum (DenseBox.py:2917): a residual that overflows 65504 there
 * reaches every weight gradient behind it as inf / NaN.  dbx_grad_guard scans the step's gradients (one flat fp32 buffer, 16-byte aligned)
 * and leaves `step_id` (> 0, increasing from step to step) in guard[0] when any element is not finite; dbx_sgd_step_guarded /
 * dbx_sgd_pack_step_guarded launched with the same guard and step_id then change NOTHING (parameters, momentum buffers and packed images
 * stay as they were) and add 1 to guard[1], the count of skipped steps a caller reads back whenever it reads the loss.  guard: two
 * int32 on the device, zero-initialised by the caller; NULL = the unguarded update.  No host synchronisation anywhere. */
int dbx_grad_guard(const float* grads, int64_t n, int32_t* guard, int32_t step_id, void* stream);
int dbx_sgd_step_guarded(float* const* ptrs, const int64_t* sizes, int32_t count, int64_t max_size, float lr, float momentum,
                         float weight_decay, int32_t first_step, int32_t* guard, int32_t step_id, void* stream);
int dbx_sgd_pack_step_guarded(int32_t dtype, const void* jobs, int32_t count, int64_t max_elems, float* const* ptrs, float lr, float momentum,
                              float weight_decay, int32_t first_step, int32_t* guard, int32_t step_id, void* stream);

/* ------------------------------------------------------------------ decode + NMS (DenseBox.py:3114-3443)
 * top-K of the score map, corner/landmark decode to float64 rows [K, 5|13], greedy NMS (keep ovr <= thresh).
 * keep[0] = count, keep[1..] = kept row indices in reference order.
 */
int dbx_detect(const float* score, const float* loc, const float* lm_heat, const float* lm_loc,
               int32_t rows, int32_t cols, int32_t K, double nms_thresh,
               double* dets, int32_t det_cols, int64_t* topk_idx, int32_t* keep, void* scratch, void* stream);
int64_t dbx_detect_scratch_bytes(int32_t rows, int32_t cols, int32_t K);
/* scratch: 5*n bytes */
int dbx_nms(const double* dets, int32_t n, int32_t det_cols, double nms_thresh, int32_t* keep, void* scratch, void* stream);

/* ---- plate rectification after decode (perspective_transform, DenseBox.py:3446-3481; OpenCV's published algorithm) ----
 * dbx_perspective_matrix: host function, cv2.getPerspectiveTransform: 3x3 row-major double map src -> dst of four (x, y)
 *   float pairs.  dbx_warp_perspective_u8: cv2.warpPerspective(img, M, (dw, dh)) with INTER_LINEAR and a zero border on an
 *   interleaved uint8 image [h][w][c] (device pointers). */
int dbx_perspective_matrix(const float* src_xy, const float* dst_xy, double* m9);
int dbx_warp_perspective_u8(const uint8_t* src, int32_t sh, int32_t sw, int32_t c, const double* m9, uint8_t* dst,
                            int32_t dh, int32_t dw, void* stream);

/* ---- data-parallel gradient exchange (new capability; the reference is single-GPU) ----
 * One process per GPU.  Rank 0 makes a 128-byte id (dbx_dp_unique_id) and hands it to the other ranks by any host channel;
 * every rank calls dbx_dp_init with its HIP device current; after each backward, dbx_dp_allreduce_sum_f32 sums the flat fp32
 * gradient buffer (or a bucket of it) over the ranks in place on `stream` -- SUM without division, because the reference
 * loss is a sum over the batch (DenseBox.py:2917).  RCCL over xGMI, resolved by dlopen at first use; the Python front end
 * (densebox_amd/dist.py) drives the same collectives through torch.distributed's "nccl" backend instead. */
int dbx_dp_unique_id(void* id128);
int dbx_dp_init(const void* id128, int32_t rank, int32_t world, void** comm);
int dbx_dp_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream);
int dbx_dp_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* DENSEBOX_HIP_H */
