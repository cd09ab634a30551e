/* srk.h -- C ABI of libsrk_gfx950.so: the MI355X (gfx950) kernels for the SR convolutional hot path.
 *
 * The reference (george-gca/sr-pytorch-lightning) has no FFI layer: its hot path is
 * `SRModel.forward()` (models/srmodel.py:156-171) calling stock torch.nn leaf modules
 * (SURVEY.md section 8(b)).  The entry points below are what a binding for that path
 * has to call; each one names the reference op it replaces.  The Python binding that
 * ships with this repo is `sr-pytorch-lightning_amd/_lib.py` (ctypes), and
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - All pointers are DEVICE pointers owned by the caller (torch tensors in practice).
 *    The library allocates nothing persistent and never frees.
 *  - Every launcher takes the HIP stream to enqueue on; it never synchronises and
 *    never touches the null stream, so it can be captured into a hipGraph.
 *  - Return value: 0 on success, otherwise a hipError_t or a negative SRK_E_* code;
 *    `srk_last_error()` returns a thread-local message.  No exceptions cross the ABI.
 *  - Launchers are re-entrant and hold no mutable global state (autograd calls
 *    backward from another thread).  The one exception is write-once and
 *    mutex-guarded: the per-device upload stream of srk_upload_prepare / _eager / _fence.
 *  - Activations are NHWC with an explicit pixel pitch and channel offset (both in
 *    elements) so that a conv can read / write a channel slice of a wider buffer
 *    (RDN dense blocks, models/rdn.py:21).  Channel counts seen by the MFMA kernels are
 *    padded to a multiple of 16 with ZERO-filled padding channels.
 *  - dtype: SRK_BF16 / SRK_F16 storage with fp32 accumulate, or SRK_F32 (fp32 MFMA,
 *    the parity mode: north_star "conv activations within 1e-3 fp32").
 */
#ifndef SRK_H
#define SRK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* srk_stream_t; /* hipStream_t */

enum { SRK_BF16 = 0, SRK_F16 = 1, SRK_F32 = 2 };

enum {
  SRK_E_BADARG = -1,   /* shape / alignment / mode not supported */
  SRK_E_NODEV = -2     /* no gfx950 device / kernel image not loadable */
};

/* output / residual addressing modes of srk_conv2d */
enum {
  SRK_OUT_NHWC = 0,      /* out[n][y][x][coff + co], dtype = args.dtype                               */
  SRK_OUT_NHWC_PS = 1,   /* conv followed by nn.PixelShuffle(r) (models/common.py:133), fused into the  */
                         /* store: out[n][y*r+i][x*r+j][coff + c]; packed channel co' = (i*r+j)*C + c   */
  SRK_OUT_PLANAR = 2     /* fp32 NCHW (the model boundary), optional PixelShuffle(r):                   */
                         /* out[n][c][y*r+i][x*r+j], co = c*r*r + i*r + j (torch channel order)         */
};

/* ---- weight packing -------------------------------------------------------------------------
 * Replaces nothing in the reference: it converts the parameters of record (OIHW fp32, the
 * state_dict layout of nn.Conv2d) into the MFMA-friendly shadow layout
 *     wpk[tap][Kin/CH][CoutP][CH]      CH = 16 bytes / sizeof(dtype)
 * forward : Kin = Cin padded to 16,  rows = Cout padded to CoutP, tap = kh*KW+kw
 * dgrad   : Kin = Cout padded to 16, rows = Cin padded to CoutP, tap flipped (conv transpose)
 * `ps_r` > 1 applies the PixelShuffle channel permutation of SRK_OUT_NHWC_PS to the Cout axis.
 * bias_pk (nullable) receives the fp32 bias padded/permuted to CoutP (forward only).           */
typedef struct {
  const float* w;      /* [Cout][Cin][KH][KW] fp32 */
  const float* bias;   /* [Cout] or NULL */
  void* wpk;           /* out */
  float* bias_pk;      /* out [CoutP] or NULL */
  int Cout, Cin, KH, KW;
  int KinP;            /* padded reduction channels (multiple of 16) */
  int CoutP;           /* padded rows */
  int dgrad;           /* 0 forward layout, 1 dgrad layout */
  int ps_r;            /* 0/1: none; r>1: NHWC pixel-shuffle permutation on the Cout axis */
  int dtype;
  int rows_layout;     /* 1 (forward, Cin == 64, KH >= 5, Cout <= 4, Cout * KW <= 32, 16-bit): BEHIND the standard layout (KH*KW*KinP*CoutP elements)   */
                       /* wpk also receives KH * 2048 elements in the (kw, co)-rows order of the direct large-kernel forward:           */
                       /* [kh][ci / 16][64 lanes][8]: lane l = MFMA row m = l % 32 = kw * Cout + co (0 beyond KW * Cout), ci = 16 ks + 8 (l / 32) + e */
} srk_pack_args;
int srk_pack_conv_weights(const srk_pack_args* a, srk_stream_t stream);
/* Same, for `n` convolutions in ONE launch: `table` is a DEVICE array of srk_pack_args (a training step re-packs
 * the forward and dgrad layouts of every conv of the model from the updated fp32 parameters).              */
int srk_pack_conv_weights_group(const srk_pack_args* table, int n, srk_stream_t stream);
/* The same launch with blocks that own one 16 x 64-channel tile of one entry each (coalesced reads through LDS).
 * srk_pack_group_tiles (host) fills tile_begin[0 .. n] -- the prefix sums of the entries' tile counts, tile_begin[n] the total --
 * from a HOST copy of the table; the launch takes device copies of both.                                              */
int srk_pack_group_tiles(const srk_pack_args* host_table, int n, int* tile_begin);
int srk_pack_conv_weights_group_tiled(const srk_pack_args* table, const int* tile_begin, int n, int total_tiles, srk_stream_t stream);

/* ---- implicit-GEMM convolution (forward and dgrad) --------------------------------------------
 * Replaces nn.Conv2d(stride 1, padding k//2) + the elementwise ops the reference issues after it:
 * DefaultConv2d (models/common.py:7-30), ReLU (common.py:99-100), `* res_scale` and `res += x`
 * (common.py:106-107, edsr.py:46-47, rcan.py:52-54,72-73,121-122, rdn.py:40,109, wdsr.py:25-26,
 * 49-50,112), nn.PixelShuffle (common.py:133, rdn.py:81,89,92, wdsr.py:88,94), torch.cat slice
 * writes (rdn.py:21,108) and the `add_mean` shift (common.py:58-71).  Epilogue order:
 *     v = acc + bias[co];  if relu: v = max(v,0);  v *= scale;  v += res;
 *     if mask and co >= mask_from: v = (mask > 0) ? v : 0;      (ReLU backward)
 *     v += post_add[c]  (planar mode only);  store.
 * The same kernel computes dgrad when given dgrad-packed weights.                                */
typedef struct {
  const void* x; int x_pitch, x_coff;  /* input NHWC, Cin channels starting at x_coff              */
  int x_ps;                            /* 0/1: plain; r>1: x is stored pixel-shuffled by r, i.e. it is */
                                       /* [N][H*r][W*r][Cin/(r*r)] and channel k = (i*r+j)*C + c     */
  int N, H, W;                         /* conv-space dims ('same' conv: output dims = input dims)    */
  int Cin;                             /* multiple of 16                                              */
  const void* wpk; const float* bias;  /* packed weights [KH*KW][Cin/CH][CoutP][CH]; bias [CoutP]/NULL */
  int CoutP;                           /* multiple of the channel tile chosen by srk_conv_tile()      */
  int Cout;                            /* channels to store (NHWC: multiple of 4, <= CoutP; planar: real) */
  int KH, KW;                          /* 1 or 3 (3x3 with one 64-channel input block in bf16/fp16 runs on   */
                                       /* the weight-stationary persistent kernel, everything else streams)   */
  int relu; float scale;
  const void* res; int res_pitch, res_coff;    /* same addressing mode and dtype as out; NULL = none */
  const void* mask; int mask_pitch, mask_coff; int mask_from;  /* NHWC/NHWC_PS addressing, dtype     */
  void* out; int out_pitch, out_coff; int out_mode; int ps_r;
  const float* post_add;               /* planar: per output channel c, or NULL                      */
  int dtype;
  int cout_real;                       /* the conv's real output channels when wpk carries the rows layout (srk_pack_args.rows_layout), else 0 */
  /* ReLU sign BITS (3x3, 64 -> 64 channels, 16-bit, NHWC out: the weight-stationary kernel; srk_conv_bits_ok): 4 bytes per pixel and
   * 32-channel half, [N*H*W][2] u32; bit i resp. 16 + i of word h <-> channel 32 h + 2 i resp. + 1 > 0 after the ReLU.
   *   relu_bits (out, nullable): written by a conv + ReLU launch (relu = 1, no res / mask / scale) next to its output;
   *   mask_bits (in, nullable, instead of `mask`): the data gradient `v = bit ? v : 0` reads these 8 bytes per pixel instead of the
   *   128-byte activation (ResBlock / RCAB backward, models/common.py:99-100: 75 MB -> 4.7 MB per launch at 256 x 48 x 48).     */
  void* relu_bits; const void* mask_bits;
} srk_conv_args;
/* 1 when srk_conv2d honours relu_bits / mask_bits for these arguments (else they must be NULL) */
int srk_conv_bits_ok(const srk_conv_args* a);
int srk_conv2d(const srk_conv_args* a, srk_stream_t stream);
/* channel tile (32/64/128) the launcher uses for a given number of output channels */
int srk_conv_tile(int Cout);

/* ---- a CHAIN of 3x3 64->64 convolutions in ONE launch, image by image (large batches) -------------------------------
 * Replaces, where the batch fills the chip with whole images (N a multiple of the CU count), the sequence of srk_conv2d launches
 * of a residual trunk -- the 2 x n_resblocks + 1 convolutions of EDSR's body (models/edsr.py:24-31,44-47; ResBlock:
 * models/common.py:74-109) in the forward pass and, given data-gradient packs, their data gradients in the backward pass.
 * A 'same' convolution of an image reads that image only, so the workgroup that owns an image runs the layers on it back
 * to back with no launch boundary and no synchronisation with other workgroups.
 *   layers_host[l] : the arguments srk_conv2d would get for layer l (layer l + 1 normally reads layer l's `out`; `res`,
 *                    `relu_bits`, `mask_bits` may name any buffer written by an EARLIER layer or before the launch);
 *   table_dev      : the same nlayers structs in device memory (srk_upload_small / srk_upload_eager), read by the kernel.
 * Every layer: 16-bit, 3x3, Cin = Cout = CoutP = 64, NHWC output, no pixel shuffle, no `mask` (sign bits only), no
 * post_add; all layers the same N, H, W, dtype and x_pitch.  A layer with KH = KW = 0 is the elementwise `out = x + res` on dense
 * 64-channel tensors (fp32 add, rounded once: the two gradient contributions of a long skip, models/edsr.py:46-47).
 * srk_conv_trunk_ok says whether a table qualifies (0: issue the layers as srk_conv2d launches).  Results are bit-identical to
 * those launches.                                                                                                          */
int srk_conv_trunk_ok(const srk_conv_args* layers_host, int nlayers);
int srk_conv_trunk(const srk_conv_args* layers_host, const void* table_dev, int nlayers, srk_stream_t stream);

/* ---- two chained 3x3 64->64 convolutions in one launch (small batches) ---------------------------
 * Replaces, at the reference's batch of 16 where a layer is one tile per CU and launches dominate, the conv pair of
 * ResBlock (models/common.py:74-109: conv, ReLU, conv, `* res_scale`, `res += x`) and of RCAB (models/rcan.py:33-55),
 * and -- given data-gradient packs -- the backward chain of either:
 *     mid = conv3x3(x, w1) + b1;  if relu_mid: mid = max(mid,0);  mid *= scale_mid;
 *     if mask: mid = (mask > 0) ? mid : 0;            (mid is rounded to the storage dtype, as the two-launch form does)
 *     out = (conv3x3(mid, w2) + b2) * scale_out + res
 * Zero padding of both convs refers to the image, exactly as two srk_conv2d calls.  All tensors NHWC with 64 channels at
 * (pitch, coff); w1 / w2 are srk_pack_conv_weights outputs for Cin = Cout = 64, b1 / b2 the packed (MFMA-row order)
 * biases or NULL.  `mid` (nullable) receives the intermediate (training keeps it for the backward pass).
 * res_from_x = 1 says res IS x (same pointer, pitch, offset): the residual is then taken from the input tile in LDS.
 * Results are bit-identical to the two-launch form.  bf16 / fp16 only.                                              */
typedef struct {
  const void* x; int x_pitch, x_coff;
  int N, H, W;
  const void* w1; const float* b1;
  const void* w2; const float* b2;
  int relu_mid; float scale_mid;
  const void* mask; int mask_pitch, mask_coff;
  void* mid; int mid_pitch, mid_coff;
  float scale_out;
  const void* res; int res_pitch, res_coff; int res_from_x;
  void* out; int out_pitch, out_coff;
  int dtype;
  /* optional channel-attention pooling of the output while it leaves (models/rcan.py:10-29 AdaptiveAvgPool2d numerator and
   * its backward): pool[(n*T + tile)*64 + c] = sum over the tile's pixels of out[.][c] * (pool_aux ? pool_aux[.][c] : 1),
   * T = srk_conv_pair_tiles(1, H, W) partial rows per sample (srk_ca_apply / srk_ca_bwd_apply take them via sums_rows /
   * gsum_rows), summed in a fixed order over the stored (rounded) values                                              */
  float* pool; const void* pool_aux; int pool_aux_pitch, pool_aux_coff;
  /* optional channel-attention step applied to the INPUT while it is loaded (0 = off); x' feeds conv 1 and goes to xo
   * (nullable); a residual must then come from memory (res_from_x = 0).  Same arithmetic and order as the stand-alone launches.
   * ca_mode 1 = srk_ca_bwd_apply: x' = x * s[n][c] + dmean[n][c] (zero outside the image), s / dmean from the MLP backward on
   *   this sample's (ca_gsum [N][ca_gsum_rows][64], ca_sums [N][ca_sums_rows][64], ca_s [N][64], ca_z [N][ca_cr], ca_w1
   *   [ca_cr][64], ca_w2 [64][ca_cr]); ca_slots (nullable): the per-sample parameter-gradient slots [dW1 | db1 | dW2 | db2]
   *   (2*64*ca_cr + ca_cr + 64 floats per sample, written by one workgroup per sample).
   * ca_mode 2 = srk_ca_apply of the PREVIOUS block: x' = x * s[n][c] + ca_x2 with s = sigmoid(W2 relu(W1 mean + b1) + b2), mean
   *   from ca_sums; ca_s_out [N][64] / ca_z_out [N][ca_cr] (nullable) receive s and the hidden activations.            */
  int ca_mode; int ca_cr;
  const float* ca_gsum; int ca_gsum_rows; const float* ca_sums; int ca_sums_rows;
  const float* ca_s; const float* ca_z; const float* ca_w1; const float* ca_w2;
  float* ca_slots;
  void* xo; int xo_pitch, xo_coff;
  const void* ca_x2; int ca_x2_pitch, ca_x2_coff;
  const float* ca_b1; const float* ca_b2; float* ca_s_out; float* ca_z_out;
} srk_conv_pair_args;
int srk_conv_pair(const srk_conv_pair_args* a, srk_stream_t stream);
/* workgroups (14x14 output tiles) such a launch has: the host uses the pair while this is about two per CU or less */
int srk_conv_pair_tiles(int N, int H, int W);

/* ---- two chained pointwise convolutions with the wide intermediate kept on chip -----------------------------------
 * Replaces the opening of WDSR's _Block_B (models/wdsr.py:30-51):
 *     h = relu(conv1x1(x; Cin -> Chid) + b1);   z = conv1x1(h; Chid -> Cmid) + b2
 * (wn(nn.Conv2d(F, 6F, 1)), ReLU, wn(nn.Conv2d(6F, int(.8F), 1))) and its backward
 *     gh = (W2^T gz) * (h > 0);   gx = W1^T gh [+ res]
 * without the Chid-channel tensor ever reaching HBM: the hidden channels are walked in slices of 64 whose activations
 * live in MFMA accumulator registers and become the next GEMM's operand in place.  Pixels are a flat list of P = N*H*W
 * NHWC rows (pitch / channel offset in elements).  Supported shapes: srk_pw_shape_ok(Cin, Chid, CoutP) -- 16-bit storage,
 * (Cin, CoutP) = (128, 128) or (64, 64), Chid a multiple of 64 (WDSR-B at n_feats 128 / 64).
 * srk_pw_pack converts the fp32 matrices of record (w1 [Chid][Cin], w2 [Cmid][Chid], biases) into the two streamed
 * layouts (sizes: srk_pw_pack_bytes(..., bwd = 0 / 1)); h is rounded to the storage dtype exactly like a stored
 * activation would be.  The backward RE-computes the pre-activation with the forward's instruction sequence, so its
 * sign is the forward's ReLU mask bit for bit; with h_out / gh_out (both or neither, [P][Chid] storage dtype) the slices
 * of h and gh are also written for the weight-gradient GEMMs (srk_conv2d_wgrad, KH = KW = 1).                       */
typedef struct {
  const float* w1; const float* b1;     /* [Chid][Cin], [Chid] or NULL */
  const float* w2; const float* b2;     /* [Cmid][Chid], [Cmid] or NULL */
  int Cin, Chid, Cmid, CoutP;           /* CoutP: padded rows of conv 2 (multiple of 64, >= Cmid) */
  void* fwd; void* bwd;                 /* out: srk_pw_pack_bytes(Cin, Chid, CoutP, 0 / 1) bytes each */
  int dtype;
} srk_pw_pack_args;
int srk_pw_shape_ok(int Cin, int Chid, int CoutP);
long long srk_pw_pack_bytes(int Cin, int Chid, int CoutP, int bwd);
int srk_pw_pack(const srk_pw_pack_args* a, srk_stream_t stream);

/* ---- weight normalisation of all weight-normed convs of a model in one launch per direction ----------------------
 * Replaces nn.utils.weight_norm (models/wdsr.py:62, applied to all 51 convs of WDSR): forward w = v * (g / ||v||) per output
 * row (torch._weight_norm, dim 0) and its backward (dv, dg from dw).  `table_dev` is a DEVICE array of jobs (srk_upload_small
 * gets it there); `row0` = number of rows of all earlier jobs, `total_rows` their sum.  forward writes w and inv (1 / ||v||,
 * kept for the backward); backward reads v, g, inv, dw and writes dv, dg.                                            */
typedef struct {
  const float* v; const float* g;   /* [rows][cols], [rows] */
  float* w; float* inv;             /* forward out: [rows][cols], [rows] */
  const float* dw;                  /* backward in */
  float* dv; float* dg;             /* backward out */
  int rows, cols, row0, pad_;
} srk_wn_job;
int srk_weight_norm_group(const srk_wn_job* table_dev, int njobs, int total_rows, int backward, srk_stream_t stream);

/* same as srk_pw_pack for `n` blocks in ONE launch; `table_dev` is a DEVICE array */
int srk_pw_pack_group(const srk_pw_pack_args* table_dev, int n, srk_stream_t stream);

typedef struct {
  const void* x; int x_pitch, x_coff;   /* [P] pixels x Cin channels */
  long long P;
  int Cin, Chid, CoutP;
  int Cout;                             /* channels of z to store (multiple of 8, <= CoutP; channels >= Cmid are zeros) */
  const void* wpk;                      /* srk_pw_pack's `fwd` */
  void* out; int out_pitch, out_coff;
  int dtype;
} srk_pw_args;
int srk_pw_forward(const srk_pw_args* a, srk_stream_t stream);

typedef struct {
  const void* x; int x_pitch, x_coff;   /* the forward's input */
  const void* gz; int gz_pitch, gz_coff; int Cz;   /* gradient of z: Cz stored channels (multiple of 8, <= CoutP) */
  long long P;
  int Cin, Chid, CoutP;
  const void* wpk;                      /* srk_pw_pack's `bwd` */
  const void* res; int res_pitch, res_coff;        /* added to gx (the skip connection's gradient) or NULL */
  void* gx; int gx_pitch, gx_coff;
  void* h_out; void* gh_out;            /* [P][Chid] or NULL (both) */
  int dtype;
} srk_pw_bwd_args;
int srk_pw_backward(const srk_pw_bwd_args* a, srk_stream_t stream);

/* weight gradients of the pointwise pair, again without h / gh in HBM (both are re-computed per 128-pixel tile):
 *     dw1[hid][in] = sum_p gh[p][hid] x[p][in];  db1[hid] = sum_p gh[p][hid];  dw2[z][hid] = sum_p h[p][hid] gz[p][z]
 * and db2[z] = sum_p gz[p][z].  Scratch: nranges = srk_pw_wgrad_ranges(P, Chid) slabs of [Chid][Cin] (dw1p), [Chid][CoutP]
 * (dw2p), [Chid] (db1p) and [CoutP] (db2p) fp32 that the launch fills and then sums in a fixed order.                   */
typedef struct {
  const void* x; int x_pitch, x_coff;
  const void* gz; int gz_pitch, gz_coff; int Cz;
  long long P;
  int Cin, Chid, Cmid, CoutP;
  const void* wpk;                      /* srk_pw_pack's `bwd` */
  float* dw1p; float* dw2p; float* db1p; float* db2p; int nranges;   /* db2p / db2: both or neither */
  float* dw1; float* db1; float* dw2; float* db2;   /* out: [Chid][Cin], [Chid] (nullable), [Cmid][Chid], [Cmid] (nullable) */
  int dtype;
} srk_pw_wgrad_args;
int srk_pw_wgrad_ranges(long long P, int Chid);
int srk_pw_wgrad(const srk_pw_wgrad_args* a, srk_stream_t stream);
/* the same in two steps, so that a backward pass with many pointwise pairs (WDSR-B: 16 blocks) finishes them with ONE launch:
 * srk_pw_wgrad_partial fills the slabs only; srk_pw_wgrad_finalize_group sums the slabs of `njobs` such calls (their argument
 * structs as a device table, e.g. written by srk_upload_small) into dw1 / db1 / dw2 / db2, `blocks_per_job` workgroups each */
int srk_pw_wgrad_partial(const srk_pw_wgrad_args* a, srk_stream_t stream);
int srk_pw_wgrad_finalize_group(const srk_pw_wgrad_args* jobs_dev, int njobs, int blocks_per_job, srk_stream_t stream);

/* ---- weight / bias gradient -------------------------------------------------------------------
 * Replaces autograd's conv weight-gradient for the convs above:
 *     dwp[tap][ci][co] += sum_{n,y,x} x[n][y+kh-ph][x+kw-pw][ci] * dy[n][y][x][co]   (fp32 atomics)
 *     dbp[co]          += sum dy
 * dwp/dbp are fp32 scratch the caller zeroes; srk_wgrad_finalize converts to OIHW.             */
typedef struct {
  const void* x; int x_pitch, x_coff; int x_ps;
  const void* dy; int dy_pitch, dy_coff; int dy_ps;
  int N, H, W;
  int Cin, Cout;            /* both multiples of 16 (padded storage channels)                     */
  int KH, KW;
  float* dwp;               /* fp32 scratch: nslabs x [KH*KW][Cin][Cout]                          */
  float* dbp;               /* fp32 scratch: nslabs x [Cout], or NULL                             */
  int nslabs;               /* value returned by srk_wgrad_slabs() for these arguments:           */
                            /*  >0: slab mode, every workgroup stores its partial sum to its own  */
                            /*      slab (no atomics, no zeroing, bitwise reproducible);          */
                            /*   0: atomic mode, ONE slab that the caller has zeroed              */
  int dtype;
  int cout_real;            /* the conv's real output channels (<= Cout), or 0 = unknown: large kernels   */
                            /* with few of them (cout_real * KW <= 32) put (kw, co) pairs on the MFMA columns */
} srk_wgrad_args;
int srk_conv2d_wgrad(const srk_wgrad_args* a, srk_stream_t stream);
/* number of slabs srk_conv2d_wgrad will write for these arguments (0 = atomic mode, see nslabs) */
int srk_wgrad_slabs(const srk_wgrad_args* a);
/* channels per (tap, ci) row of a slab (and of a dbp row): Cout, or 4 for the compact slabs of large kernels with few real output
 * channels (cout_real <= 4); the caller sizes dwp / dbp and sets srk_wgrad_finalize's CoutP with it */
int srk_wgrad_slab_cout(const srk_wgrad_args* a);

typedef struct {
  const float* dwp; const float* dbp;   /* from srk_conv2d_wgrad                                 */
  int nslabs;                           /* slabs to sum (0 or 1: a single slab)                  */
  float* dw; float* db;                 /* OIHW fp32 [Cout][Cin][KH][KW], [Cout]; db may be NULL */
  int Cout, Cin, KH, KW;                /* real sizes                                            */
  int CinP, CoutP;                      /* padded sizes used by dwp                              */
  int ps_r;                             /* channel permutation of SRK_OUT_NHWC_PS on Cout        */
  float scale;                          /* multiplies the result (res_scale, loss un-scaling)    */
  int accumulate;                       /* 0: dw = v, 1: dw += v                                 */
} srk_wgrad_fin_args;
int srk_wgrad_finalize(const srk_wgrad_fin_args* a, srk_stream_t stream);

/* ---- grouped weight gradient: every 3x3 weight gradient of a training step in ONE dispatch ------------------------
 * The weight gradients of a step are independent of each other; per layer they are 37 (EDSR-baseline) to 411 (RCAN)
 * launches of only 144 tiles at the reference's batch of 16 (configs/train_default_sr.yml:3).  The host side defers
 * them to the end of backward (torch.autograd's final callback) and issues two launches: this kernel, which walks a
 * device table of jobs with equal tile counts per workgroup whatever the layer, and srk_wgrad_finalize_group.
 *   srk_wgrad_group_ok   : 1 when `a` can be a job (16-bit, 3x3, slab mode: what srk_wgrad_slabs() > 0 means).
 *   srk_wgrad_group_plan : in : jobs[i] with dbp != NULL where a bias gradient is wanted; `scratch` NULL = sizing pass.
 *                          out: jobs[i].nslabs, *nblocks, *scratch_floats; with `scratch` (device, fp32, that many floats)
 *                               also jobs[i].dwp / dbp (slab regions inside it) and, when given, the HOST images of the
 *                               device tables: table_host [n * srk_wgrad_group_job_bytes()], block_job_host [nblocks].
 *   srk_conv2d_wgrad_group: the launch; both tables in DEVICE memory (srk_upload_small gets them there).           */
int srk_wgrad_group_ok(const srk_wgrad_args* a);
int srk_wgrad_group_job_bytes(void);
int srk_wgrad_group_plan(srk_wgrad_args* jobs, int n, float* scratch, void* table_host, int* block_job_host,
                         int* nblocks, long long* scratch_floats);
int srk_conv2d_wgrad_group(const void* table_dev, const int* block_job_dev, int nblocks, int dtype, srk_stream_t stream);
/* `n` finalizations of 3x3 gradients (KH = KW = 3) in one launch: `table_dev` is a DEVICE array of srk_wgrad_fin_args,
 * `blocks_per_job` workgroups each; slabs are summed in slab order (bitwise reproducible) */
int srk_wgrad_finalize_group(const srk_wgrad_fin_args* table_dev, int n, int blocks_per_job, srk_stream_t stream);
/* Copies a small HOST table (<= 4 MiB) to 16-byte-aligned DEVICE memory through kernel arguments: capturable into a
 * hipGraph (a replay rewrites the same bytes), no pinned staging buffer.                                          */
int srk_upload_small(void* dst_dev, const void* src_host, long long nbytes, srk_stream_t stream);
/* The same upload for a table that a hipGraph UNDER CAPTURE will own and whose memory the capture site keeps alive as long as the graph
 * (ops.static_tables): the bytes are written once, now, on a stream of the library's own (no graph nodes); srk_upload_fence() -- after the
 * capture has ended, before the first replay -- waits for them.  srk_upload_prepare() creates that stream and must run outside any capture.
 * The stream belongs to the CURRENT device of the calling thread (one per device, created once under a mutex): all three calls of one table
 * must be made with the table's device current.                                                                                          */
int srk_upload_prepare(void);
int srk_upload_eager(void* dst_dev, const void* src_host, long long nbytes);
int srk_upload_fence(void);

/* ---- input unfold (model boundary) -----------------------------------------------------------------
 * Head convs 3->F (edsr.py:21-22, rcan.py:92-93, rdn.py:57-58, wdsr.py:69-71) and WDSR's 5x5 skip conv
 * (wdsr.py:90-94) have Cin <= 4: too thin for a 16-channel MFMA K-chunk.  The boundary kernel fuses the
 * input mean shift (common.py:58-71 sign=-1, wdsr.py:103-105), the NCHW fp32 -> NHWC conversion and an
 * im2col of the K x K window:
 *     dst[n][y][x][ci*KH*KW + kh*KW + kw] = x[n][ci][y+kh-ph][x+kw-pw] - sub[ci]      (0 outside the image)
 * so the conv becomes a 1x1 conv over Cin*KH*KW (padded to 16) channels whose weight matrix is
 * w.reshape(Cout, Cin*KH*KW) -- the OIHW parameter itself -- and runs through srk_conv2d /
 * srk_conv2d_wgrad like every other layer.                                                          */
typedef struct {
  const float* x;           /* NCHW fp32 [N][Cin][H][W]                                            */
  const float* sub;         /* per-input-channel value subtracted before the conv, or NULL         */
  void* dst; int dst_pitch, dst_coff;
  int N, Cin, H, W, KH, KW;
  int Kstore;               /* channels written: >= Cin*KH*KW, multiple of 16, extra are zero      */
  int dtype;
} srk_unfold_args;
int srk_unfold_nchw(const srk_unfold_args* a, srk_stream_t stream);

/* ---- layout conversion at the model boundary ------------------------------------------------------
 * NCHW fp32 <-> NHWC dtype.  `ps_r` > 1 on to_nhwc un-shuffles: dst[n][y][x][c*r*r+i*r+j] =
 * src[n][c][y*r+i][x*r+j] (the adjoint of the fused PixelShuffle store in planar mode).        */
typedef struct {
  const float* src; void* dst; int dst_pitch, dst_coff;
  int N, C, H, W;           /* dst dims: C channels (padded with zeros up to Cstore), H x W      */
  int Cstore; int ps_r; float scale; int dtype;
} srk_to_nhwc_args;
int srk_nchw_to_nhwc(const srk_to_nhwc_args* a, srk_stream_t stream);

typedef struct {
  const void* src; int src_pitch, src_coff; float* dst;
  int N, C, H, W; int dtype;
} srk_to_nchw_args;
int srk_nhwc_to_nchw(const srk_to_nchw_args* a, srk_stream_t stream);

/* ---- RCAN channel attention -----------------------------------------------------------------------
 * CALayer.forward (models/rcan.py:10-29) fused with RCAB's `res += x` (rcan.py:52-54):
 *   sums[n][s][c] = sum over the pixels of block s of t (or t*u)   (srk_ca_pool: wave/LDS reductions, ONE plain store
 *                   per block and channel -- S = srk_ca_splits(N, HW) blocks per sample, no atomics, nothing to zero,
 *                   bitwise reproducible; the consumers add the S partials in block order)
 *   z = relu(W1 mean + b1); s = sigmoid(W2 z + b2); out = t * s + res    (srk_ca_apply)
 * and its backward (srk_ca_pool with u = g, srk_ca_bwd_apply).  W1:[Cr][C], W2:[C][Cr] are the raw
 * fp32 parameters conv_du.0.weight / conv_du.2.weight.                                         */
int srk_ca_splits(int N, int HW);
typedef struct {
  const void* t; int t_pitch, t_coff;
  const void* u; int u_pitch, u_coff;   /* NULL: plain sum of t; else sum of t*u (backward)      */
  float* sums;                          /* [N][S][C] fp32 partial sums, S = srk_ca_splits(N, HW) */
  int N, HW, C; int dtype;
} srk_ca_pool_args;
int srk_ca_pool(const srk_ca_pool_args* a, srk_stream_t stream);

typedef struct {
  const void* t; int t_pitch, t_coff;
  const void* res; int res_pitch, res_coff;     /* nullable */
  const float* sums;                    /* [N][S][C] from srk_ca_pool                            */
  const float* w1; const float* b1; const float* w2; const float* b2;
  float* s_out; float* z_out;           /* [N][C], [N][Cr] saved for backward (nullable)         */
  void* out; int out_pitch, out_coff;
  int N, HW, C, Cr; int dtype;
  int sums_rows;                        /* partial rows per sample in `sums`; 0 = srk_ca_splits(N, HW) (srk_ca_pool's)   */
} srk_ca_apply_args;
int srk_ca_apply(const srk_ca_apply_args* a, srk_stream_t stream);

typedef struct {
  const void* g; int g_pitch, g_coff;   /* gradient w.r.t. the CA output (t*s)                   */
  const float* gsum;                    /* [N][S][C] = partial sums of g*t from srk_ca_pool(t,u=g) */
  const float* sums; const float* s; const float* z;   /* saved by forward                      */
  const float* w1; const float* w2;
  float* dw1; float* db1; float* dw2; float* db2;      /* out: PER-SAMPLE contributions; each pointer is the   */
                                        /* sample-0 slot, sample n is 2*C*Cr + Cr + C floats further; the  */
                                        /* caller sums over n (fixed order: reproducible)                  */
  void* gt; int gt_pitch, gt_coff;      /* out: gradient w.r.t. t = g*s + dmean/HW               */
  int N, HW, C, Cr; int dtype;
  int sums_rows, gsum_rows;             /* partial rows per sample in `sums` / `gsum`; 0 = srk_ca_splits(N, HW)          */
} srk_ca_bwd_args;
int srk_ca_bwd_apply(const srk_ca_bwd_args* a, srk_stream_t stream);
/* dst[i] = sum over r < n of src[r*k + i] (row order: reproducible) for `njobs` entries of a DEVICE table in one launch: the
 * sums over the batch of the per-sample slots srk_ca_bwd_apply writes, for every CALayer of a backward pass at once
 * (replaces the conv_du weight / bias gradient reductions autograd performs per layer, models/rcan.py:10-29).        */
typedef struct { const float* src; float* dst; int n; int k; } srk_rowsum_job;
int srk_rowsum_group(const srk_rowsum_job* table_dev, int njobs, int max_k, srk_stream_t stream);

/* ---- data step feeding the path ------------------------------------------------
 * _SRDataset._get_item / _get_patch in 'train' mode (srdata.py:64-91,137-169): crop an LR patch and the matching
 * HR patch, rotate both counter-clockwise by rot*90 degrees, horizontal flip, vertical flip, uint8 -> float / 255,
 * HWC -> CHW.  One launch for a whole batch: `table` is a DEVICE array of per-sample descriptors (images stay on
 * the GPU as uint8 HWC), outputs are the NCHW fp32 batch tensors that SRModel.training_step consumes.        */
typedef struct {
  const uint8_t* lr; const uint8_t* hr;   /* uint8 [H][W][C] images of this sample (device)               */
  int lr_h, lr_w, hr_h, hr_w;
  int top, left;                          /* LR patch origin (row, column); the HR origin is scale * that  */
  int rot;                                /* 0..3 quarter turns counter-clockwise                          */
  int hflip, vflip;
} srk_patch_desc;
typedef struct {
  const srk_patch_desc* table; int N; int C; int patch_lr; int scale;
  float* lr_out;                          /* [N][C][patch_lr][patch_lr]                                    */
  float* hr_out;                          /* [N][C][patch_lr*scale][patch_lr*scale]                        */
} srk_patch_args;
int srk_sample_patches(const srk_patch_args* a, srk_stream_t stream);

/* ---- metric core on the device ---------------------------------------------------
 * validation_step clamps both images to [0,1] and calls piq.psnr (srmodel.py:224-232,582).  This reduction
 * returns per image the sum of squared differences of the clamped images and the element count, either over all
 * channels (RGB PSNR, the reference's definition) or over the BT.601 luma with a `shave`-pixel border removed
 * (PSNR-Y, the convention BASELINE.json names).  PSNR = 10 log10(count / sse) is left to the caller.        */
typedef struct {
  const float* sr; const float* hr;       /* NCHW fp32                                                     */
  int N, C, H, W;
  int luma;                               /* 0: all channels; 1: BT.601 Y of the 3 channels                */
  int shave;                              /* border removed on every side (luma or not)                    */
  double* sse;                            /* [N] sums, caller-zeroed                                       */
} srk_sse_args;
int srk_image_sse(const srk_sse_args* a, srk_stream_t stream);

/* ---- L1 loss of the training step (reference srmodel.py:160-171 -> F.l1_loss, 'mean' reduction), fused:
 * forward reads sr and hr once, writes per-block partial sums of |sr - hr| (summed in a fixed order by the caller:
 * reproducible) and the sign of (sr - hr) as int8; backward turns the sign map into the gradient
 * sign * (*gout) * scale without touching sr / hr again.  gout is a DEVICE scalar (no host sync, graph-capturable). */
typedef struct srk_l1_args {
  const float* sr;                        /* forward: [n] ; backward: unused                                */
  const float* hr;
  long long n;
  signed char* sign;                      /* [n] sign(sr - hr) in {-1, 0, 1}: written by forward, read by backward */
  double* partial;                        /* forward: [srk_l1_blocks(n)] partial sums                       */
  const float* gout;                      /* backward: device scalar                                        */
  float scale;                            /* backward: weight / n                                           */
  float* grad;                            /* backward: [n]                                                  */
} srk_l1_args;
int srk_l1_blocks(long long n);
int srk_l1_loss_fwd(const srk_l1_args* a, srk_stream_t stream);
int srk_l1_loss_bwd(const srk_l1_args* a, srk_stream_t stream);
/* *out = (partial[0] + ... + partial[nb-1]) / n: the forward's partial sums -> the loss value, one launch */
int srk_l1_loss_mean(const double* partial, int nb, long long n, float* out, srk_stream_t stream);

/* ---- SSIM with piq.ssim's defaults (reference srmodel.py:52-53,567-593 -> piq.ssim): images are average-pooled by
 * `pool` = max(1, round(min(H, W) / 256)) (floor division of the extent, as F.avg_pool2d), filtered with the separable
 * 11-tap Gaussian (sigma), and the SSIM map of the VALID region ((Hp-10) x (Wp-10)) is summed per (image, channel)
 * plane.  The caller divides by the map size and averages over channels and images.  Inputs are used as given
 * (the model clamps before its metrics).  fp32 NCHW planar. ------------------------------------------------------ */
typedef struct srk_ssim_args {
  const float* x;                         /* [N][C][H][W]                                                   */
  const float* y;
  int N, C, H, W;
  int pool;                               /* >= 1                                                           */
  float sigma, k1, k2;                    /* 1.5, 0.01, 0.03                                                */
  double* sums;                           /* [N*C] sums of the SSIM map, caller-zeroed                      */
} srk_ssim_args;
int srk_image_ssim(const srk_ssim_args* a, srk_stream_t stream);

/* ---- remaining conv models: SRResNet and DDBPN ---------------------------------------------------
 * im2col / col2im on NHWC tensors for any kernel size K, stride and zero padding:
 *   cols[n][oy][ox][(kh*K + kw)*C + c] = x[n][oy*stride + kh - pad][ox*stride + kw - pad][c]          (srk_unfold_nhwc)
 *   out[n][y][x][c] = bias[c] + sum of cols[n][iy][ix][(kh*K + kw)*C + c] over y = iy*stride - pad + kh, x = ...  (srk_fold_nhwc)
 * nn.Conv2d(k, stride, padding) (models/ddbpn.py:10-24 `projection_conv(up=False)`, the 9x9 tail conv of
 * models/srresnet.py:24-29) = srk_unfold_nhwc -> 1x1 srk_conv2d with the weight as a [Cout][K*K*Cin] matrix;
 * nn.ConvTranspose2d (ddbpn.py `projection_conv(up=True)`) = 1x1 srk_conv2d to K*K*Cout channels -> srk_fold_nhwc.
 * Each kernel is the other's adjoint, so the data gradients use the same pair.                                   */
typedef struct {
  const void* x; int x_pitch, x_coff;     /* NHWC input, C channels from x_coff                                   */
  void* cols; int cols_pitch;             /* out [N][Ho][Wo][>= K*K*C]                                             */
  int N, H, W, C;                         /* C: multiple of 16 bytes / sizeof(dtype)                               */
  int K, stride, pad;
  int Ho, Wo;                             /* (H + 2 pad - K) / stride + 1                                           */
  int dtype;
} srk_unfold_nhwc_args;
int srk_unfold_nhwc(const srk_unfold_nhwc_args* a, srk_stream_t stream);

typedef struct {
  const void* cols; int cols_pitch;       /* [N][Hi][Wi][>= K*K*C]                                                  */
  const float* bias;                      /* [C] added once per output element, or NULL                             */
  void* out; int out_pitch, out_coff;     /* NHWC [N][Ho][Wo][C]                                                    */
  int N, Hi, Wi, C;
  int K, stride, pad;
  int Ho, Wo;                             /* ConvTranspose2d: (Hi - 1)*stride - 2 pad + K                           */
  int dtype;
} srk_fold_nhwc_args;
int srk_fold_nhwc(const srk_fold_nhwc_args* a, srk_stream_t stream);

/* D-DBPN's projection convolutions at scale 4, directly (csrc/proj.hip): nn.Conv2d / nn.ConvTranspose2d(32, 32, kernel 8, stride 4,
 * padding 2) -- /root/reference models/ddbpn.py:10-24 `projection_conv` as `DenseProjection` uses it (ddbpn.py:42-53) -- on NHWC
 * 16-bit tensors, no column tensor.  Replaces the cuDNN strided / transposed convolutions behind those modules and their autograd.
 * One weight convention for both module kinds: w4[cl][ch][ky][kx], cl = channel on the LOW-resolution side, ch = channel on the
 * HIGH-resolution side (Conv2d's [out][in][ky][kx], ConvTranspose2d's [in][out][ky][kx]).  (N, H, W) are the LOW-resolution dims;
 * the other tensor is [N][4H][4W].
 *   srk_proj_down : HR -> LR   out[q][cl] = bias[cl] + sum xh[4q - 2 + k][ch] w4[cl][ch][k]     (Conv2d forward; ConvTranspose2d dgrad)
 *   srk_proj_up   : LR -> HR   out[4q - 2 + k][ch] += x[q][cl] w4[cl][ch][k], + bias[ch]         (ConvTranspose2d forward; Conv2d dgrad)
 *   srk_proj_wgrad: dw4[cl][ch][k] (=, +=) sum_q g[q][cl] xh[4q - 2 + k][ch]                     ((xh, g) = (x, dy) resp. (dy, x))
 * srk_proj_pack turns fp32 w4 into the MFMA fragment order of both directions: wpk (srk_proj_pack_bytes() bytes, 16-byte aligned)
 * holds the `down` fragments first, the `up` fragments at byte srk_proj_pack_bytes() / 2.                                     */
typedef struct {
  const void* x; int x_pitch;             /* input NHWC, 32 channels used, pitch in elements                                */
  void* out; int out_pitch;
  const void* wpk;                        /* the direction's half of srk_proj_pack's output                                 */
  const float* bias;                      /* [32] or NULL                                                                   */
  int N, H, W;                            /* low-resolution dims                                                            */
  int dtype;                              /* SRK_BF16 / SRK_F16                                                             */
  const float* slope; int slope_stride;   /* fused nn.PReLU (ddbpn.py:42-53: every projection conv is followed by one), or NULL: */
  void* pre; int pre_pitch;               /* out = prelu(stored conv output), pre (or NULL) = the stored conv output, for its backward */
} srk_proj_args;
long long srk_proj_pack_bytes(void);
int srk_proj_pack(const float* w4, void* wpk, int dtype, srk_stream_t stream);
/* the same for n (weights, packed buffer) pairs in ONE launch: `table_dev` is a device array (D-DBPN: its 33 projections per step) */
typedef struct { const float* w4; void* wpk; } srk_proj_pack_job;
int srk_proj_pack_group(const srk_proj_pack_job* table_dev, int n, int dtype, srk_stream_t stream);
int srk_proj_down(const srk_proj_args* a, srk_stream_t stream);
int srk_proj_up(const srk_proj_args* a, srk_stream_t stream);
typedef struct {
  const void* xh; int xh_pitch;           /* high-resolution operand [N][4H][4W][>= 32]                                     */
  const void* g; int g_pitch;             /* low-resolution operand [N][H][W][>= 32]                                        */
  float* scratch;                         /* srk_proj_wgrad_scratch_floats(N, H, W) floats                                  */
  float* dw;                              /* [32][32][8][8] fp32                                                            */
  int accumulate;                         /* 1: dw += (an existing gradient)                                                */
  int N, H, W;
  int dtype;
  float* db;                              /* bias gradient [32] (sum of the upstream gradient over its pixels), or NULL     */
  int bias_side;                          /* which operand is the upstream gradient: 1 = g (Conv2d), 2 = xh (ConvTranspose2d) */
  int db_accumulate;
} srk_proj_wgrad_args;
long long srk_proj_wgrad_scratch_floats(int N, int H, int W);
int srk_proj_wgrad(const srk_proj_wgrad_args* a, srk_stream_t stream);

/* Per-channel partial sums over P pixels of an NHWC tensor (fp32 accumulate), one plain store per block and channel:
 * partial[b][0][c], partial[b][1][c] for b < srk_chan_stats_blocks(P); the caller adds the blocks in order.
 *   mode 0: sum x, sum x^2          nn.BatchNorm2d batch statistics (srresnet.py:16-21 via common.py:97-98) in ONE pass over
 *                                   values shifted by the tensor's first pixel (shift_out != NULL), or one pass for the mean
 *                                   and a second pass with shift = mean for the variance
 *   mode 1: sum y, sum x*y          BatchNorm backward (y = upstream gradient)
 *   mode 2: sum over x <= 0 of x*y  nn.PReLU slope gradient (partial[b][1] = 0)                                   */
typedef struct {
  const void* x; int x_pitch, x_coff;
  const void* y; int y_pitch, y_coff;     /* modes 1, 2                                                             */
  long long P; int C;                     /* C <= 256                                                               */
  int mode;
  float* partial;                         /* [blocks][2][C]                                                         */
  int dtype;
  const float* shift;                     /* [C] or NULL: x is replaced by x - shift[c] in every sum (second pass of a  */
                                          /* two-pass variance, centred BatchNorm backward: no cancellation in fp32)   */
  float* shift_out;                       /* [C] or NULL (then `shift` must be NULL): shift[c] = x[pixel 0][c], also    */
                                          /* stored here -- the shifted-data variance: E[(x-K)^2] - (E[x-K])^2 with K a */
                                          /* value of the data loses no digits to |mean| >> std                         */
  void* gate_out; int gate_pitch;         /* mode 2 only, or NULL: gate_out[p][c] = y * (x > 0 ? 1 : slope[c]) -- nn.PReLU's */
  const float* slope; int slope_stride;   /* input gradient written by the pass that sums its slope gradient (one read of x, y) */
  /* mode 3: nn.BatchNorm2d's backward BEHIND an nn.PReLU (srresnet.py:16-21) in one pass: x = the BatchNorm's input, y = the gradient
   * of the PReLU's output; with yb = gate_a x + gate_d (the BatchNorm's output, recomputed) and gp = y (yb > 0 ? 1 : slope):
   * partial[b][0] = sum gp, partial[b][1] = sum (x - shift) gp, partial2[b] = sum over yb <= 0 of yb y (the slope's gradient)   */
  const float* gate_a; const float* gate_d;   /* [C] fp32: the forward scale / shift (srk_chan_finalize mode 1, rows 2 and 3)   */
  float* partial2;                        /* [blocks][C]                                                                    */
} srk_chan_stats_args;
int srk_chan_stats_blocks(long long P);
int srk_chan_stats(const srk_chan_stats_args* a, srk_stream_t stream);
/* The per-channel step between srk_chan_stats and srk_chan_apply of nn.BatchNorm2d (models/common.py:97-98 in srresnet.py:16-21)
 * and nn.PReLU, as one launch: the ordered sum over the blocks of `partial` [nblocks][2][C] and the vector arithmetic.
 *   mode 0: out[0] = mean = sum0 / M
 *   mode 1 (sums of x - K, K = `mean`: the mean of a first pass, or srk_chan_stats' shift_out): the batch mean = K + sum0/M
 *           (out row 4), var = max(sum1/M - (sum0/M)^2, 0); running_mean / running_var (nullable, [Creal]) updated with
 *           `momentum` (unbiased variance, like torch); out rows: invstd = rsqrt(var + eps), gamma (weight padded with 0),
 *           a = gamma * invstd, d = beta - mean * a, mean  (srk_chan_apply: out = a x + d)
 *   mode 2 (sum0 = sum dy, sum1 = sum (x - mean) dy; batch statistics): out rows: dgamma = invstd sum1, dbeta = sum0,
 *           a = gamma invstd, b = -a invstd dgamma / M, d = -a dbeta / M - b mean     (dx = a dy + b x + d)
 *   mode 3 (running statistics): out rows: dgamma, dbeta, a = gamma invstd
 *   mode 4: out[0][c] = sum0 (PReLU slope gradients); total = 1: one number, summed over the channels too; `dgamma_acc`
 *           (nullable, [Creal] or [1]): the slope's existing .grad buffer, += the result                               */
typedef struct {
  const float* partial; int nblocks; int C; int Creal;
  int mode; int total;
  float M, eps, momentum;
  const float* mean; const float* invstd; const float* gamma;
  const float* weight; const float* bias;
  float* running_mean; float* running_var;
  float* out;                             /* [rows][C] fp32 */
  long long* nbt;                         /* mode 1, nullable: nn.BatchNorm2d.num_batches_tracked, += 1                          */
  float* dgamma_acc; float* dbeta_acc;    /* modes 2 / 3, nullable, [Creal]: the parameters' existing .grad buffers, += dgamma / dbeta */
  const float* partial2; int total2;      /* modes 2 / 3, nullable: srk_chan_stats mode 3's third sums [nblocks][C] -> out row 5 = the  */
  float* dslope_acc;                      /* PReLU slope gradient (total2: ONE number, summed over the channels); += into dslope_acc    */
} srk_chan_finalize_args;
int srk_chan_finalize(const srk_chan_finalize_args* a, srk_stream_t stream);
/* srk_chan_stats and srk_chan_finalize as ONE launch: the block that finishes last does the finalize step.  `f->partial` and
 * `f->nblocks` are taken from `a`; `counter`: one int in device memory, 0 before the first use (the kernel leaves it 0) and not shared
 * by launches that may run at the same time.                                                                          */
int srk_chan_stats_finalize(const srk_chan_stats_args* a, const srk_chan_finalize_args* f, int* counter, srk_stream_t stream);

/* out[p][c] = post( (a[c]*x + b[c]*y + d[c]) * gate ),  gate = (z > 0 ? 1 : slope[c*slope_stride]) when z is given,
 * post = PReLU with the same slope when post_prelu.  a / b / d NULL = 1 / 1 / 0; y NULL = no second input.
 * BatchNorm apply (+ residual in y), BatchNorm backward (x = dy, y = saved input), nn.PReLU forward and backward.  */
typedef struct {
  const void* x; int x_pitch, x_coff;
  const void* y; int y_pitch, y_coff;
  const void* z; int z_pitch, z_coff;
  const float* a; const float* b; const float* d;   /* [C] fp32                                                     */
  const float* slope; int slope_stride;    /* nn.PReLU weight: stride 1 = per channel, 0 = one shared value           */
  int post_prelu;
  void* out; int out_pitch, out_coff;
  long long P; int C;
  int dtype;
  const float* gate_a; const float* gate_d;   /* [C] or NULL.  Given (with z and slope): the gate's argument is gate_a z + gate_d and   */
                                              /* only the first term passes it: out = a x gate + b y + d (BatchNorm backward behind a  */
                                              /* PReLU: x = dy, y = z = the BatchNorm's input)                                          */
} srk_chan_apply_args;
int srk_chan_apply(const srk_chan_apply_args* a, srk_stream_t stream);

/* ---- the last upsampling stage and the tail conv as ONE linear map (csrc/hr_tail.hip) ----------------------------------------
 * Replaces, on the 16-bit path, the end of EDSR / RCAN / RDN: UpscaleBlock's last `conv3x3(Ci -> 4C) -> nn.PixelShuffle(2)` stage
 * (models/common.py:112-139, no activation on this path) followed by the tail `conv3x3(C -> O)` (edsr.py:37-38,49-52, rcan.py:83-87,
 * 102-104, rdn.py:85-95), and their autograd.  Two convolutions with nothing non-linear between them are one linear map of the
 * upsampler's input X: at X's resolution, with the 4 sub-pixel positions (a, b) of an output pixel as channels k = o*4 + a*2 + b,
 *     T[(2y+a, 2x+b)][o] = beff[k] + sum_{ci, f in [-2,2]^2} weff[k][ci][f] X[(y,x) + f][ci]  -  (border terms on the outermost ring)
 * a 5x5 convolution Ci -> 4O (srk_conv2d / srk_conv2d_wgrad: the direct large-kernel kernels) whose weights are sums of products
 * Wt Wu.  The C-channel tensor at the doubled resolution (1.2 GB at EDSR-baseline's batch of 256) and its gradient never exist.
 * The border terms make the result EXACT: the tail conv pads the shuffled tensor with zeros, so output pixels of the outermost ring
 * lose the taps that leave the image; those terms are linear in X too (one row / column of 5 taps per edge, one tap per corner).
 *   srk_hrtail_collapse   : (wt, bt, wu, bu) -> weff, beff, wedge, bedge, wcor, bcor                          (parameter-sized)
 *   srk_hrtail_edge_fwd   : out -= border terms (out: what srk_conv2d wrote with weff, planar fp32 + PixelShuffle(2))
 *   srk_hrtail_edge_bwd_x : dx -= (border terms)^T g    (dx: what the data-gradient srk_conv2d wrote)
 *   srk_hrtail_edge_bwd_w : correlations of g and x over the edge rows / columns / corner pixels -> eedge, e0, ecor, k0
 *   srk_hrtail_expand     : (r = dL/dweff, r0 = dL/dbeff from srk_conv2d_wgrad; eedge ...) -> dwt, dbt, dwu, dbu  (parameter-sized)
 * Layouts (fp32): weff [4O][Ci][5][5]; wedge / eedge [4][2O][Ci][5] and bedge / e0 [4][2O]: edge 0 top, 1 bottom (row kk = o*2 + b, 5
 * taps along x), 2 left, 3 right (kk = o*2 + a, 5 taps along y); wcor / ecor [4][O][Ci] and bcor / k0 [4][O]: corner a*2 + b.
 * wu's output channels are in torch's PixelShuffle order c*4 + i*2 + j.  O <= 4.                                            */
typedef struct {
  const float* wt; const float* bt;     /* tail conv [O][C][3][3], [O] or NULL                                    */
  const float* wu; const float* bu;     /* upsampler conv [4C][Ci][3][3], [4C] or NULL                            */
  int O, C, Ci;
  float* weff; float* beff;             /* collapse: out; the launches in between: srk_pack_conv_weights' input   */
  float* wedge; float* bedge; float* wcor; float* bcor;
  const void* x; int x_pitch;           /* NHWC 16-bit [N][H][W][>= Ci]                                           */
  int N, H, W; int dtype;
  float* out;                           /* edge_fwd: [N][O][2H][2W] fp32, corrected in place                      */
  const float* g;                       /* edge_bwd_*: gradient of out, [N][O][2H][2W] fp32                       */
  void* dx; int dx_pitch;               /* edge_bwd_x: NHWC 16-bit [N][H][W][>= Ci], corrected in place           */
  float* eedge; float* e0; float* ecor; float* k0;   /* edge_bwd_w: out; expand: in                               */
  float* scratch;                       /* edge_bwd_w: srk_hrtail_scratch_floats(N, Ci) floats                    */
  const float* r; const float* r0;      /* expand: dL/dweff [4O][Ci][5][5], dL/dbeff [4O]                         */
  float* dwt; float* dbt; float* dwu; float* dbu;    /* expand: out (dbt / dbu nullable)                          */
} srk_hrtail_args;
int srk_hrtail_collapse(const srk_hrtail_args* a, srk_stream_t stream);
int srk_hrtail_edge_fwd(const srk_hrtail_args* a, srk_stream_t stream);
int srk_hrtail_edge_bwd_x(const srk_hrtail_args* a, srk_stream_t stream);
long long srk_hrtail_scratch_floats(int N, int Ci);
int srk_hrtail_edge_bwd_w(const srk_hrtail_args* a, srk_stream_t stream);
int srk_hrtail_expand(const srk_hrtail_args* a, srk_stream_t stream);

/* ---- optimizer: Adam over every parameter tensor in one launch -----------------------------------
 * Replaces torch.optim.Adam(...).step() as models/srmodel.py:145-154 configures it (torch defaults; :602-603 drops every
 * user-supplied hyper-parameter).  fp32 parameters, gradients and moments.  The tensors are described by a DEVICE table
 * (srk_adam_slot), the work by a device list of blocks (srk_adam_block: `count` elements of tensor `slot` from `start`);
 * the first and second moments of all tensors live in two flat buffers `m` / `v`, tensor i at `state_off` floats.
 *     t = steps[step_idx] + 1;  g' = maximize ? -g : g;  g' += weight_decay * p;
 *     m += (g' - m) * (1 - beta1);  v = beta2 * v + (1 - beta2) * g' * g';
 *     p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * `steps` (device floats: the number of updates each tensor has had -- torch counts per parameter, a tensor without a
 * gradient skips the step) are advanced by the call itself (a second tiny launch) for the tensors in the table, so a
 * captured hipGraph can replay it; `ticket` is reserved (may be NULL).                                                   */
typedef struct { float* p; const float* g; long long state_off; long long n; long long step_idx; } srk_adam_slot;
typedef struct { int slot; int count; long long start; } srk_adam_block;
typedef struct {
  const srk_adam_slot* slots; const srk_adam_block* blocks; int nslots, nblocks;
  float* m; float* v;
  float* steps; unsigned int* ticket;
  float lr, beta1, beta2, eps, weight_decay; int maximize;
  float one_minus_beta1, one_minus_beta2;   /* 1 - beta rounded from double, as torch forms them */
} srk_adam_args;
int srk_adam_step(const srk_adam_args* a, srk_stream_t stream);
/* The same update under DYNAMIC LOSS SCALING kept on the device (fp16 training: the reference's `precision: 16` runs Lightning's
 * "16-mixed" = autocast + torch.amp.GradScaler, configs/all.yml:122; GradScaler decides on the HOST whether to call
 * optimizer.step, which a hipGraph replay cannot do).  scaler_state: 8 device floats {scale, growth tracker, found_inf, growth
 * factor, backoff factor, growth interval, skipped steps, reserved}.  The gradients in the table are still multiplied by `scale`:
 *   srk_adam_step_scaled : found_inf |= any gradient not finite; if not found_inf: the update on g / scale, step counts += 1
 *                          (a step with a non-finite gradient changes NOTHING, like GradScaler.step)
 *   srk_loss_scale_update: once per step behind the last parameter group: found_inf ? scale *= backoff, tracker = 0
 *                          : (++tracker == interval ? scale *= growth, tracker = 0);  found_inf = 0     (GradScaler.update)      */
int srk_adam_step_scaled(const srk_adam_args* a, float* scaler_state, srk_stream_t stream);
/* the two halves of srk_adam_step_scaled for optimizers with SEVERAL parameter groups: GradScaler.step checks every group's
 * gradients before it updates any (torch/amp/grad_scaler.py: `found_inf` over all groups, then optimizer.step or nothing), so
 * call srk_adam_check_scaled for every group first, then srk_adam_update_scaled for every group, then srk_loss_scale_update */
int srk_adam_check_scaled(const srk_adam_args* a, float* scaler_state, srk_stream_t stream);
int srk_adam_update_scaled(const srk_adam_args* a, const float* scaler_state, srk_stream_t stream);
int srk_loss_scale_update(float* scaler_state, srk_stream_t stream);

/* ---- misc ------------------------------------------------------------------------------------------ */
const char* srk_last_error(void);
int srk_version(void);
/* number of compute units of the current device (0 when no device) */
int srk_device_cus(void);

#ifdef __cplusplus
}
#endif
#endif /* SRK_H */
