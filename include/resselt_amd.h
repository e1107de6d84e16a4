/*
 * resselt_amd.h — C-ABI of the MI355X (gfx950) super-resolution forward-pass engine.
 *
 * The reference (rewaifu/resselt) has NO FFI: its hot path is `model.forward(x)` of
 * nn.Modules that delegate every operation to PyTorch ATen.  Each entry point below
 * names the reference ATen op sequence (file:line under /root/reference) that it
 * replaces.  The library is loaded with ctypes by `resselt_amd/engine/lib.py`; the
 * binding a maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  All data pointers are DEVICE pointers
 *     owned by the caller (PyTorch caching allocator); the library never allocates,
 *     never synchronises and keeps no global state besides a thread-local error string.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *   - return value: 0 = ok, < 0 = argument error (RSA_E_*), > 0 = hipError_t from a launch.
 *
 * Activation storage ("split planes", the engine's internal HBM layout)
 *   A C-channel feature map of H x W pixels is stored channel-blocked by 8 ("NCHW8c"):
 *       hi[n][plane = c/8][y][x][c%8]   bf16   (round-to-nearest-even of the f32 value)
 *       lo[n][plane = c/8][y][x][c%8]   bf16   (bf16 of the rounding residual v - hi)
 *   One (plane,y,x) cell is a 16-byte "unit": 8 channels of one pixel = one MFMA
 *   k-group operand (v_mfma_f32_16x16x32_bf16 B fragment) and one coalesced 16 B lane load.
 *   Dense concatenation (reference torch.cat along dim 1) is a plane offset into a shared
 *   buffer, never a copy.  Residual streams are additionally kept in f32 as
 *       f32[n][plane4 = c/4][y][x][c%4]          ("NCHW4c")
 *   which is exactly the accumulator fragment of the MFMA (4 consecutive channels / lane).
 */
#ifndef RESSELT_AMD_H
#define RESSELT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSA_VERSION 400 /* 0.4.0: rsa_conv2d_pair (cross-layer fusion of residual dense block convolutions), fp16 saturation + overflow status */

/* error codes (negative = argument errors) */
#define RSA_OK 0
#define RSA_E_ARG (-1)       /* null / out-of-range argument */
#define RSA_E_UNSUPPORTED (-2) /* combination not compiled in */
#define RSA_E_ALIGN (-3)     /* pointer not 16-byte aligned */
#define RSA_E_FP16_RANGE (-5) /* rsa_check_status only: rsa_check_finite has seen an infinity or a NaN since the last call (an activation of a
                                one-product fp16 layer left the format's range, or the input was not finite): rerun with three bf16 products */
#define RSA_E_INTERNAL (-4)  /* a kernel reported a protocol failure (ring schedule hand-off timed out): results of the launches since the
                                last rsa_check_status() == RSA_OK are not to be trusted */

/* activation selector of the fused epilogue */
enum rsa_act {
  RSA_ACT_NONE = 0,
  RSA_ACT_LRELU = 1, /* act_param = negative slope; reference utilities/block.py:17-30 */
  RSA_ACT_MISH = 2,  /* reference archs/spanplus/arch.py:121 (nn.Mish)                  */
  RSA_ACT_SILU = 3,  /* reference archs/span/arch.py:164 (nn.SiLU)                      */
  RSA_ACT_GELU = 4,  /* erf GELU, reference archs/swinir/arch.py:34-40 (nn.GELU)        */
  RSA_ACT_SPAB_GATE = 5, /* y = (acc + res1) * (sigmoid(acc) - 0.5); spanplus/arch.py:126-127 */
  RSA_ACT_PRELU = 6      /* per-channel slopes act_vec[cout]; nn.PReLU(num_parameters=C), compact/arch.py:42-52 */
};

/* dtype of plain tensors crossing the boundary.  F32 / F16 / BF16: NCHW float tensors.  RSA_U8: an 8-bit IMAGE, channel-interleaved
 * [N][H][W][C] as image decoders deliver it (the reference leaves both conversions to its callers):
 *   read  (rsa_nchw_to_planes):           v = byte / 255   (a true division: bit-identical to torch's img.float() / 255)
 *   write (rsa_conv2d final store):       byte = round-half-even(clamp(v, 0, 1) * 255)   (torch: (y.clamp(0, 1) * 255).round()) */
enum rsa_dtype { RSA_F32 = 0, RSA_F16 = 1, RSA_BF16 = 2, RSA_U8 = 3 };

/* 16-bit element format of split planes and packed weights.  A plane buffer has ONE format; `hi` is the round-to-nearest-even of the f32
 * value in that format and `lo` (optional) the same rounding of the residual v - hi:
 *   RSA_PF_BF16: 8 + 8 significant bits with both halves, f32 range                       -> v_mfma_f32_16x16x32_bf16
 *   RSA_PF_F16 : 11 significant bits with hi alone, 22 with both, |v| < 65504             -> v_mfma_f32_16x16x32_f16
 * The one-product fp16 mode (products == 1, in_fmt == RSA_PF_F16) reads hi only: a third of the matrix instructions and half the
 * activation bytes of products == 3; the engine's `precision = 'auto'` uses it where the error budget allows. */
enum rsa_plane_fmt { RSA_PF_BF16 = 0, RSA_PF_F16 = 1 };

/*
 * One fused convolution launch.
 *
 * Replaces, in one kernel, the reference sequence
 *   [torch.cat of earlier outputs] -> [nn.Upsample(x2, nearest)] -> nn.Conv2d(k=1|3, s=1, zero pad k/2, bias)
 *   -> [activation] -> [* alpha + residual] -> [* beta + residual2] -> [nn.PixelShuffle]
 * i.e. utilities/block.py:148-200 (conv_block), :454-465 (ResidualDenseBlock_5C.forward),
 * :340-344 (RRDB.forward), :83-91 (ShortcutBlock), :510-537 (upconv_block), :477-507
 * (pixelshuffle_block); archs/spanplus/arch.py:94-130; archs/swinir/arch.py:34-40 (Linear = k1 conv).
 *
 * Arithmetic: implicit GEMM on v_mfma_f32_16x16x32_bf16 / _f16 (in_fmt), f32 accumulate.
 *   products == 1 : acc += hi(a)*hi(w)                                 ("bf16" / "fp16")
 *   products == 3 : acc += hi(a)*hi(w) + lo(a)*hi(w) + hi(a)*lo(w)     ("bf16x3": ~16-bit operands; "fp16x3": ~22-bit)
 */
typedef struct rsa_conv_params {
  /* geometry */
  int32_t batch;        /* N */
  int32_t H, W;         /* OUTPUT height/width in pixels (before pixel_shuffle) */
  int32_t ksize;        /* 1 or 3 */
  int32_t upsample2x;   /* 1: input map is (H/2 x W/2), nearest-upsampled on read */
  int32_t cin_planes;   /* input planes (of 8 channels) consumed, starting at in_hi/in_lo */
  int32_t cout;         /* real output channels */
  int32_t products;     /* 1 or 3 */

  /* input, split planes; strides in 16-byte units */
  const void* in_hi;
  const void* in_lo;        /* may be NULL when products == 1 */
  int64_t in_plane_stride;  /* units between planes  (= Hin*Win for a dense tensor) */
  int64_t in_batch_stride;  /* units between images  */

  /* weights in the packed layout described below (resselt_amd/engine/pack.py), bias f32[round_up(cout,16)] */
  const void* w_packed;
  const float* bias; /* may be NULL */

  /* epilogue */
  int32_t act;      /* enum rsa_act */
  float act_param;  /* LeakyReLU slope */
  float alpha;      /* used when res1 != NULL (or act == SPAB gate) */
  const float* res1; /* f32 NCHW4c [N][ceil(cout/4)][H][W][4] */
  float beta;
  const float* res2;

  /* outputs; each may be NULL */
  void* out_hi;             /* split planes, written at plane offset out_plane_off */
  void* out_lo;             /* NULL allowed (bf16 single-plane consumers) */
  int32_t out_plane_off;    /* first plane written (cout/8 planes follow; tail channels zeroed) */
  int64_t out_plane_stride; /* units */
  int64_t out_batch_stride; /* units */
  float* out_f32;           /* f32 NCHW4c residual stream */

  void* out_nchw;           /* final plain tensor [N][cout/r^2][H*r][W*r], dtype out_dtype -- or, for RSA_U8, the 8-bit image
                               [N][H*r][W*r][cout/r^2]; exclusive with out_hi/out_f32/res1/res2 (separate kernel instantiation) */
  int32_t out_dtype;        /* enum rsa_dtype */
  int32_t pixel_shuffle;    /* r >= 1 (depth-to-space factor applied while storing out_nchw; r > 1: out_nchw 16-byte aligned) */
  float out_scale;          /* out_nchw value = v * out_scale + out_shift[oc]  (SwinIR x/img_range + mean, */
  const float* out_shift;   /*   archs/swinir/arch.py:1013); NULL = none */
  const float* act_vec;     /* RSA_ACT_PRELU: negative slopes, f32[round_up(cout,16)], 16-byte aligned */
  const void* out_base;     /* optional with out_nchw: plain [N][cout/r^2][H][W] tensor of dtype out_dtype whose pixel (y,x) is
                               ADDED to all r x r output pixels it covers (nearest-upsampled base image, compact/arch.py:61-64) */
  int32_t out_base_div;     /* 0: the base image is H x W as described above.  > 0: the base image is out_base_h x out_base_w and output pixel
                               (Y, X) receives base pixel (min(Y / div, h-1), min(X / div, w-1)) -- F.interpolate(x, scale_factor=div) of the
                               UNPADDED input under a padded / unshuffled convolution grid (rtmosr/arch.py:383-387) */
  int32_t out_base_h, out_base_w;
  int32_t w_layout;         /* layout of w_packed: must equal rsa_conv_weight_layout(this descriptor); see rsa_pack_weights */
  /* Residual operands as SPLIT PLANES instead of f32 maps (value = hi + lo, ~16 bits): the residual stream of a residual dense block
   * is then stored once (the planes the next convolution reads) instead of twice.  res1_hi excludes res1, res2_hi excludes res2;
   * the planes start at the residual's channel 0; strides in 16-byte units, shared by both; lo pointers may be NULL (hi only). */
  const void* res1_hi;
  const void* res1_lo;
  const void* res2_hi;
  const void* res2_lo;
  int64_t res_plane_stride;
  int64_t res_batch_stride;
  /* enum rsa_plane_fmt of the three plane operands (0 = bf16, the default of a zeroed descriptor) */
  int32_t in_fmt;   /* in_hi / in_lo AND w_packed: selects the matrix instruction */
  int32_t out_fmt;  /* out_hi / out_lo */
  int32_t res_fmt;  /* res1_hi / res1_lo / res2_hi / res2_lo */
  int32_t tile_order; /* ring schedule only: 0 = output tiles in band order from the top of the map, 1 = the same order reversed (bottom first).
                         Alternating it between consecutive layers makes a layer start on the rows its producer wrote last, which are still in
                         the 256 MB Infinity Cache (a 1080p layer moves 0.4-0.9 GB); other schedules ignore it.  Any other value: RSA_E_ARG */
  /* Round 4: the lo halves of an fp16 residual stream as 8-bit codes.  A residual dense block's `x5 * 0.2 + x` (reference
   * utilities/block.py:463-465) carries its stream as hi (fp16) + lo; |lo| is at most half an ulp of hi = 2^12 f32 ulps, so a signed byte
   * codes it as the distance from f32(hi) to the value in steps of 32 f32 ulps, counted along the f32 bit patterns:
   *   code = min((sat16(bits(v) - bits(f32(hi))) + 16) >> 5, 127) & 0xff,   value = as_float(bits(f32(hi)) + (code << 5)).
   * hi + code keep 19 significant bits (fp16 hi + fp16 lo: 22) and the stream is 3 bytes per channel instead of 4; with hi = 0 or subnormal
   * any code decodes to within 2^-24 of the value, a non-finite hi keeps the stream's hi non-finite.  An lo8 plane holds 8-byte units
   * [n][plane][y][x][8 codes]: the plane stride is that of the hi planes (in units), the batch stride is lo8_batch_stride (8-byte units).
   * Bits of lo8_flags: RSA_LO8_RES1 (res1_lo), RSA_LO8_RES2 (res2_lo), RSA_LO8_OUT (out_lo); only with fp16 planes (res_fmt / out_fmt =
   * RSA_PF_F16) and plane residuals / outputs that have hi + lo. */
  int32_t lo8_flags;
  int32_t reserved_lo8;     /* must be 0 */
  int64_t lo8_batch_stride; /* 8-byte units between images of an lo8 buffer (shared by the flagged operands) */
} rsa_conv_params;
#define RSA_LO8_RES1 1
#define RSA_LO8_RES2 2
#define RSA_LO8_OUT 4

/* Launch `n` fused convolutions in order on `stream` (one host call per forward pass).  Both return RSA_E_INTERNAL, without launching,
 * when a kernel of an EARLIER call has reported a protocol failure that rsa_check_status has not yet been asked about. */
int rsa_conv2d(const rsa_conv_params* p, void* stream);
int rsa_conv2d_list(const rsa_conv_params* list, int32_t n, void* stream);

/* Cross-layer fusion of a residual dense block (SURVEY.md 8b `sr_rdb_fused`; reference utilities/block.py:454-465): descriptors `a` and `b`
 * are a FUSABLE PAIR when b is the growth convolution that follows a in the block -- both 3x3, one fp16 product, 32 output channels, bias +
 * LeakyReLU / linear into hi-only fp16 planes, a writing the four planes right behind its own input planes and b reading a's input planes
 * plus those four (conv1 -> conv2, conv3 -> conv4).  rsa_conv2d_pair runs both in one launch that streams the common input through LDS once
 * (b's last 32 input channels never leave the chip before b has used them; a's output is still written for the later layers); the result
 * is bit-identical to rsa_conv2d(a) followed by rsa_conv2d(b).  rsa_conv2d_list fuses such neighbours by itself unless RSA_CONV_PAIR=0
 * is set in the environment.  rsa_conv_pair_fusable: 1 when the list would fuse (a, b), else 0.  rsa_conv2d_pair on anything else:
 * RSA_E_UNSUPPORTED.  Both descriptors carry their own packed weights (layout 1), exactly as for separate launches. */
int rsa_conv2d_pair(const rsa_conv_params* a, const rsa_conv_params* b, void* stream);
int rsa_conv_pair_fusable(const rsa_conv_params* a, const rsa_conv_params* b);

/* Failure word of the ring schedule (csrc/conv_ring.h: a hand-off between the loader wave and the compute waves that timed out makes the
 * kernel drain with wrong pixels).  The kernels report into host-visible memory, so this call never synchronises: it sees the failures of
 * every launch that has COMPLETED.  Call it after the stream (or an event behind the forward) has been synchronised to judge that forward.
 * Returns RSA_OK, or RSA_E_INTERNAL once (the word is cleared; rsa_last_error_string says how many hand-offs failed). */
int rsa_check_status(void);

/* Range guard of the fp16 plane format.  The fp16 epilogues convert with v_cvt_pk_f16_f32, which turns a value beyond +-65504 into an
 * infinity; the infinity (or the NaN that inf - inf / inf * 0 make of it) then travels with the residual stream of the network to its end --
 * the reference's `x5 * 0.2 + x` (utilities/block.py:463-465) carries it -- where ONE pass over a small tensor finds it.  This call scans
 * `count` elements of a plain array (`dtype`: RSA_F32 / RSA_F16 / RSA_BF16; a split-plane buffer is such an array of its 16-bit format) on
 * `stream` and adds to a host-visible word when it meets a non-finite value; it never synchronises.  rsa_check_status() then returns
 * RSA_E_FP16_RANGE once (after RSA_E_INTERNAL, which has priority).  The engine runs it behind every forward of a model whose precision
 * policy has fp16 layers (0.06 % of an RRDBNet 1080p frame) and `precision = 'auto'` answers a hit by re-running in three bf16 products.
 * (A saturating convert was rejected: it would turn an out-of-range activation into a finite wrong value that no later check can see.) */
int rsa_check_finite(const void* data, int32_t dtype, int64_t count, void* stream);

/* Bytes of the packed weight blob for a (cout, cin_planes, ksize, products) convolution. */
int64_t rsa_packed_weight_bytes(int32_t cout, int32_t cin_planes, int32_t ksize, int32_t products);
/* the same for a given layout (layouts 0..2 have the size above; layout 3 is 32 K steps of 4 cout tiles = 256 KiB) */
int64_t rsa_packed_weight_bytes_layout(int32_t cout, int32_t cin_planes, int32_t ksize, int32_t products, int32_t layout);

/* Number of 16-channel cout tiles one workgroup computes for `cout` output channels (1..4); the grid has
 * ceil(ceil(cout/16) / tiles) slabs in y.  Exposed so host code and tests can reason about launch geometry. */
int rsa_conv_cout_tiles(int32_t cout);

/*
 * Weight packing: OIHW f32 weights (device pointer, contiguous [cout][cin][k][k]) -> the MFMA A-fragment blob of
 * rsa_packed_weight_bytes(cout, cin_planes, ksize, products) bytes that rsa_conv2d streams.  Replaces what
 * nn.Module.load_state_dict does with an nn.Conv2d / nn.Linear weight (reference registry.py:113).  Two K orders exist and the
 * schedule a descriptor dispatches to fixes which one it reads; ask with rsa_conv_weight_layout(descriptor) (every field except
 * w_packed / w_layout filled in) and pass the answer as `layout` here and as rsa_conv_params.w_layout:
 *   0  blob[chunk q][tap t][cout_tile][hi|lo][lane 0..63][8] bf16, lane l: cout = 16*tile + (l & 15), cin = 32*q + 8*(l >> 4) + j
 *   1  tap-pair order of the ring schedule (3x3, three products, whole 32-channel chunks): resselt_amd/csrc/pack.hip
 *   2  the same per 16-channel half chunk, five K steps each (an odd number of half chunks, e.g. 48 input channels)
 *   3  nearest x2 upsampling + 3x3 as four 2x2 phase convolutions on the source map, taps pre-summed (64 -> 64 channels;
 *      resselt_amd/csrc/conv_ring_up.h)
 * (hi = RNE of w in `fmt` (enum rsa_plane_fmt), lo = the same rounding of w - hi; only hi when products == 1).
 */
int rsa_conv_weight_layout(const rsa_conv_params* p);
int rsa_pack_weights(const float* w_oihw, int32_t cout, int32_t cin, int32_t cin_planes, int32_t ksize, int32_t products, int32_t layout,
                     int32_t fmt, void* out, void* stream);

/* Name of the kernel a descriptor dispatches to (matches the rocprofv3 kernel names; bench.py groups its rooflines by it). */
const char* rsa_conv_kernel_name(const rsa_conv_params* p);

/* Debug: hand-offs of the ring schedule that ran into their spin bound since the last call (always 0 in a correct build; tests assert it).
 * Synchronises the device (unlike rsa_check_status, which is the product's way to learn of the same event). */
int rsa_debug_ring_aborts(void);
/* Debug: polls a hand-off of the ring schedule may spend before it gives up (default 2^18; 1 forces the failure path: tests). */
int rsa_debug_set_ring_spin_limit(int32_t polls);
/* Debug: force the ring schedule on (1) / off (0) for descriptors built afterwards, or follow RSA_CONV_RING again (-1).  Descriptors carry
 * the layout they were built for, so change it only between building descriptor sets (in-process A/B timing). */
int rsa_debug_set_ring(int32_t mode);
/* Debug: pair fusion inside rsa_conv2d_list on (1) / off (0), or follow RSA_CONV_PAIR again (-1): in-process A/B timing. */
int rsa_debug_set_pair(int32_t mode);

/*
 * Plain NCHW tensor [N][C][src_h][src_w] -> split planes of size H x W, with per-channel affine v = (x - mean[c]) * scale.
 * Replaces the implicit NCHW read of the first conv, `(x - mean) * img_range` (archs/span/arch.py:232-234,
 * archs/swinir/arch.py:966-967) and, when H > src_h or W > src_w, the right/bottom REFLECT padding of
 * pad_to_multiple (utilities/padding.py:24-29; SwinIR.check_image_size).  Channels padded to 8 with zeros.
 * unshuffle = r > 1 additionally fuses torch.pixel_unshuffle(x, r) (RRDBNet x2plus/x1 front end, archs/esrgan/arch.py:130-137):
 * the planes then hold C*r*r channels of an (H x W) grid covering H*r x W*r source pixels.
 */
int rsa_nchw_to_planes(const void* x, int32_t dtype, int32_t batch, int32_t C, int32_t H, int32_t W, int32_t src_h, int32_t src_w,
                       int32_t unshuffle, const float* mean, float scale, void* out_hi, void* out_lo, int64_t out_plane_stride,
                       int64_t out_batch_stride, int32_t out_fmt, void* stream);

/* split planes / f32 NCHW4c -> plain NCHW (debug + parity of intermediates) */
int rsa_planes_to_nchw(const void* hi, const void* lo, int64_t plane_stride, int64_t batch_stride, int32_t batch,
                       int32_t C, int32_t H, int32_t W, int32_t fmt, float* out, void* stream);

/*
 * DySample upsampler head: sigmoid-gated learned offsets -> bilinear border gather over channel groups -> 1x1 conv.
 * Replaces resselt/utilities/dysample.py:47-83 after the two 1x1 offset/scope convs (run as ONE rsa_conv2d whose
 * f32 NCHW4c output holds offset channels [0, oc) and scope channels [oc, 2*oc), oc = 2*groups*scale^2).
 */
typedef struct rsa_dysample_params {
  int32_t batch;
  int32_t H, W;          /* low-resolution size */
  int32_t C;             /* feature channels, multiple of 4*groups */
  int32_t groups;        /* 4 in the reference */
  int32_t scale;
  int32_t out_ch;        /* 1..8 */
  const float* x_f32;    /* features, f32 NCHW4c [N][C/4][H][W][4] */
  const float* offscope; /* f32 NCHW4c [N][2*oc/4][H][W][4] */
  const float* init_pos; /* [oc] (registered buffer of the reference module, dysample.py:43-45) */
  const float* end_w;    /* [out_ch][C]; NULL = x_f32 is PRE-PROJECTED: C == 4*groups, channel 4g+o = sum over the channels c of group g of
                            W_end[o][c] * x[c] (the 1x1 end conv applied per group at low resolution; sampling is linear), out_ch <= 4 */
  const float* end_b;    /* [out_ch] or NULL */
  void* out_nchw;        /* [N][out_ch][H*scale][W*scale] */
  int32_t out_dtype;     /* enum rsa_dtype */
} rsa_dysample_params;

int rsa_dysample(const rsa_dysample_params* p, void* stream);

/*
 * nn.LayerNorm over the channel axis of a token map (tokens = pixels).
 * Replaces norm1 / norm2 / patch_embed.norm / norm of resselt/archs/swinir/arch.py:306,333,640,959 together with the
 * flatten/transpose/view round trips of PatchEmbed / PatchUnEmbed (:638-642,679-682): the map never changes layout.
 */
typedef struct rsa_layernorm_params {
  int32_t batch;
  int32_t H, W;
  int32_t C;               /* channels normalised over */
  float eps;
  const float* x_f32;      /* f32 NCHW4c [N][ceil(C/4)][H][W][4] */
  const float* gamma;      /* [C] */
  const float* beta;       /* [C] */
  void* out_hi;            /* split planes [N][ceil(C/8)][H][W][8], tail channels zeroed; may be NULL */
  void* out_lo;            /* may be NULL */
  int64_t out_plane_stride; /* 16-byte units */
  int64_t out_batch_stride;
  float* out_f32;          /* optional f32 NCHW4c copy of the result */
  int32_t out_fmt;         /* enum rsa_plane_fmt of out_hi / out_lo */
  int32_t reserved0;       /* must be 0 */
} rsa_layernorm_params;

int rsa_layernorm(const rsa_layernorm_params* p, void* stream);

/*
 * (Shifted-)window multi-head self-attention core: softmax(q k^T + B[idx] (+ mask)) v for every window and head.
 * Replaces, between the qkv and proj Linear layers, resselt/archs/swinir/arch.py:141-170 (WindowAttention.forward) and the
 * torch.roll / window_partition / window_reverse / calculate_mask data movement of SwinTransformerBlock.forward (:295-335).
 * Input planes: [(which*heads + head)*4, +4) hold 32 channels (head_dim zero-padded) of q (pre-scaled), k, v.
 * bias_frag: relative_position_bias_table gathered by relative_position_index and laid out in accumulator-fragment order
 *            [head][query tile 2][key tile 2][lane 64][16] f32 (resselt_amd/archs/swinir/arch.py::bias_fragments);
 *            key slots beyond window^2 carry -1e30.
 */
typedef struct rsa_window_attn_params {
  int32_t batch;
  int32_t H, W;            /* multiples of `window` */
  int32_t heads;
  int32_t window;          /* <= 8 */
  int32_t shift;           /* 0 or window/2 */
  int32_t products;        /* 1 or 3 */
  const void* qkv_hi;
  const void* qkv_lo;      /* may be NULL when products == 1 */
  int64_t qkv_plane_stride; /* 16-byte units */
  int64_t qkv_batch_stride;
  const float* bias_frag;
  void* out_hi;            /* planes [head*4, +4) : attention output, head_dim padded to 32 */
  void* out_lo;
  int64_t out_plane_stride;
  int64_t out_batch_stride;
} rsa_window_attn_params;

int rsa_window_attention(const rsa_window_attn_params* p, void* stream);

/* ------------------------------------------------------------------------------------------- fused Swin block halves
 * The two halves of SwinTransformerBlock.forward (resselt/archs/swinir/arch.py:295-335), each as ONE launch that reads the f32
 * residual stream once and writes it once; everything between (LayerNorm output, q / k / v, attention output, the MLP's hidden map)
 * stays in LDS and registers.  One workgroup = 64 tokens (a window, or 64 consecutive tokens for the MLP).
 *
 * rsa_swin_attn_block:  out = x + proj(window_attention(qkv(norm1(x))))      (:295-330 with WindowAttention.forward :133-173,
 *                       torch.roll / window_partition / window_reverse / calculate_mask as index arithmetic)
 * rsa_swin_mlp_block :  out = x + fc2(GELU(fc1(norm2(x))))                   (:331-335 with Mlp.forward :34-40)
 *
 * Weights are rsa_pack_weights blobs in layout 0 (ksize 1): wqkv over rows regrouped per head (row (which, head, d) with head_dim
 * zero-padded to 32 and the q rows pre-scaled; cin_planes = ceil(C/8)), wproj over columns padded the same way (cin_planes =
 * 4*heads), w1 [hidden][C] (cin_planes = ceil(C/8)), w2 [C][hidden] (cin_planes = ceil(hidden/8)).  Biases are f32 vectors padded
 * with zeros to a multiple of 16.  bias_frag16: relative_position_bias_table[relative_position_index] in the accumulator order of
 * 16x16 tiles, [head][key tile 4][query tile 4][lane 64][4] f32: lane l, element r <-> key 16*kt + 4*(l >> 4) + r, query
 * 16*qt + (l & 15), every value multiplied by log2(e) (the kernel's softmax runs in base 2); key slots beyond window^2 carry -1e30.  Limits: C <= 256 (a multiple of 4), heads <= 8, head_dim <= 32,
 * window <= 8, hidden <= 512; `out` may be `x` (in place). */
typedef struct rsa_swin_attn_block_params {
  int32_t batch;
  int32_t H, W;            /* multiples of `window` */
  int32_t C;               /* embedding width */
  int32_t heads;
  int32_t window;          /* <= 8 */
  int32_t shift;           /* 0 or window/2 */
  int32_t products;        /* 1 or 3 */
  float eps;               /* LayerNorm epsilon */
  const float* x;          /* f32 NCHW4c [N][ceil(C/4)][H][W][4] */
  const float* gamma;      /* norm1 weight / bias, [C] */
  const float* beta;
  const void* wqkv;        /* packed, cout = 3*heads*32 */
  const float* bqkv;       /* [3*heads*32] */
  const float* bias_frag16;
  const void* wproj;       /* packed, cout = C, cin_planes = 4*heads */
  const float* bproj;      /* [C] padded to 16 */
  float* out;              /* f32 NCHW4c, same shape as x */
} rsa_swin_attn_block_params;

int rsa_swin_attn_block(const rsa_swin_attn_block_params* p, void* stream);

typedef struct rsa_swin_mlp_block_params {
  int32_t batch;
  int32_t H, W;
  int32_t C;
  int32_t hidden;
  int32_t products;        /* 1 or 3 */
  float eps;
  const float* x;          /* f32 NCHW4c */
  const float* gamma;      /* norm2 weight / bias, [C] */
  const float* beta;
  const void* w1;          /* packed, cout = hidden, cin_planes = ceil(C/8) */
  const float* b1;         /* [hidden] padded to 16 */
  const void* w2;          /* packed, cout = C, cin_planes = ceil(hidden/8) */
  const float* b2;         /* [C] padded to 16 */
  float* out;              /* f32 NCHW4c */
  void* out_hi;            /* optional split-plane copy of the result (the input of the convolution that follows a block group) */
  void* out_lo;
  int64_t out_plane_stride; /* 16-byte units */
  int64_t out_batch_stride;
  int32_t fmt;             /* enum rsa_plane_fmt of w1 / w2 and of the kernel's internal images (selects the matrix instruction): RSA_PF_F16 with
                              products == 1 is the one-product fp16 form; out_hi / out_lo are bf16 planes in every form */
  int32_t reserved0;       /* must be 0 */
} rsa_swin_mlp_block_params;

int rsa_swin_mlp_block(const rsa_swin_mlp_block_params* p, void* stream);

/* rsa_swin_block: both halves in one launch -- out = x1 + fc2(GELU(fc1(norm2(x1)))), x1 = x + proj(window_attention(qkv(norm1(x))))
 * (SwinTransformerBlock.forward, resselt/archs/swinir/arch.py:295-335).  x1 stays in registers: the residual stream is read once and
 * written once per block.  Operands and limits as for the two half launches above (resselt_amd/csrc/swin_block_full.hip). */
typedef struct rsa_swin_block_params {
  int32_t batch;
  int32_t H, W;            /* multiples of `window` */
  int32_t C;
  int32_t heads;
  int32_t window;          /* <= 8 */
  int32_t shift;           /* 0 or window/2 */
  int32_t hidden;
  int32_t products;        /* 1 or 3 */
  float eps;
  const float* x;          /* f32 NCHW4c */
  const float* gamma1;     /* norm1 */
  const float* beta1;
  const void* wqkv;
  const float* bqkv;
  const float* bias_frag16;
  const void* wproj;
  const float* bproj;
  const float* gamma2;     /* norm2 */
  const float* beta2;
  const void* w1;
  const float* b1;
  const void* w2;
  const float* b2;
  float* out;              /* f32 NCHW4c; may be x */
  void* out_hi;            /* optional split-plane copy of the result */
  void* out_lo;
  int64_t out_plane_stride; /* 16-byte units */
  int64_t out_batch_stride;
  int32_t fmt;             /* enum rsa_plane_fmt of the packed weights, of the on-chip operand images and of out_hi / out_lo.  fp16 is compiled
                              for products == 1: that instantiation keeps no lo images, its LDS array is 64 KB and two windows share a CU */
  int32_t reserved0;       /* must be 0 */
} rsa_swin_block_params;

int rsa_swin_block(const rsa_swin_block_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------------------------- DAT ops
 * Building blocks of the Dual Aggregation Transformer path (reference archs/dat/arch.py).  Tokens are pixels; every map is in
 * the split-plane layout [N][planes][H][W][8] (bf16 hi, optional lo).  Attention maps use the head-padded channel layout of
 * rsa_window_attention: head h owns channels [32h, 32h+32), head_dim <= 32, pad channels are zero. */

/* Rectangular (shifted) window attention core of Spatial_Attention.forward (arch.py:224-267) together with the zero padding,
 * torch.roll, img2windows / windows2img and calculate_mask of Adaptive_Spatial_Attention.forward (arch.py:336-411, 446-492).
 * One call = one branch (one window orientation) over `heads` consecutive head slots starting at `head0`. */
typedef struct rsa_rect_attn_params {
  int32_t batch;
  int32_t H, W;             /* token map size; tokens outside it (padding up to Hp x Wp) have q = k = v = 0 */
  int32_t Hp, Wp;           /* padded grid: multiples of win_h / win_w, >= H / W */
  int32_t win_h, win_w;     /* win_h * win_w <= 256 */
  int32_t shift_h, shift_w; /* 0 (no mask) or the cyclic shift; 0 <= shift < win */
  int32_t heads;            /* head slots this call processes */
  int32_t head0;            /* first of them */
  int32_t heads_total;      /* head slots per q / k / v group: q planes [0, 4*heads_total), then k, then v */
  int32_t products;         /* 1 or 3 */
  const void* qkv_hi;
  const void* qkv_lo;       /* may be NULL when products == 1 */
  int64_t qkv_plane_stride; /* 16-byte units */
  int64_t qkv_batch_stride;
  const float* bias_frag;   /* [heads][T][T][64][16] f32, T = tiles of 32 tokens (1, 2, 4 or 8): S^T accumulator order, -1e30 on padded keys */
  void* out_hi;             /* planes [(head0 + h)*4, +4) */
  void* out_lo;
  int64_t out_plane_stride;
  int64_t out_batch_stride;
  /* Cross-window mode (0 = off): keys / values come from the kwin_h x kwin_w window that starts kpad pixels up-left of the query
   * window, zeros outside the map -- OCAB of HAT (reference archs/hat/arch.py:403-470: nn.Unfold with padding).  bias_frag is then
   * [heads][QT][KT][64][16] with QT = ceil(win_h*win_w / 32), KT = ceil(kwin_h*kwin_w / 32).  No shift, Hp == H, Wp == W. */
  int32_t kwin_h, kwin_w;
  int32_t kpad_h, kpad_w;
  /* Wide heads: a head slot is head_chunks x 4 planes (head_dim <= 32*head_chunks, zero-padded), 0 or 1 = the 32-channel slots above;
   * 2..4 (self-attention only): DRCT's dense groups run heads of 46..122 channels (reference archs/drct/arch.py:204-329).  q planes
   * [0, 4*head_chunks*heads_total), then k, then v; out planes [(head0 + h)*4*head_chunks, +4*head_chunks). */
  int32_t head_chunks;
  int32_t fmt;              /* enum rsa_plane_fmt of the q / k / v planes AND of the output planes (0 = bf16).  RSA_PF_F16 with products == 1: the
                               one-product fp16 form (v_mfma_f32_32x32x16_f16; probabilities rounded to fp16) */
  int32_t reserved0;        /* must be 0 */
} rsa_rect_attn_params;
int rsa_rect_attention(const rsa_rect_attn_params* p, void* stream);

/* Channel ("transposed") attention weights of Adaptive_Channel_Attention.forward (arch.py:577-585):
 *   attn[b][h] = softmax_j( normalize(q)[i] . normalize(k)[j] * temperature[h] )   over ALL tokens of image b,
 * written as the packed bf16 hi/lo weights of a block-diagonal 1x1 convolution (rsa_conv2d, cin = cout = 32*heads) so that
 * `attn @ v` is one more launch of the convolution kernel.  Two deterministic stages (per-chunk partial Gram matrices in
 * `workspace`, then one workgroup per (image, head)); no atomics. */
typedef struct rsa_channel_attn_params {
  int32_t batch;
  int32_t H, W;
  int32_t heads;
  int32_t head_dim;          /* <= 32 */
  int32_t products;          /* layout of w_packed: 3 = hi+lo, 1 = hi only */
  const void* q_hi;          /* planes [4h, 4h+4) = head h */
  const void* q_lo;          /* may be NULL */
  const void* k_hi;
  const void* k_lo;
  int64_t plane_stride;      /* 16-byte units (same for q and k) */
  int64_t batch_stride;
  const float* temperature;  /* [heads] */
  float* workspace;          /* >= rsa_channel_attn_workspace_bytes() */
  void* w_packed;            /* [batch] blobs of rsa_packed_weight_bytes(32*heads, 4*heads, 1, products); off-diagonal blocks must be zero */
  int32_t fmt;               /* enum rsa_plane_fmt of q / k AND of the packed weights written (0 = bf16, the default of a zeroed descriptor; round 4: fp16 planes) */
  int32_t reserved1;         /* must be 0 */
} rsa_channel_attn_params;
int64_t rsa_channel_attn_workspace_bytes(int32_t batch, int32_t H, int32_t W, int32_t heads);
int rsa_channel_attention_weights(const rsa_channel_attn_params* p, void* stream);

/* Depthwise 3x3 convolution, zero padding 1 (arch.py:52, 322, 540):  out = act(dw(x') + bias) [* mul]
 * with x' = x, or x' = (x - mean_p) * rstd_p * gamma_c + beta_c when `stats` is given (the LayerNorm of SpatialGate, arch.py:55-59,
 * applied on the fly from per-pixel statistics; padding stays zero AFTER the normalisation).  BatchNorm(eval) is folded by the host. */
typedef struct rsa_dwconv_params {
  int32_t batch;
  int32_t H, W;
  int32_t planes;            /* 8 channels each */
  int32_t act;               /* RSA_ACT_NONE or RSA_ACT_GELU */
  const void* in_hi;
  const void* in_lo;         /* may be NULL */
  int64_t in_plane_stride;
  int64_t in_batch_stride;
  const float* weight;       /* [planes*8][9] */
  const float* bias;         /* [planes*8] */
  const float* stats;        /* optional [batch][H*W][2] = (mean, rstd) */
  const float* gamma;        /* [planes*8], with stats */
  const float* beta;
  const void* mul_hi;        /* optional elementwise multiplier map */
  const void* mul_lo;
  int64_t mul_plane_stride;
  int64_t mul_batch_stride;
  void* out_hi;
  void* out_lo;              /* may be NULL */
  int64_t out_plane_stride;
  int64_t out_batch_stride;
  int32_t fmt;               /* enum rsa_plane_fmt of every plane operand of this call (0 = bf16, the default of a zeroed descriptor; round 4: fp16 planes) */
  int32_t reserved1;         /* must be 0 */
} rsa_dwconv_params;
int rsa_dwconv3x3(const rsa_dwconv_params* p, void* stream);
/* Depthwise 5x5, zero padding 2, weight [planes*8][25]: OmniShift of RTMoSR re-parameterised to one kernel (archs/rtmosr/arch.py:253-289).
 * act must be RSA_ACT_NONE and stats NULL; the optional multiplier map is supported. */
int rsa_dwconv5x5(const rsa_dwconv_params* p, void* stream);

/* Per-pixel LayerNorm statistics over channels [0, C) of a plane range: stats[b][pixel] = (mean, 1/sqrt(var + eps)). */
int rsa_plane_stats(const void* in_hi, const void* in_lo, int64_t plane_stride, int64_t batch_stride, int32_t batch, int32_t H, int32_t W,
                    int32_t C, float eps, float* stats, void* stream);
/* the same over planes of format `fmt` (enum rsa_plane_fmt; rsa_plane_stats = bf16 planes) */
int rsa_plane_stats_fmt(const void* in_hi, const void* in_lo, int64_t plane_stride, int64_t batch_stride, int32_t batch, int32_t H, int32_t W,
                        int32_t C, float eps, int32_t fmt, float* stats, void* stream);

/* channel_interaction of the AIM (arch.py:326-332): gate[b][c] = sigmoid( W2 . gelu(W1 . mean_pixels(x[b]) + b1) + b2 ).
 * BatchNorm(eval) folded into W1/b1 by the host.  Deterministic two-stage mean; `workspace` >= rsa_channel_gate_workspace_bytes(). */
typedef struct rsa_channel_gate_params {
  int32_t batch;
  int32_t H, W;
  int32_t planes;            /* C = 8*planes (pad channels carry zero weights) */
  int32_t hidden;            /* <= 128 */
  const void* in_hi;
  const void* in_lo;         /* may be NULL */
  int64_t in_plane_stride;
  int64_t in_batch_stride;
  const float* w1;           /* [hidden][C] */
  const float* b1;           /* [hidden] */
  const float* w2;           /* [C][hidden] */
  const float* b2;           /* [C] */
  float* workspace;
  float* gate;               /* [batch][C] */
  int32_t relu;              /* 0 = GELU hidden, sigmoid gate (DAT); 1 = ReLU, sigmoid (the RCAN-style channel attention of HAT's CAB,
                                archs/hat/arch.py:28-35); 2 = ReLU, Hardsigmoid (RTMoSR's CSELayer, archs/rtmosr/arch.py:7-22) */
  int32_t fmt;               /* enum rsa_plane_fmt of every plane operand of this call (0 = bf16, the default of a zeroed descriptor; round 4: fp16 planes) */
} rsa_channel_gate_params;
int64_t rsa_channel_gate_workspace_bytes(int32_t batch, int32_t H, int32_t W, int32_t planes);
int rsa_channel_gate(const rsa_channel_gate_params* p, void* stream);

/* Adaptive Interaction Module combine (arch.py:494-508 and 595-607):  s = w2 . gelu(W1 . src[p] + b1) + b2  (spatial_interaction, BN folded)
 *   mode 0 (spatial block): src = att,  out = att * gate[c] + sigmoid(s) * conv
 *   mode 1 (channel block): src = conv, out = att * sigmoid(s) + conv * gate[c] */
typedef struct rsa_aim_params {
  int32_t batch;
  int32_t H, W;
  int32_t planes;
  int32_t hidden;            /* <= 16 */
  int32_t mode;
  const void* att_hi;
  const void* att_lo;
  int64_t att_plane_stride;
  int64_t att_batch_stride;
  const void* conv_hi;
  const void* conv_lo;
  int64_t conv_plane_stride;
  int64_t conv_batch_stride;
  const float* gate;         /* [batch][8*planes] */
  const float* w1;           /* [hidden][8*planes] */
  const float* b1;           /* [hidden] */
  const float* w2;           /* [hidden] */
  float b2;
  void* out_hi;
  void* out_lo;
  int64_t out_plane_stride;
  int64_t out_batch_stride;
  int32_t fmt;               /* enum rsa_plane_fmt of every plane operand of this call (0 = bf16, the default of a zeroed descriptor; round 4: fp16 planes) */
  int32_t reserved1;         /* must be 0 */
} rsa_aim_params;
int rsa_aim_combine(const rsa_aim_params* p, void* stream);

/* out = base + x * gate[b][c] * scale on f32 maps [N][ceil(C/4)][H][W][4], x in split planes; gate rows have 8*ceil(C/8) entries.
 * HAT's `shortcut + conv_x * conv_scale` with the CAB's channel attention as the gate (reference archs/hat/arch.py:37-39, 345). */
int rsa_gated_add(const void* x_hi, const void* x_lo, int64_t plane_stride, int64_t batch_stride, int32_t batch, int32_t H, int32_t W, int32_t C,
                  const float* gate, float scale, const float* base_f32, float* out_f32, void* stream);

/* ---------------------------------------------------------------------------------------------------------------- RTMoSR ops
 * (reference archs/rtmosr/arch.py; every RepConv / OmniShift is re-parameterised to one kernel by the host) */

/* RMSNorm over channels (arch.py:32-37): out = scale[c] * x / (||x||_2 / sqrt(C) + eps) + offset[c];  f32 map in, split planes out. */
int rsa_rmsnorm(const float* x_f32, int32_t batch, int32_t H, int32_t W, int32_t C, float eps, const float* scale, const float* offset, void* out_hi,
                void* out_lo, int64_t out_plane_stride, int64_t out_batch_stride, void* stream);

/* ParPixelUnshuffle's two reads of its input (arch.py:292-299), `planes` planes of H x W (both even):
 *   unshuffled_f32: PixelUnshuffle(2) as an f32 map [N][planes*8][H/2][W/2][4] (f32 group c holds the 2x2 block of channel c) -- the
 *                   residual operand of the RepConv that follows;  pool: MaxPool2d(2) as split planes [N][planes][H/2][W/2][8]. */
int rsa_unshuffle_pool(const void* in_hi, const void* in_lo, int64_t in_plane_stride, int64_t in_batch_stride, int32_t batch, int32_t H, int32_t W,
                       int32_t planes, float* unshuffled_f32, void* pool_hi, void* pool_lo, int64_t pool_plane_stride, int64_t pool_batch_stride,
                       void* stream);

/* The gate of GatedCNNBlock.forward (arch.py:334-336):  out = mish(g) * cat(i, PixelShuffle(2)(c * gate)).
 * g = planes [0, g_planes) and i = planes [g_planes, g_planes + i_planes) of the fc1 output `f` (H x W); c = (g_planes - i_planes)*4
 * planes at H/2 x W/2 (the OmniShift output); gate = the SE layer's per-channel factors of c, or NULL. */
typedef struct rsa_gated_shuffle_params {
  int32_t batch;
  int32_t H, W;              /* both even */
  int32_t g_planes;          /* planes of g = planes of the output */
  int32_t i_planes;          /* planes of i */
  const void* f_hi;
  const void* f_lo;          /* may be NULL */
  int64_t f_plane_stride;
  int64_t f_batch_stride;
  const void* c_hi;
  const void* c_lo;
  int64_t c_plane_stride;
  int64_t c_batch_stride;
  const float* gate;         /* [batch][gate_stride] or NULL */
  int64_t gate_stride;
  void* out_hi;
  void* out_lo;
  int64_t out_plane_stride;
  int64_t out_batch_stride;
} rsa_gated_shuffle_params;
int rsa_gated_shuffle_mul(const rsa_gated_shuffle_params* p, void* stream);

/* 8-bit images either side of the path (the reference leaves both steps to its callers):
 *   rsa_image_u8_to_nchw   uint8 [N][H][W][C] (interleaved, as image decoders deliver it) -> float [N][C][H][W], v / 255
 *   rsa_nchw_to_image_u8   float [N][C][H][W] -> uint8 [N][H][W][C], round-half-even(clamp(v, 0, 1) * 255)  (torch: (y.clamp(0,1)*255).round())
 * dtype = rsa_dtype of the float tensor. */
int rsa_image_u8_to_nchw(const uint8_t* img, int32_t batch, int32_t H, int32_t W, int32_t C, void* out, int32_t dtype, void* stream);
int rsa_nchw_to_image_u8(const void* x, int32_t dtype, int32_t batch, int32_t C, int32_t H, int32_t W, uint8_t* img, void* stream);

/* version / errors */
int rsa_version(void);
const char* rsa_last_error_string(void);

#ifdef __cplusplus
}
#endif
#endif /* RESSELT_AMD_H */
