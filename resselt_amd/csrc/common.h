// common.h — shared host-side helpers of the C-ABI library (error string, dispatch prototypes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "resselt_amd.h"

namespace rsa {

// Records a message for rsa_last_error_string() (thread-local) and returns `code`.
int set_error(int code, const char* msg);

// conv_mfma.hip
int conv_launch(const rsa_conv_params& p, hipStream_t stream);
int conv_validate(const rsa_conv_params& p);  // every argument check of conv_launch, nothing launched
int conv_nct(int cout);
const char* conv_kernel_name(const rsa_conv_params& p);

// Packed-weight layouts (rsa_pack_weights): 0 = tap-major chunks (conv_kernel.h, gemm_k1.hip), 1 = tap-pair order (conv_ring.h)
enum { RSA_WL_TAPS = 0, RSA_WL_PAIRS = 1, RSA_WL_HALFPAIRS = 2, RSA_WL_UPPHASE = 3 };  // 2: the ring schedule's half mode (odd number of half chunks)
                                                                                       // 3: four 2x2 phase kernels of an upsampling layer (conv_ring_up.h)

// A descriptor takes the ring schedule (conv_ring.h) iff: 3x3, three products, split-plane / f32-map outputs (final NCHW stores: the
// three-tile shape only) and either whole
// 32-channel input chunks with two, three or four cout tiles (three: no fused upsampling), or an odd number of 16-channel half chunks
// with three cout tiles (the 48 -> 48 layers of the SPAN family: half mode).  RSA_CONV_RING=0 in the environment switches the schedule off (A/B runs).
bool conv_ring_enabled();
void conv_ring_override(int v);
// Round 3: the same schedule in ONE product on fp16 hi planes and in three fp16 products on fp16 split planes (every shape, no fused
// upsampling): the residual dense blocks / the trunk convolution of RRDBNet and the SPAN family under their 'auto' precision policies.
inline bool conv_ring_eligible(const rsa_conv_params& p) {
  const int ct = (p.cout + 15) / 16;
  if (p.ksize != 3 || p.cin_planes < 2 || (p.cin_planes & 1)) return false;
  const bool whole = (p.cin_planes & 3) == 0;
  if (p.products == 1 && p.in_fmt != RSA_PF_F16) return false;  // plain-bf16 mode stays on the chunk-barrier kernels
  if (p.in_fmt == RSA_PF_F16 && p.upsample2x) return false;      // fp16 planes: no fused upsampling instantiated
  // final stores: the three-tile shape (the pixel-shuffle heads of SPAN / Compact) and, in three bf16 products over whole chunks, one or two
  // cout tiles (the 64 -> 3 last convolution of RRDBNet / SwinIR's nearest+conv head: the two-stream shape, its second cout tile multiplies
  // zero weights when Cout <= 16 -- the layer is bound by its 256 B of input per pixel, not by the matrix pipes)
  if (p.out_nchw != nullptr) return !p.upsample2x && (ct == 3 || (ct <= 2 && whole && p.products == 3 && p.in_fmt == RSA_PF_BF16));
  if (whole) return ct == 2 || ct == 4 || (ct == 3 && !p.upsample2x);  // (ct == 1 with plane outputs: no layer of the ten architectures has it)
  return ct == 3 && !p.upsample2x;
}
// conv5 of a residual dense block in the one-product fp16 mode (conv_ring.h, XRES): four cout tiles, whole chunks, hi + lo plane output, and
// residual 1 = the first 64 channels of the layer's own input planes (hi + lo; an optional second plane residual), nothing else in the
// epilogue.  RSA_RING_XRES=0 in the environment switches the form off (A/B runs).
bool conv_ring_xres_enabled();
inline bool conv_ring_xres_eligible(const rsa_conv_params& p) {
  return p.ksize == 3 && p.products == 1 && p.in_fmt == RSA_PF_F16 && !p.upsample2x && p.cout == 64 && p.cin_planes >= 8 && (p.cin_planes & 3) == 0 &&
         p.out_nchw == nullptr && p.out_f32 == nullptr && p.res1 == nullptr && p.res2 == nullptr && p.out_hi != nullptr && p.out_lo != nullptr &&
         p.out_fmt == RSA_PF_F16 && p.res_fmt == RSA_PF_F16 && p.res1_hi != nullptr && p.res1_lo != nullptr && p.res1_hi == p.in_hi &&
         p.res_plane_stride == p.in_plane_stride && p.res_batch_stride == p.in_batch_stride && (p.res2_hi == nullptr || p.res2_lo != nullptr) &&
         (p.act == RSA_ACT_NONE || (p.act == RSA_ACT_LRELU && p.act_param >= 0.f && p.act_param <= 1.f)) &&
         (p.lo8_flags == 0 || (p.lo8_flags | RSA_LO8_OUT) == (RSA_LO8_RES1 | RSA_LO8_OUT | (p.res2_hi != nullptr ? RSA_LO8_RES2 : 0)));  // lo halves all fp16, all 8-bit, or 8-bit residuals with an fp16 lo output (the last block of a trunk)
}
// The growth convolutions of a dense block in the one-product fp16 mode: hi-only fp16 plane output, LeakyReLU (slope in [0, 1]) or no
// activation, whole cout tiles, no residual, no f32 map (conv_ring.h, XRES 2: the epilogue shape EM 1 called directly).
inline bool conv_ring_em1_eligible(const rsa_conv_params& p) {
  return p.products == 1 && p.in_fmt == RSA_PF_F16 && p.out_fmt == RSA_PF_F16 && p.out_nchw == nullptr && p.out_f32 == nullptr && p.out_hi != nullptr &&
         p.out_lo == nullptr && p.res1 == nullptr && p.res2 == nullptr && p.res1_hi == nullptr && p.res2_hi == nullptr && (p.cout & 15) == 0 &&
         (p.act == RSA_ACT_NONE || (p.act == RSA_ACT_LRELU && p.act_param >= 0.f && p.act_param <= 1.f));
}
// The re-parameterised 3x3 layers of the SPAN family in the one-product fp16 mode (conv_ring.h, XRES 3): three cout tiles, at most three
// 16-channel half chunks (the 45 KB weight blob lives in LDS), fp16 plane output, Mish / SiLU / SPAB gate (shortcut as fp16 planes) / LeakyReLU /
// no activation, nothing else in the epilogue.  RSA_RING_XRES=0 switches the form off.
inline bool conv_ring_span_eligible(const rsa_conv_params& p) {
  const bool gate = p.act == RSA_ACT_SPAB_GATE;
  return p.ksize == 3 && p.products == 1 && p.in_fmt == RSA_PF_F16 && p.out_fmt == RSA_PF_F16 && !p.upsample2x && p.cout == 48 && (p.cin_planes & 3) != 0 &&
         p.cin_planes <= 6 && p.out_nchw == nullptr && p.out_f32 == nullptr && p.out_hi != nullptr && p.res1 == nullptr && p.res2 == nullptr && p.res2_hi == nullptr &&
         (gate ? (p.res1_hi != nullptr && p.res_fmt == RSA_PF_F16) : p.res1_hi == nullptr) &&
         (gate || p.act == RSA_ACT_MISH || p.act == RSA_ACT_SILU || p.act == RSA_ACT_NONE || (p.act == RSA_ACT_LRELU && p.act_param >= 0.f && p.act_param <= 1.f));
}
// Nearest x2 upsampling + 3x3 as four 2x2 phase convolutions on the source map (conv_ring_up.h): 64 -> 64 channels, LeakyReLU / none,
// split-plane output only -- the upconv layers of RRDBNet and of SwinIR's nearest+conv head.  RSA_CONV_UP2=0 switches it off (A/B runs).
bool conv_up2_enabled();
inline bool conv_ring_up2_eligible(const rsa_conv_params& p) {
  return p.ksize == 3 && p.products == 3 && p.in_fmt == RSA_PF_BF16 && p.out_fmt == RSA_PF_BF16 && p.upsample2x && p.cin_planes == 8 && p.cout == 64 && !(p.H & 1) && !(p.W & 1) && p.out_hi != nullptr &&
         p.out_lo != nullptr && p.out_f32 == nullptr && p.out_nchw == nullptr && p.res1 == nullptr && p.res2 == nullptr && p.res1_hi == nullptr &&
         p.res2_hi == nullptr && (p.act == RSA_ACT_NONE || (p.act == RSA_ACT_LRELU && p.act_param >= 0.f && p.act_param <= 1.f));
}
// Cross-layer fusion (conv_ring_pair.h): two CONSECUTIVE growth convolutions of a residual dense block -- conv1 -> conv2 or conv3 -> conv4 of
// utilities/block.py:454-465 -- run as one launch that streams their common input planes through LDS once.  (a, b) is such a pair iff both are
// one-product fp16 3x3 layers with 32 output channels and a bias + LeakyReLU / linear hi-only plane epilogue (conv_ring_em1_eligible), a writes the
// four planes right behind its own input planes in the same buffer and b reads exactly a's input planes + those four (the dense concatenation).
// RSA_CONV_PAIR=0 in the environment (or rsa_debug_set_pair(0)) makes rsa_conv2d_list launch them one by one again (A/B runs).
bool conv_pair_enabled();
void conv_pair_override(int v);
inline bool conv_pair_eligible(const rsa_conv_params& a, const rsa_conv_params& b) {
  return a.ksize == 3 && b.ksize == 3 && !a.upsample2x && !b.upsample2x && a.cout == 32 && b.cout == 32 && conv_ring_em1_eligible(a) && conv_ring_em1_eligible(b) &&
         a.w_layout == RSA_WL_PAIRS && b.w_layout == RSA_WL_PAIRS && a.cin_planes >= 4 && (a.cin_planes & 3) == 0 && b.cin_planes == a.cin_planes + 4 &&
         a.batch == b.batch && a.H == b.H && a.W == b.W && a.in_hi == b.in_hi && a.in_plane_stride == b.in_plane_stride && a.in_batch_stride == b.in_batch_stride &&
         a.out_hi == a.in_hi && a.out_plane_off == a.cin_planes && a.out_plane_stride == a.in_plane_stride && a.out_batch_stride == a.in_batch_stride &&
         (b.out_hi != b.in_hi || b.out_plane_off >= b.cin_planes) && a.w_packed != nullptr && b.w_packed != nullptr;
}
int conv_launch_pair(const rsa_conv_params& a, const rsa_conv_params& b, hipStream_t stream);  // conv_inst_ringpair.hip; the caller has validated both
unsigned int conv_ring_pair_aborts();

inline int conv_ring_layout(const rsa_conv_params& p) {
  if (conv_up2_enabled() && conv_ring_up2_eligible(p)) return RSA_WL_UPPHASE;
  return (p.cin_planes & 3) == 0 ? RSA_WL_PAIRS : RSA_WL_HALFPAIRS;
}
inline int conv_weight_layout(const rsa_conv_params& p) { return (conv_ring_enabled() && conv_ring_eligible(p)) ? conv_ring_layout(p) : RSA_WL_TAPS; }

// conv_inst_ring*.hip
int conv_launch_ring(const rsa_conv_params& p, hipStream_t stream);
int conv_launch_ring_up2(const rsa_conv_params& p, hipStream_t stream);
// bytes of a packed blob in a given layout (layout 3 is larger than the chunked ones: 32 K steps per 64 x 64 layer)
int64_t packed_weight_bytes(int cout, int cin_planes, int ksize, int products, int layout);
unsigned int conv_ring_aborts();
// What the launcher hands every ring kernel beside the descriptor: the host-visible failure word (pinned, device-mapped; a timed-out hand-off
// adds to it and rsa_check_status reads it without a synchronisation) and the bound of a spin (rsa_debug_set_ring_spin_limit).
struct RingAux {
  unsigned int* fail_word;
  int spin_limit;
};
RingAux ring_aux();  // conv_mfma.hip
unsigned int* range_word();  // conv_mfma.hip: device pointer of the host-visible word rsa_check_finite adds to (NULL: no pinned memory)
// host-visible failure word of the ring kernels (conv_mfma.hip): 0 = nothing to report, else RSA_E_INTERNAL with the count in the error string
int conv_check_status();
void conv_set_ring_spin_limit(int polls);

}  // namespace rsa
