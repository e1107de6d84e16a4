// conv_mfma.hip — argument validation and dispatch of the fused convolution (kernel template: conv_kernel.h; instantiations:
// conv_inst_*.hip; wide 1x1 layers: gemm_k1.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <stdio.h>

#include <atomic>
#include <mutex>

#include "common.h"
#include "resselt_amd.h"

namespace rsa {

int gemm_k1_launch(const rsa_conv_params& p, hipStream_t stream);  // gemm_k1.hip; -100 = not applicable
int conv_launch_k3p3u0(const rsa_conv_params& p, int nct, hipStream_t stream);
int conv_launch_k3p3u1(const rsa_conv_params& p, int nct, hipStream_t stream);
int conv_launch_k3p1u0(const rsa_conv_params& p, int nct, hipStream_t stream);
int conv_launch_k3p1u1(const rsa_conv_params& p, int nct, hipStream_t stream);
int conv_launch_k3p1u0_f16(const rsa_conv_params& p, int nct, hipStream_t stream);
int conv_launch_k3p1u1_f16(const rsa_conv_params& p, int nct, hipStream_t stream);
int conv_launch_k1p1_f16(const rsa_conv_params& p, int nct, hipStream_t stream);
int conv_launch_k3p3u0_f16(const rsa_conv_params& p, int nct, hipStream_t stream);
int conv_launch_k1p3_f16(const rsa_conv_params& p, int nct, hipStream_t stream);
int conv_launch_k1p3(const rsa_conv_params& p, int nct, hipStream_t stream);
int conv_launch_k1p1(const rsa_conv_params& p, int nct, hipStream_t stream);

// ---- status words of the library: one pinned, device-mapped block per process, allocated at the first ring launch or range check ----
//   word 0: hand-offs of the ring kernels that timed out (conv_ring.h)      word 1: non-finite values seen by rsa_check_finite
static std::atomic<unsigned int*> g_fail_host{nullptr};
static std::atomic<unsigned int*> g_fail_dev{nullptr};
static std::atomic<int> g_spin_limit{1 << 18};
static unsigned int* status_words_dev() {
  unsigned int* dev = g_fail_dev.load(std::memory_order_acquire);
  if (dev == nullptr) {
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    dev = g_fail_dev.load(std::memory_order_acquire);
    if (dev == nullptr) {
      unsigned int* host = nullptr;
      void* d = nullptr;
      // (lazily, not at load time: the library must stay loadable, and its symbols checkable, on a host without a GPU; a first launch under
      //  stream capture would allocate inside the capture -- EngineModule runs one eager pass before it captures)
      if (hipHostMalloc((void**)&host, 64, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) == hipSuccess && host != nullptr) {
        host[0] = host[1] = 0u;
        if (hipHostGetDevicePointer(&d, host, 0) == hipSuccess && d != nullptr) {
          g_fail_host.store(host, std::memory_order_release);
          dev = (unsigned int*)d;
          g_fail_dev.store(dev, std::memory_order_release);
        }
      }
      (void)hipGetLastError();  // without the word the kernels still count into g_ring_aborts (rsa_debug_ring_aborts)
    }
  }
  return dev;
}
RingAux ring_aux() { return RingAux{status_words_dev(), g_spin_limit.load(std::memory_order_relaxed)}; }
unsigned int* range_word() {
  unsigned int* dev = status_words_dev();
  return dev == nullptr ? nullptr : dev + 1;
}
void conv_set_ring_spin_limit(int polls) { g_spin_limit.store(polls < 1 ? 1 : polls, std::memory_order_relaxed); }
int conv_check_status() {
  unsigned int* host = g_fail_host.load(std::memory_order_acquire);
  if (host == nullptr) return RSA_OK;
  const unsigned int n = __atomic_exchange_n(&host[0], 0u, __ATOMIC_ACQ_REL);
  const unsigned int r = __atomic_exchange_n(&host[1], 0u, __ATOMIC_ACQ_REL);
  char msg[200];
  if (n != 0) {
    snprintf(msg, sizeof(msg), "ring schedule: %u hand-off(s) between loader and compute waves timed out; the affected launches produced wrong pixels", n);
    return set_error(RSA_E_INTERNAL, msg);
  }
  if (r != 0) {
    snprintf(msg, sizeof(msg), "fp16 range: %u block(s) of a checked tensor hold non-finite values (an activation left the fp16 range of a one-product layer, or the "
                               "input was not finite); run the model with precision 'bf16x3'", r);
    return set_error(RSA_E_FP16_RANGE, msg);
  }
  return RSA_OK;
}
static bool conv_failure_pending() {
  unsigned int* host = g_fail_host.load(std::memory_order_acquire);
  return host != nullptr && __atomic_load_n(host, __ATOMIC_ACQUIRE) != 0;
}

static std::atomic<int> g_ring_override{-1};  // rsa_debug_set_ring: -1 = follow the environment, 0 / 1 = forced (in-process A/B runs)
void conv_ring_override(int v) { g_ring_override.store(v < 0 ? -1 : (v ? 1 : 0)); }
bool conv_up2_enabled() {
  static const bool on = [] {
    const char* e = getenv("RSA_CONV_UP2");
    return !(e != nullptr && e[0] == '0');
  }();
  return on;
}
bool conv_ring_xres_enabled() {
  static const bool on = [] {
    const char* e = getenv("RSA_RING_XRES");
    return !(e != nullptr && e[0] == '0');
  }();
  return on;
}
bool conv_ring_enabled() {
  static const bool on = [] {
    const char* e = getenv("RSA_CONV_RING");
    return e == nullptr || e[0] != '0';
  }();
  const int o = g_ring_override.load(std::memory_order_relaxed);
  return o < 0 ? on : o != 0;
}

// Name of the kernel a descriptor dispatches to (bench.py groups its per-kernel roofline by it; matches the rocprofv3 kernel names).
const char* conv_kernel_name(const rsa_conv_params& p) {
  if (p.w_layout == RSA_WL_UPPHASE) return "rsa::conv_ring_up2 (x2 upsampling as four 2x2 phases)";
  if (p.w_layout != RSA_WL_TAPS) {
    const int ct = (p.cout + 15) >> 4;
    if (p.products == 1) {
      if (ct == 3 && conv_ring_xres_enabled() && conv_ring_span_eligible(p)) return "rsa::conv_ring<3,0,0,HM,f16,1,XRES 3> (SPAN-family 48-channel layer, one fp16 product, weights resident in LDS, direct epilogue)";
      if (ct == 4 && conv_ring_xres_enabled() && conv_ring_xres_eligible(p)) return "rsa::conv_ring<1,0,0,0,f16,1,XRES> (conv5 of a dense block, one fp16 product, residual hi halves from the ring)";
      return ct == 2 ? "rsa::conv_ring<2,0,0,0,f16,1> (Cout<=32, one fp16 product)" : ct == 3 ? (p.out_nchw != nullptr ? "rsa::conv_ring<3,0,1,HM,f16,1> (Cout 33..48, final store, one fp16 product)" : "rsa::conv_ring<3,0,0,HM,f16,1> (Cout 33..48, one fp16 product)") : "rsa::conv_ring<1,0,0,0,f16,1> (Cout 49..64, one fp16 product)";
    }
    if (p.in_fmt == RSA_PF_F16)
      return ct == 2 ? "rsa::conv_ring<2,0,0,0,f16,3> (Cout<=32, three fp16 products)" : ct == 3 ? (p.out_nchw != nullptr ? "rsa::conv_ring<3,0,1,HM,f16,3> (Cout 33..48, final store, three fp16 products)" : "rsa::conv_ring<3,0,0,HM,f16,3> (Cout 33..48, three fp16 products)") : "rsa::conv_ring<1,0,0,0,f16,3> (Cout 49..64, three fp16 products)";
    if (ct <= 2 && p.out_nchw != nullptr) return "rsa::conv_ring<2,0,1> (Cout<=32, final store)";
    return ct == 2 ? "rsa::conv_ring<2,UP,0> (Cout<=32)" : ct == 3 ? (p.out_nchw != nullptr ? "rsa::conv_ring<3,0,1,HM> (Cout 33..48, final store)" : "rsa::conv_ring<3,0,0,HM> (Cout 33..48)") : "rsa::conv_ring<1,UP,0> (Cout 49..64)";
  }
  if (p.ksize == 1 && p.cout >= 96 && p.out_nchw == nullptr) return "rsa::gemm_k1_kernel";
  if (p.ksize == 3 && p.products == 3 && conv_nct(p.cout) == 2 && p.cout <= 32) return "rsa::conv_kernel_pp";
  return p.out_nchw != nullptr ? "rsa::conv_kernel<..., OUTK=1> (final store)" : "rsa::conv_kernel";
}

int conv_nct(int cout) {
  const int ct = (cout + 15) / 16;
  if (ct <= 4) return ct;
  // more than one slab: pick the tile count (4 or 3) that wastes the fewest padded cout-tiles
  const int w4 = ((ct + 3) / 4) * 4 - ct;
  const int w3 = ((ct + 2) / 3) * 3 - ct;
  return (w3 < w4) ? 3 : 4;
}

// every argument check of a descriptor, nothing launched (rsa_conv2d_list validates both halves of a fused pair with it)
int conv_validate(const rsa_conv_params& p) {
  if (p.batch < 1 || p.H < 1 || p.W < 1 || p.cin_planes < 1 || p.cout < 1) return set_error(RSA_E_ARG, "conv: bad geometry");
  if (p.ksize != 1 && p.ksize != 3) return set_error(RSA_E_UNSUPPORTED, "conv: ksize must be 1 or 3");
  if (p.products != 1 && p.products != 3) return set_error(RSA_E_UNSUPPORTED, "conv: products must be 1 or 3");
  if ((unsigned)p.in_fmt > RSA_PF_F16 || (unsigned)p.out_fmt > RSA_PF_F16 || (unsigned)p.res_fmt > RSA_PF_F16 || (unsigned)p.tile_order > 1u)
    return set_error(RSA_E_ARG, "conv: in_fmt / out_fmt / res_fmt must be an rsa_plane_fmt, tile_order 0 or 1");
  if (conv_failure_pending()) return set_error(RSA_E_INTERNAL, "conv: an earlier ring-schedule launch reported a failed hand-off; call rsa_check_status()");
  if (p.upsample2x && (p.ksize != 3 || (p.H & 1) || (p.W & 1))) return set_error(RSA_E_UNSUPPORTED, "conv: upsample2x needs k3 and even H, W");
  if (p.in_hi == nullptr || p.w_packed == nullptr) return set_error(RSA_E_ARG, "conv: null input/weights");
  if (p.products == 3 && p.in_lo == nullptr) return set_error(RSA_E_ARG, "conv: products=3 needs in_lo");
  if (p.act == RSA_ACT_SPAB_GATE && p.res1 == nullptr && p.res1_hi == nullptr) return set_error(RSA_E_ARG, "conv: SPAB gate needs res1");
  if ((p.res1 != nullptr && p.res1_hi != nullptr) || (p.res2 != nullptr && p.res2_hi != nullptr))
    return set_error(RSA_E_ARG, "conv: a residual is either an f32 map or split planes, not both");
  if ((p.res1_lo != nullptr && p.res1_hi == nullptr) || (p.res2_lo != nullptr && p.res2_hi == nullptr)) return set_error(RSA_E_ARG, "conv: residual lo planes without hi planes");
  if ((p.res1_hi != nullptr || p.res2_hi != nullptr) && (p.res_plane_stride * 32 >= (int64_t)1 << 32 || (p.cout & 7)))
    return set_error(RSA_E_UNSUPPORTED, "conv: plane residuals need cout % 8 == 0 and planes below 128 Mpx");
  if (p.act < 0 || p.act > RSA_ACT_PRELU) return set_error(RSA_E_ARG, "conv: bad act");
  if (p.act == RSA_ACT_PRELU && (p.act_vec == nullptr || ((uintptr_t)p.act_vec & 15))) return set_error(RSA_E_ARG, "conv: PReLU needs 16-byte aligned act_vec");
  if (p.out_base != nullptr && p.out_nchw == nullptr) return set_error(RSA_E_ARG, "conv: out_base only applies to the out_nchw store");
  if (((uintptr_t)p.in_hi | (uintptr_t)p.in_lo | (uintptr_t)p.w_packed | (uintptr_t)p.out_hi | (uintptr_t)p.out_lo | (uintptr_t)p.out_f32 |
       (uintptr_t)p.res1 | (uintptr_t)p.res2 | (uintptr_t)p.res1_hi | (uintptr_t)p.res1_lo | (uintptr_t)p.res2_hi | (uintptr_t)p.res2_lo) & 15)
    return set_error(RSA_E_ALIGN, "conv: pointers must be 16-byte aligned");
  if (p.pixel_shuffle > 1 && ((uintptr_t)p.out_nchw & 15)) return set_error(RSA_E_ALIGN, "conv: out_nchw must be 16-byte aligned for a depth-to-space store (vector stores)");
  if (p.in_plane_stride * 64 >= (int64_t)1 << 32) return set_error(RSA_E_UNSUPPORTED, "conv: input plane too large for a 4-plane buffer descriptor (>= 64 Mpx); band the image");
  if (p.out_plane_stride * 32 >= (int64_t)1 << 32 || (int64_t)p.H * p.W * 64 >= (int64_t)1 << 32)
    return set_error(RSA_E_UNSUPPORTED, "conv: output plane too large for 32-bit lane offsets; band the image");
  if ((uintptr_t)p.bias & 15) return set_error(RSA_E_ALIGN, "conv: bias must be 16-byte aligned (and padded to a multiple of 16 floats)");
  if (p.out_nchw != nullptr) {
    if (p.out_hi != nullptr || p.out_f32 != nullptr || p.res1 != nullptr || p.res2 != nullptr || p.res1_hi != nullptr || p.res2_hi != nullptr ||
        p.act == RSA_ACT_SPAB_GATE)
      return set_error(RSA_E_UNSUPPORTED, "conv: out_nchw is a final store: no plane/f32 outputs or residuals with it");
    const int ps = p.pixel_shuffle > 1 ? p.pixel_shuffle : 1;
    if (p.cout % (ps * ps) != 0) return set_error(RSA_E_ARG, "conv: cout not divisible by pixel_shuffle^2");
    if (p.out_dtype < RSA_F32 || p.out_dtype > RSA_U8) return set_error(RSA_E_ARG, "conv: bad out_dtype");
    if (p.out_dtype == RSA_U8 && p.out_base != nullptr) return set_error(RSA_E_UNSUPPORTED, "conv: an 8-bit image store takes no base image");
  }
  if (p.lo8_flags != 0) {
    if ((p.lo8_flags & ~(RSA_LO8_RES1 | RSA_LO8_RES2 | RSA_LO8_OUT)) || p.reserved_lo8 != 0 || p.lo8_batch_stride < 1)
      return set_error(RSA_E_ARG, "conv: lo8_flags / lo8_batch_stride");
    if (((p.lo8_flags & RSA_LO8_RES1) && (p.res1_lo == nullptr || p.res_fmt != RSA_PF_F16)) || ((p.lo8_flags & RSA_LO8_RES2) && (p.res2_lo == nullptr || p.res_fmt != RSA_PF_F16)) ||
        ((p.lo8_flags & RSA_LO8_OUT) && (p.out_lo == nullptr || p.out_fmt != RSA_PF_F16)))
      return set_error(RSA_E_ARG, "conv: an lo8 operand needs its lo pointer and fp16 planes");
    if (p.ksize == 1 && p.cout >= 96 && p.out_nchw == nullptr) return set_error(RSA_E_UNSUPPORTED, "conv: the wide k1 schedule has no lo8 epilogue");
  }
  if (p.w_layout < RSA_WL_TAPS || p.w_layout > RSA_WL_UPPHASE) return set_error(RSA_E_ARG, "conv: unknown w_layout");
  if (p.w_layout != RSA_WL_TAPS) {  // ring schedule (conv_ring.h): the descriptor carries the K order its weights were packed in
    // (layout 3 stays valid when RSA_CONV_UP2 / the debug override change what rsa_conv_weight_layout would answer now)
    const bool ok = p.w_layout == RSA_WL_UPPHASE ? conv_ring_up2_eligible(p)
                                                 : conv_ring_eligible(p) && p.w_layout == ((p.cin_planes & 3) == 0 ? RSA_WL_PAIRS : RSA_WL_HALFPAIRS);
    if (!ok) return set_error(RSA_E_ARG, "conv: w_layout 1 / 2 / 3 on a descriptor the ring schedule does not take that way (ask rsa_conv_weight_layout)");
    if (p.in_plane_stride * 32 >= (int64_t)1 << 32) return set_error(RSA_E_UNSUPPORTED, "conv: input plane too large for 32-bit lane offsets; band the image");
    return RSA_OK;
  }
  if (p.in_fmt == RSA_PF_F16 && p.products == 3 && p.upsample2x)
    return set_error(RSA_E_UNSUPPORTED, "conv: three fp16 products are not compiled with a fused x2 upsampling (use bf16 planes for that layer)");
  return RSA_OK;
}

int conv_launch(const rsa_conv_params& p, hipStream_t stream) {
  const int vrc = conv_validate(p);
  if (vrc != RSA_OK) return vrc;
  if (p.w_layout != RSA_WL_TAPS) {  // ring schedule (conv_ring.h)
    const int rc = conv_launch_ring(p, stream);
    return rc ? set_error(rc, "conv: ring kernel launch failed") : RSA_OK;
  }
  if (p.ksize == 1) {  // wide k1 layers (nn.Linear over tokens): weight-stationary GEMM schedule, gemm_k1.hip
    const int g = gemm_k1_launch(p, stream);
    if (g != -100) return g == 0 ? RSA_OK : set_error(g, "conv: gemm_k1 launch failed");
  }
  const int nct = conv_nct(p.cout);
  const bool f16 = p.in_fmt == RSA_PF_F16;
  int rc;
  if (p.ksize == 3) {
    if (p.upsample2x)
      rc = (p.products == 3) ? conv_launch_k3p3u1(p, nct, stream) : f16 ? conv_launch_k3p1u1_f16(p, nct, stream) : conv_launch_k3p1u1(p, nct, stream);
    else if (p.products == 3)
      rc = f16 ? conv_launch_k3p3u0_f16(p, nct, stream) : conv_launch_k3p3u0(p, nct, stream);
    else
      rc = f16 ? conv_launch_k3p1u0_f16(p, nct, stream) : conv_launch_k3p1u0(p, nct, stream);
  } else if (p.products == 3) {
    rc = f16 ? conv_launch_k1p3_f16(p, nct, stream) : conv_launch_k1p3(p, nct, stream);
  } else {
    rc = f16 ? conv_launch_k1p1_f16(p, nct, stream) : conv_launch_k1p1(p, nct, stream);
  }
  if (rc != 0) return set_error(rc, "conv: kernel launch failed");
  return RSA_OK;
}

}  // namespace rsa
