// Shared device/host helpers for libsgta_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SGTA_OK 0
#define SGTA_EBADARG (-1)
#define SGTA_ETOOBIG (-2)   // a size beyond a stated maximum of an entry point that owns its scratch memory

// launch-site error check: returns hipError_t (>0) from the enclosing extern "C" entry point
#define SGTA_LAUNCH_CHECK()                         \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Tuning switches of the tools/ scripts (kernel-variant A/B runs): read from the environment only in a build with
// -DSGTA_TOOLS (make TOOLS=1); the product library has no environment switches and always takes ``dflt``.
#ifdef SGTA_TOOLS
#include <cstdlib>
static inline int sgta_tool_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
#else
static inline int sgta_tool_env(const char*, int dflt) { return dflt; }
#endif

// Host-side caches of per-device facts (resident work-groups, function attributes) are indexed by the CURRENT device:
// one process may drive several GPUs (tests, replicas), and hipFuncSetAttribute is per-device state.
#define SGTA_MAX_DEVICES 64
static inline int sgta_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev % SGTA_MAX_DEVICES;
}

// XCD-aware 1-D block remap: blocks b and b+8 share an XCD (round-robin dispatch, speed only).
// Logical ids are laid out so that ``inner`` consecutive logical ids (e.g. the cout tiles of one
// pixel tile) run back-to-back on ONE XCD and hit its L2.  Bijective for any (n_outer, inner);
// logical id >= n_outer*inner means "no work".
__device__ __forceinline__ void xcd_remap(int bid, int inner, int& outer_id, int& inner_id) {
  int xcd = bid & 7;
  int slot = bid >> 3;
  inner_id = slot % inner;
  outer_id = (slot / inner) * 8 + xcd;
}

// Storage element of the 16-bit data path: bfloat16 (F16 = false: 8 significant bits, fp32's range) or IEEE half (F16 = true:
// 11 significant bits, |x| <= 65504 -- values beyond are clamped on store, never written as infinity).  Both run on the same
// matrix instructions' shapes with fp32 accumulation; only the operand format and the store conversion differ.
typedef float f32x8 __attribute__((ext_vector_type(8)));
template <bool F16> struct H16;
template <> struct H16<false> {
  typedef __bf16 T;
  typedef __bf16 x4 __attribute__((ext_vector_type(4)));
  typedef __bf16 x8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x16 mfma32(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ x4 cvt4(f32x4 v) { return __builtin_convertvector(v, x4); }
  static __device__ __forceinline__ x8 cvt8(f32x8 v) { return __builtin_convertvector(v, x8); }
};
template <> struct H16<true> {
  typedef _Float16 T;
  typedef _Float16 x4 __attribute__((ext_vector_type(4)));
  typedef _Float16 x8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ f32x16 mfma32(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ x4 cvt4(f32x4 v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(v[j], -65504.0f), 65504.0f);
    return __builtin_convertvector(v, x4);
  }
  static __device__ __forceinline__ x8 cvt8(f32x8 v) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fminf(fmaxf(v[j], -65504.0f), 65504.0f);
    return __builtin_convertvector(v, x8);
  }
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }
