// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "licv_hip.h"

typedef unsigned short bf16_t;   // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short short8;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(4))) float floatx4;
typedef __attribute__((ext_vector_type(16))) float floatx16;

__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// round-to-nearest-even f32 -> bf16; NaN stays NaN (the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&h);
}
// value of f after a round trip through bf16 (models the rounding point of an unfused bf16 torch op)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// host-side error plumbing (defined in abi.cpp)
int licv_set_error(int code, const char* fmt, ...);
#define LICV_CHECK_ARG(cond, ...) do { if (!(cond)) return licv_set_error(LICV_E_BADARG, __VA_ARGS__); } while (0)
#define LICV_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); \
    if (e__ != hipSuccess) return licv_set_error(LICV_E_HIP, "%s: %s", __func__, hipGetErrorString(e__)); } while (0)
