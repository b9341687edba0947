// Shared device/host helpers for libfgvc_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/fgvc_hip.h"

namespace fgvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WAVE = 64;
constexpr int IDX_EMPTY = 0x7fffffff;

// thread-local error text, exported through fgvc_last_error()
void set_error(const char* fmt, ...);

#define FGVC_REQUIRE(cond, code, ...)        \
  do {                                       \
    if (!(cond)) {                           \
      ::fgvc::set_error(__VA_ARGS__);        \
      return (code);                         \
    }                                        \
  } while (0)

#define FGVC_CHECK_LAUNCH(what)                                                   \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      ::fgvc::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));   \
      return FGVC_ERR_LAUNCH;                                                     \
    }                                                                             \
  } while (0)

// Sorted (score desc, index asc) list of the K best candidates seen so far; lives in VGPRs
// (every loop over K is fully unrolled so the arrays are statically indexed).
template <int K>
struct TopK {
  float v[K];
  int ix[K];

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      v[j] = -INFINITY;
      ix[j] = IDX_EMPTY;
    }
  }
  __device__ __forceinline__ bool accepts(float s, int id) const {
    return s > v[K - 1] || (s == v[K - 1] && id < ix[K - 1]);
  }
  // bubble (s,id) down the list; the displaced tail element falls off the end
  __device__ __forceinline__ void insert(float s, int id) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const bool b = s > v[j] || (s == v[j] && id < ix[j]);
      const float tv = v[j];
      const int ti = ix[j];
      v[j] = b ? s : tv;
      ix[j] = b ? id : ti;
      s = b ? tv : s;
      id = b ? ti : id;
    }
  }
};

// soft-argmax read-out order: value desc, HIGHER index first among equals
template <int K>
struct TopKHi {
  float v[K];
  int ix[K];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      v[j] = -INFINITY;
      ix[j] = -1;
    }
  }
  __device__ __forceinline__ bool accepts(float s, int id) const {
    return s > v[K - 1] || (s == v[K - 1] && id > ix[K - 1]);
  }
  __device__ __forceinline__ void insert(float s, int id) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const bool b = s > v[j] || (s == v[j] && id > ix[j]);
      const float tv = v[j];
      const int ti = ix[j];
      v[j] = b ? s : tv;
      ix[j] = b ? id : ti;
      s = b ? tv : s;
      id = b ? ti : id;
    }
  }
};

// Blocks are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2): give every XCD a
// CONTIGUOUS chunk of the logical grid so neighbouring tiles (which share key windows) hit one L2.
// Bijective for any n (cdna_hip_programming.md T1).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

__host__ __device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__host__ __device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__host__ __device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace fgvc
