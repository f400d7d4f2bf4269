// Shared device/host helpers for the LaneGCN hot-path kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lgcn.h"

namespace lgcn {

constexpr int kC = LGCN_C;          // channels
constexpr int kSub = LGCN_TM;       // rows of one CSR sub-tile (16)
constexpr int kLDA = kC + 4;        // padded LDS row stride (floats): 528 B rows,
                                    // conflict-free for ds_read_b128 over 16 rows
constexpr int kWave = 64;

using f32x16 = float __attribute__((ext_vector_type(16)));

#define LGCN_CHECK_PTR(p) do { if ((p) == nullptr) return LGCN_EINVAL; } while (0)
#define LGCN_CHECK_ALIGN16(p) do { if ((reinterpret_cast<uintptr_t>(p) & 15u) != 0) return LGCN_EALIGN; } while (0)

// ReLU that keeps a NaN a NaN, like ATen's relu / clamp_min (v_max_f32 -- fmaxf -- returns the other operand):
// an overflow of the 16-bit operand planes (inf - inf) or a NaN in the input must reach the output, where the
// caller can see it, instead of being clamped to a plausible-looking zero on the way.  gfx950 has the IEEE 754-2019
// maximum as one instruction (v_maximum3_f32).
#ifdef LGCN_RELU_CND      // diagnostic build only (make relucnd, tools/relu_variant_check.py): the compare + select form
__device__ __forceinline__ float relu_nan(float x) { return x < 0.f ? 0.f : x; }
#else
__device__ __forceinline__ float relu_nan(float x) { return __builtin_elementwise_maximum(x, 0.f); }
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a release/acquire fence over ALL address spaces:
// hipcc puts s_waitcnt vmcnt(0) in front of the s_barrier, i.e. every global load still in flight (prefetched weight
// fragments, rows requested for a later phase) and every global store is waited for at each barrier.  Where the threads
// of a workgroup talk to each other through LDS alone, this barrier lets that traffic stay in flight.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LGCN_OK : static_cast<int>(e);
}

// Blocks b and b+8 share an XCD (round-robin dispatch); give every XCD one
// contiguous chunk of tiles so that neighbouring tiles (same scene, shared
// gather rows) hit the same L2.  Bijective for any n.
__device__ __forceinline__ int xcd_chunk_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = bid & 7, k = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

}  // namespace lgcn
