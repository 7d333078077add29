// common.h -- shared host-side plumbing for libgenpc_hip.so (error state,
// per-device scratch pool, launch geometry helpers).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define GENPC_API extern "C" __attribute__((visibility("default")))

namespace genpc {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kNumCU = 256;        // MI355X
constexpr int kNumSIMD = kNumCU * 4;

// Records the message for genpc_last_error() and prints it like the reference
// does (chamfer3D.cu:147, emd_cuda.cu:278); returns false on error.
bool check(hipError_t e, const char *what);
void set_error(const char *msg);

// Grow-only per-device scratch (split-target partial minima, EMD work lists).
// Stream-ordered use only: every consumer is enqueued on the same stream as
// its producer, so reuse across calls on one stream is safe; calls on different
// streams of one device get different slots.
// If `zero_prefix` > 0 the first zero_prefix bytes of a NEWLY allocated block are
// zeroed (stream-ordered) and *fresh is set; an existing block is returned as is.
void *workspace(int slot, size_t bytes, hipStream_t stream, bool *fresh = nullptr, size_t zero_prefix = 0);
int arith_mode();

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline long long ceil_div64(long long a, long long b) { return (a + b - 1) / b; }

}  // namespace genpc
