// Diagnostic only (tools/build_variant.py NAME WORK -DC4_TIMELINE; not part of libc4a0_hip.so): every workgroup of the
// tower, the hidden-layer GEMMs and the output + step launch appends {kind, tag, block, CU, start, end} (s_memrealtime,
// 100 MHz) to a buffer the host hands in, so that the schedule of two sessions' kernels AS THE DEVICE RAN IT can be read --
// rocprofv3's kernel trace serialises the dispatches of the two queues and shows each kernel alone (tools/pair_timeline.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef C4_TIMELINE
struct C4TlRec { uint32_t kind, tag, block, hw; uint64_t t0, t1; };   // 32 bytes; hw = HW_ID | XCC_ID << 28
struct C4TlBuf { unsigned long long n, cap; C4TlRec rec[1]; };
namespace { __device__ C4TlBuf* c4_tl_buf = nullptr; }                // one per translation unit, all set to the same buffer
// The record's index is drawn at the START of the workgroup (the atomic's round trip is hidden under the kernel) and the record
// is written with plain stores at the end: drawing it at the end put ~2 us on every launch's tail (134 vs 104 us per round).
#define C4_TL_BEGIN()                                                                                           \
  const uint64_t c4_tl_t0 = __builtin_amdgcn_s_memrealtime();                                                   \
  C4TlBuf* const c4_tl_b = c4_tl_buf;                                                                           \
  unsigned long long c4_tl_i = ~0ull;                                                                           \
  if (threadIdx.x == 0 && c4_tl_b) c4_tl_i = atomicAdd(&c4_tl_b->n, 1ull)
#define C4_TL_END(kind, tagp)                                                                                   \
  do {                                                                                                          \
    if (threadIdx.x == 0 && c4_tl_b && c4_tl_i < c4_tl_b->cap) {                                                \
      uint32_t hw, xcc;                                                                                         \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                          \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                        \
      C4TlRec r{(uint32_t)(kind), (uint32_t)((uintptr_t)(tagp) >> 8), (uint32_t)blockIdx.x,                     \
                (hw & 0x0FFFFFFFu) | (xcc << 28), c4_tl_t0, __builtin_amdgcn_s_memrealtime()};                  \
      c4_tl_b->rec[c4_tl_i] = r;                                                                                \
    }                                                                                                           \
  } while (0)
#define C4_TL_SETTER(name)                                                                                      \
  extern "C" int name(void* buf) {                                                                              \
    return hipMemcpyToSymbol(HIP_SYMBOL(c4_tl_buf), &buf, sizeof buf) == hipSuccess ? 0 : 1;                    \
  }
#else
#define C4_TL_BEGIN() do { } while (0)
#define C4_TL_END(kind, tagp) do { } while (0)
#define C4_TL_SETTER(name)
#endif
