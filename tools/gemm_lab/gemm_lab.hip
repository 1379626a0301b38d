// gemm_lab.hip -- EXPERIMENT (not part of the product): where does a 4096 x 1344 x 1344 bf16 GEMM
// spend its time on MI355X?  Variants selected by -DVARIANT:
//   0 full kernel, 1 loads only (glds ring, no MFMA), 2 compute only (no global loads after the prologue)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef VARIANT
#define VARIANT 0
#endif
#ifndef WM_
#define WM_ 2
#endif
#ifndef WN_
#define WN_ 4
#endif
#ifndef NSTAGE_
#define NSTAGE_ 4
#endif
constexpr int WM = WM_, WN = WN_;
constexpr int BM = 64 * WM, BN = 48 * WN, BK = 64;
constexpr int kThreads = 64 * WM * WN, kWaves = WM * WN;
constexpr int kStageBytes = (BM + BN) * BK * 2;
constexpr int NSTAGE = NSTAGE_;
constexpr int kLoadsPerWave = (BM + BN) / 8 / kWaves;
static_assert(((BM + BN) / 8) % kWaves == 0 && (BM / 8) % kWaves == 0 && (BN / 8) % kWaves == 0, "even glds split");

struct GemmParams { const uint16_t* x; const uint16_t* w; const float* bias; uint16_t* y; int ldx, ldy, M, N, K, relu; };

__global__ __launch_bounds__(kThreads, 1) void gemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(256))) uint8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int li = lane & 15, lg = lane >> 4;
  const int tiles_m = p.M / BM, tiles_n = p.N / BN;
  int b = blockIdx.x;
  { const int per_xcd = (tiles_m * tiles_n) / 8; if ((tiles_m * tiles_n) % 8 == 0) b = (b % 8) * per_xcd + (b / 8); }
  const int tm0 = (b / tiles_n) * BM, tn0 = (b % tiles_n) * BN;
  auto issue_loads = [&](int kt) __attribute__((always_inline)) {
    uint8_t* xs = lds + (kt % NSTAGE) * kStageBytes;
    uint8_t* ws = xs + BM * BK * 2;
    const int r8 = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int c = 0; c < BM / 8 / kWaves; c++) {
      const int chunk = wave + kWaves * c; const int row = chunk * 8 + r8;
      const uint16_t* src = p.x + (size_t)(tm0 + row) * p.ldx + kt * BK + ((slot ^ (row & 7)) * 8);
      __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(xs + chunk * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < BN / 8 / kWaves; c++) {
      const int chunk = wave + kWaves * c; const int row = chunk * 8 + r8;
      const uint16_t* src = p.w + (size_t)(tn0 + row) * p.K + kt * BK + ((slot ^ (row & 7)) * 8);
      __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(ws + chunk * 1024), 16, 0, 0);
    }
  };
  f32x4 acc[3][4];
#pragma unroll
  for (int tn = 0; tn < 3; tn++)
#pragma unroll
    for (int tm = 0; tm < 4; tm++) acc[tn][tm] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int KT = p.K / BK;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; s++) if (s < KT) issue_loads(s);
  for (int kt = 0; kt < KT; kt++) {
    const int younger = (KT - 1 - kt) < (NSTAGE - 2) ? (KT - 1 - kt) : (NSTAGE - 2);
#if VARIANT != 2
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kLoadsPerWave) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoadsPerWave) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    if (kt == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
#if VARIANT != 2
    if (kt + NSTAGE - 1 < KT) issue_loads(kt + NSTAGE - 1);
#endif
#if VARIANT != 1
    const uint8_t* xs = lds + ((VARIANT == 2 ? kt % (NSTAGE - 1) : kt % NSTAGE)) * kStageBytes;
    const uint8_t* ws = xs + BM * BK * 2;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      bf16x8 bfr[4], afr[3];
#pragma unroll
      for (int tm = 0; tm < 4; tm++) { const int row = wm * 64 + tm * 16 + li; bfr[tm] = *reinterpret_cast<const bf16x8*>(xs + row * 128 + (((4 * kk + lg) ^ (row & 7)) * 16)); }
#pragma unroll
      for (int tn = 0; tn < 3; tn++) { const int row = wn * 48 + tn * 16 + li; afr[tn] = *reinterpret_cast<const bf16x8*>(ws + row * 128 + (((4 * kk + lg) ^ (row & 7)) * 16)); }
#pragma unroll
      for (int tn = 0; tn < 3; tn++)
#pragma unroll
        for (int tm = 0; tm < 4; tm++) acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[tn], bfr[tm], acc[tn][tm], 0, 0, 0);
    }
#endif
  }
#pragma unroll
  for (int tn = 0; tn < 3; tn++) {
    const int n = tn0 + wn * 48 + tn * 16 + 4 * lg;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int tm = 0; tm < 4; tm++) {
      const int m = tm0 + wm * 64 + tm * 16 + li;
      f32x4 v = acc[tn][tm] + bv;
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
      const bf16x4 o = __builtin_convertvector(v, bf16x4);
      *reinterpret_cast<uint2*>(p.y + (size_t)m * p.ldy + n) = __builtin_bit_cast(uint2, o);
    }
  }
}


#ifndef LOADS
#define LOADS 1
#endif
// Software-pipelined variant: fragments of the NEXT k-step are requested before the MFMAs of the
// current one; one barrier per K-tile, placed between the two k-steps.
__global__ __launch_bounds__(kThreads, 1) void gemm_pipe_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(256))) uint8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int li = lane & 15, lg = lane >> 4;
  const int tiles_m = p.M / BM, tiles_n = p.N / BN;
  int b = blockIdx.x;
  { const int per_xcd = (tiles_m * tiles_n) / 8; if ((tiles_m * tiles_n) % 8 == 0) b = (b % 8) * per_xcd + (b / 8); }
  const int tm0 = (b / tiles_n) * BM, tn0 = (b % tiles_n) * BN;
  auto issue_loads = [&](int kt) __attribute__((always_inline)) {
    uint8_t* xs = lds + (kt % NSTAGE) * kStageBytes;
    uint8_t* ws = xs + BM * BK * 2;
    const int r8 = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int c = 0; c < BM / 8 / kWaves; c++) {
      const int chunk = wave + kWaves * c; const int row = chunk * 8 + r8;
      const uint16_t* src = p.x + (size_t)(tm0 + row) * p.ldx + kt * BK + ((slot ^ (row & 7)) * 8);
      __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(xs + chunk * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < BN / 8 / kWaves; c++) {
      const int chunk = wave + kWaves * c; const int row = chunk * 8 + r8;
      const uint16_t* src = p.w + (size_t)(tn0 + row) * p.K + kt * BK + ((slot ^ (row & 7)) * 8);
      __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(ws + chunk * 1024), 16, 0, 0);
    }
  };
  struct Frags { bf16x8 b[4], a[3]; };
  auto read_frags = [&](int kt, int kk, Frags& f) __attribute__((always_inline)) {
    const uint8_t* xs = lds + (kt % NSTAGE) * kStageBytes;
    const uint8_t* ws = xs + BM * BK * 2;
#pragma unroll
    for (int tm = 0; tm < 4; tm++) { const int row = wm * 64 + tm * 16 + li; f.b[tm] = *reinterpret_cast<const bf16x8*>(xs + row * 128 + (((4 * kk + lg) ^ (row & 7)) * 16)); }
#pragma unroll
    for (int tn = 0; tn < 3; tn++) { const int row = wn * 48 + tn * 16 + li; f.a[tn] = *reinterpret_cast<const bf16x8*>(ws + row * 128 + (((4 * kk + lg) ^ (row & 7)) * 16)); }
  };
  f32x4 acc[3][4];
#pragma unroll
  for (int tn = 0; tn < 3; tn++)
#pragma unroll
    for (int tm = 0; tm < 4; tm++) acc[tn][tm] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mfmas = [&](const Frags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int tn = 0; tn < 3; tn++)
#pragma unroll
      for (int tm = 0; tm < 4; tm++) acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[tn], f.b[tm], acc[tn][tm], 0, 0, 0);
  };
  const int KT = p.K / BK;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; s++) if (s < KT) issue_loads(s);
  // tile 0 landed (NSTAGE - 2 younger tiles may be in flight)
  if (NSTAGE - 2 >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kLoadsPerWave) : "memory");
  else if (NSTAGE - 2 == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoadsPerWave) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  Frags f0, f1;
  read_frags(0, 0, f0);
  for (int kt = 0; kt < KT; kt++) {
#ifdef NOLDS
    if (kt == 0)
#endif
    read_frags(kt, 1, f1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f0);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < KT) {
      // tile kt+1 must have landed; tiles kt+2 .. kt+NSTAGE-2 may stay in flight
      const int remaining = KT - 2 - kt;                       // tiles younger than kt+1 that exist
      const int younger = remaining < (NSTAGE - 3) ? remaining : (NSTAGE - 3);
#if LOADS
      if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoadsPerWave) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      __builtin_amdgcn_s_barrier();
#if LOADS
      if (kt + NSTAGE - 1 < KT) issue_loads(kt + NSTAGE - 1);
#endif
#ifndef NOLDS
      read_frags(LOADS ? kt + 1 : (kt + 1) % (NSTAGE - 1), 0, f0);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f1);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int tn = 0; tn < 3; tn++) {
    const int n = tn0 + wn * 48 + tn * 16 + 4 * lg;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int tm = 0; tm < 4; tm++) {
      const int m = tm0 + wm * 64 + tm * 16 + li;
      f32x4 v = acc[tn][tm] + bv;
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = fmaxf(v[r], 0.f);
      const bf16x4 o = __builtin_convertvector(v, bf16x4);
      *reinterpret_cast<uint2*>(p.y + (size_t)m * p.ldy + n) = __builtin_bit_cast(uint2, o);
    }
  }
}

#ifdef PIPE
#define KERNEL gemm_pipe_kernel
#else
#define KERNEL gemm_kernel
#endif

int main(int argc, char** argv) {
  const int M = 4096, K = 1344, N = argc > 1 ? atoi(argv[1]) : 1344;
  uint16_t *x, *w, *y; float* bias;
  hipMalloc(&x, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&y, (size_t)M * N * 2); hipMalloc(&bias, N * 4);
  std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
  for (auto& v : hx) v = 0x3c00 + (rand() & 0x3ff);
  for (auto& v : hw) v = 0x3c00 + (rand() & 0x3ff);
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, N * 4);
  GemmParams p{x, w, bias, y, K, N, M, N, K, 1};
  hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * kStageBytes);
  dim3 grid((M / BM) * (N / BN));
  for (int i = 0; i < 10; i++) KERNEL<<<grid, kThreads, NSTAGE * kStageBytes>>>(p);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  const int R = 200;
  for (int i = 0; i < R; i++) KERNEL<<<grid, kThreads, NSTAGE * kStageBytes>>>(p);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%s LOADS=%d VARIANT %d tile %dx%d waves %d stages %d N=%d grid %d: %.2f us  (%s)\n",
#ifdef PIPE
         "PIPE",
#else
         "BASE",
#endif
         LOADS, VARIANT, BM, BN, kWaves, NSTAGE, N, grid.x, ms * 1e3 / R, hipGetErrorString(hipGetLastError()));
  return 0;
}
