// feed_lab.hip -- EXPERIMENT (not part of the product): how fast can ONE compute unit take the operands of the heads'
// hidden-layer GEMM (128 x 192 tile, 64-deep k-tiles, 8 wavefronts) out of L2, by which path?
//
//   mode 0  both operands by LDS-DMA (buffer_load ... lds), 40 pieces of 1 KB per k-tile: what c4_head_gemm_kernel does
//   mode 1  the X tile only by LDS-DMA (16 pieces)
//   mode 2  the W tile only, global_load_dwordx4 straight into the MFMA A-fragment registers (6 loads per wavefront
//           and k-tile; the two wavefronts that share a W row block both load it)
//   mode 3  1 + 2 together: do the two paths add up or share one pipe?
//   mode 4  both operands global_load_dwordx4 -> registers -> ds_write_b128 (classic register staging)
//   mode 5  W by registers, each row block loaded by ONE wavefront only (what a 1 x 4 or k-split wavefront layout would ask for)
//   mode 6  both operands global_load_dwordx4 -> registers in the DMA pieces' shape (8 rows x 128 bytes per instruction), no LDS write
//   mode 7  W by LDS-DMA (24 pieces) + X by mode 6's register loads (16 pieces): do a coalesced register stream and the DMA add up?
//   mode 8  mode 0 with a 4-deep ring (three k-tiles = 120 KB in flight): is mode 0 bound by bytes in flight?
//
// No MFMA, no fragment reads: the feed alone, with the kernel's ring (3 k-tiles, loads two k-tiles ahead, counted
// vmcnt, one s_barrier per k-tile).  Reports us per launch for K = 1344 and K = 2688 (same tiles, twice the k-tiles):
// the difference / 21 is the time per k-tile.
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/feed_lab/feed_lab.hip -o /tmp/feed_lab && /tmp/feed_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BM = 128, BN = 192, BK = 64, kWaves = 8, NSTAGE = 3;   // (mode 8: 4 stages of LDS, see run())
constexpr int kStageBytes = (BM + BN) * BK * 2;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE>
__global__ __launch_bounds__(512, 1) void feed(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, uint4* __restrict__ sink,
                                               int M, int N, int K, int ld, int tiles_n) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % 2, wn = wave / 2;
  const int tm0 = (blockIdx.x / tiles_n) * BM, tn0 = (blockIdx.x % tiles_n) * BN;
  const int r8 = lane >> 3, slot = lane & 7;
  constexpr bool kDmaX = MODE == 0 || MODE == 1 || MODE == 3 || MODE == 8;
  constexpr bool kDmaW = MODE == 0 || MODE == 7 || MODE == 8;
  constexpr bool kRegW = MODE == 2 || MODE == 3 || MODE == 5;
  constexpr bool kStage = MODE == 4 || MODE == 6 || MODE == 7;     // pieces into registers (6, 7: not written to LDS; 7: the X pieces only)
  constexpr int kRing = MODE == 8 ? 4 : 3;
  // DMA / staging pieces of this wavefront: piece c = wave + 8 i; c < 16: X rows 8c..8c+7, else W rows 8(c-16)..
  constexpr int LP = kDmaW || kStage ? 5 : 2;
  constexpr int kStageBytesR = kStageBytes;
  uint32_t src_off[5];
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const int c = wave + kWaves * i;
    if (c < BM / 8) src_off[i] = (uint32_t)(tm0 + c * 8 + r8) * ld * 2u + (uint32_t)((slot ^ r8) * 16);
    else src_off[i] = (uint32_t)(tn0 + (c - BM / 8) * 8 + r8) * ld * 2u + (uint32_t)((slot ^ r8) * 16);
  }
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((uint32_t)M * ld * 2u), 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((uint32_t)N * ld * 2u), 0x00020000);
  // register path for W: fragment (a, half) of this wavefront: row tn0 + 48 wn + 16 a + (lane & 15), bytes 64 half + 16 (lane >> 4)
  // MODE 5: only wavefronts with wm == 0 load (no duplicate request for a row block)
  const bool loads_w = MODE != 5 || wm == 0;
  uint32_t wreg_off[3];
#pragma unroll
  for (int a = 0; a < 3; a++) wreg_off[a] = (uint32_t)(tn0 + 48 * wn + 16 * a + (lane & 15)) * ld * 2u + 16u * (lane >> 4);
  uint4 wr[4][6];
  uint4 sr[4][5];
  uint4 acc = make_uint4(0, 0, 0, 0);
  const int KT = K / BK;

  auto issue = [&](int kt, auto ring) __attribute__((always_inline)) {
    constexpr int R = decltype(ring)::value;
    uint8_t* st = lds + R * kStageBytes;
    if (kDmaX || kDmaW) {
#pragma unroll
      for (int i = 0; i < LP; i++) {
        const int c = wave + kWaves * i;
        if (!kDmaW && c >= BM / 8) continue;
        if (!kDmaX && c < BM / 8) continue;
        __builtin_amdgcn_raw_ptr_buffer_load_lds((c < BM / 8) ? x_rsrc : w_rsrc, (__attribute__((address_space(3))) void*)(st + c * 1024), 16,
                                                 (int)src_off[i], kt * (BK * 2), 0, 0);
      }
    }
    if (kStage) {
#pragma unroll
      for (int i = 0; i < 5; i++) {
        const int c = wave + kWaves * i;
        if (MODE == 7 && c >= BM / 8) continue;
        const uint8_t* base = (c < BM / 8) ? (const uint8_t*)x : (const uint8_t*)w;
        sr[R][i] = *reinterpret_cast<const uint4*>(base + src_off[i] + kt * (BK * 2));
      }
    }
    if (kRegW && loads_w) {
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int h = 0; h < 2; h++) wr[R][2 * a + h] = *reinterpret_cast<const uint4*>((const uint8_t*)w + wreg_off[a] + kt * (BK * 2) + 64 * h);
    }
  };
  constexpr int kOpsPerTile = (kDmaX ? 2 : 0) + (kDmaW ? 3 : 0) + (kStage ? (MODE == 7 ? 2 : 5) : 0) + (kRegW ? 6 : 0);   // MODE 5's idle wavefronts: fewer, waits are then conservative
  auto consume = [&](int kt, auto ring) __attribute__((always_inline)) {
    constexpr int R = decltype(ring)::value;
    if (MODE == 5 && !loads_w) { wait_vmcnt<0>(); } else { wait_vmcnt<(kRing - 1) * kOpsPerTile>(); }
    if (MODE == 4) {
      uint8_t* st = lds + R * kStageBytes;
#pragma unroll
      for (int i = 0; i < 5; i++) *reinterpret_cast<uint4*>(st + (wave + kWaves * i) * 1024 + lane * 16) = sr[R][i];
    }
    if (MODE == 6 || MODE == 7) {
#pragma unroll
      for (int i = 0; i < 5; i++) {
        if (MODE == 7 && wave + kWaves * i >= BM / 8) continue;
        acc.x ^= sr[R][i].x; acc.y ^= sr[R][i].y; acc.z ^= sr[R][i].z; acc.w ^= sr[R][i].w;
      }
    }
    if (kRegW && loads_w) {
#pragma unroll
      for (int j = 0; j < 6; j++) { acc.x ^= wr[R][j].x; acc.y ^= wr[R][j].y; acc.z ^= wr[R][j].z; acc.w ^= wr[R][j].w; }
    }
    __builtin_amdgcn_s_barrier();
  };
  using R0 = std::integral_constant<int, 0>; using R1 = std::integral_constant<int, 1>; using R2 = std::integral_constant<int, 2>;
  using R3 = std::integral_constant<int, 3>;
  if (kRing == 3) {
    issue(0, R0{}); issue(1, R1{});
    for (int kt = 0; kt < KT; kt += 3) {     // KT % 3 == 0
      issue(kt + 2, R2{}); consume(kt, R0{});
      issue(kt + 3, R0{}); consume(kt + 1, R1{});
      issue(kt + 4, R1{}); consume(kt + 2, R2{});
    }
  } else {
    issue(0, R0{}); issue(1, R1{}); issue(2, R2{});
    for (int kt = 0; kt + 3 < KT; kt += 4) {   // KT = 21 or 42: the last one or two k-tiles are not consumed (timing only)
      issue(kt + 3, R3{}); consume(kt, R0{});
      issue(kt + 4, R0{}); consume(kt + 1, R1{});
      issue(kt + 5, R1{}); consume(kt + 2, R2{});
      issue(kt + 6, R2{}); consume(kt + 3, R3{});
    }
  }
  wait_vmcnt<0>();
  if (kStage || kDmaX) acc.x ^= *reinterpret_cast<const uint32_t*>(lds + tid * 4);
  if (acc.x == 0x12345678u) sink[blockIdx.x * 512 + tid] = acc;      // keeps the loads alive, (almost) never taken
}

template <int MODE>
float run(const uint16_t* x, const uint16_t* w, uint4* sink, int M, int N, int K, int ld, int reps) {
  const int tiles_n = N / BN, tiles = (M / BM) * tiles_n;
  auto k = feed<MODE>;
  const int kLds = (MODE == 8 ? 4 : 3) * kStageBytes;
  HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  for (int i = 0; i < 5; i++) k<<<tiles, 512, kLds>>>(x, w, sink, M, N, K, ld, tiles_n);
  hipEvent_t a, b;
  HIP(hipEventCreate(&a)); HIP(hipEventCreate(&b));
  HIP(hipEventRecord(a));
  for (int i = 0; i < reps; i++) k<<<tiles, 512, kLds>>>(x, w, sink, M, N, K, ld, tiles_n);
  HIP(hipEventRecord(b));
  HIP(hipEventSynchronize(b));
  float ms = 0;
  HIP(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 2048, ld = 2688 + 512;   // rows padded past K so that the reads two k-tiles past the end stay inside
  uint16_t *x, *w; uint4* sink;
  HIP(hipMalloc(&x, (size_t)(M + 8) * ld * 2)); HIP(hipMalloc(&w, (size_t)(2688 + 8) * ld * 2)); HIP(hipMalloc(&sink, 512 * 512 * 16));
  HIP(hipMemset(x, 1, (size_t)(M + 8) * ld * 2)); HIP(hipMemset(w, 2, (size_t)(2688 + 8) * ld * 2));
  const char* names[] = {"0 dma X+W (the kernel today)", "1 dma X only", "2 regs W only (both wm load)", "3 dma X + regs W", "4 regs X+W -> ds_write", "5 regs W only (one wavefront per row block)", "6 regs X+W, coalesced pieces, no LDS write",
                         "7 dma W + coalesced regs X", "8 dma X+W, 4-deep ring"};
  for (int N : {2688, 1344}) {
    printf("M %d N %d: %d workgroups of 128 x 192\n", M, N, (M / BM) * (N / BN));
    for (int mode = 0; mode < 9; mode++) {
      float t[2];
      for (int ki = 0; ki < 2; ki++) {
        const int K = ki ? 2688 : 1344;
        switch (mode) {
          case 0: t[ki] = run<0>(x, w, sink, M, N, K, ld, 200); break;
          case 1: t[ki] = run<1>(x, w, sink, M, N, K, ld, 200); break;
          case 2: t[ki] = run<2>(x, w, sink, M, N, K, ld, 200); break;
          case 3: t[ki] = run<3>(x, w, sink, M, N, K, ld, 200); break;
          case 4: t[ki] = run<4>(x, w, sink, M, N, K, ld, 200); break;
          case 5: t[ki] = run<5>(x, w, sink, M, N, K, ld, 200); break;
          case 6: t[ki] = run<6>(x, w, sink, M, N, K, ld, 200); break;
          case 7: t[ki] = run<7>(x, w, sink, M, N, K, ld, 200); break;
          default: t[ki] = run<8>(x, w, sink, M, N, K, ld, 200); break;
        }
      }
      printf("  mode %-46s K=1344 %6.2f us  K=2688 %6.2f us  -> %.3f us per k-tile (%.0f cycles at 2.1 GHz)\n", names[mode], t[0], t[1], (t[1] - t[0]) / 21.0,
             (t[1] - t[0]) / 21.0 * 2100.0);
    }
  }
  return 0;
}
