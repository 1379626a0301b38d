// c4_head_gemm.hip -- the hidden layers of ConnectFourNet's two heads (reference src/c4a0/nn.py:75-100:
// Linear(F, F) + BatchNorm1d + ReLU, BN folded) as a hand-written bf16 MFMA GEMM for gfx950:
//
//     Y[M, N] = act(X[M, K] . W[N, K]^T + bias[N])        X, W, Y bf16; accumulation and bias in f32
//
// Why not the library GEMM: the evaluator must be a FUNCTION OF THE POSITION.  hipBLASLt picks its
// kernel (tile, MFMA shape, stream-K split) from M, so the low bits of a position's outputs depended
// on the batch it sat in and on its row -- and with them the samples of a whole play_games call
// (callback mode vs device mode, resident_games, concurrent sessions, tail narrowing).  Here every
// output element is ONE fixed chain: k-tiles ascending, two v_mfma_f32_16x16x32_bf16 per 64-deep
// k-tile, no split-K, no atomics -- the same bits whatever M, whatever row, whichever tile
// configuration below computes it (the configurations differ in what a workgroup owns, never in an
// element's summation order).
//
// Mapping: D[n][m] = sum_k W[n][k] X[m][k] -- the weights are the MFMA's A operand, the activations its
// B operand, so a lane ends up with 4 consecutive output features of one board: one 8-byte store.
// A workgroup owns a BM x BN tile of Y; X and W k-tiles (64 deep = 128-byte rows) stream global -> LDS
// by direct-to-LDS DMA (global_load_lds_dwordx4, 1 KB per wave-instruction) into an NSTAGE ring; the
// 16-byte slot of k-group g of row r lives at slot g ^ (r & 7) (the XOR goes on the per-lane SOURCE
// address, the LDS image stays lane-linear), which makes every ds_read_b128 of a fragment bank-conflict
// free.  One raw s_barrier per k-tile, counted vmcnt waits: loads stay in flight across barriers.
// The DMA pieces of a k-tile are issued one or two at a time BETWEEN the MFMA groups of an earlier k-tile
// (never as a burst in front of them), fragment reads are software-pipelined across the two 32-deep
// halves of a k-tile, and hipcc's counted lgkmcnt waits let MFMAs start on the first fragments.
// Default: 128 x 192 tile, 8 wavefronts (64 x 48 each), 3-deep ring (120 KB) -- chosen in the bench, see
// c4_linear_bf16 below; the other configurations are kept as measured alternatives (same bits).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <type_traits>

#include "../../include/c4a0_hip.h"
#include "c4_host.hpp"
#include "c4_timeline.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;   // k-tile depth: 128-byte rows in LDS

struct GemmParams {
  const uint16_t* x;     // [M][ldx]
  const uint16_t* w;     // [N][K]
  const float* bias;     // [N]
  uint16_t* y;           // [M][ldy]
  uint32_t M, N, K, ldx, ldy, relu;
  // tile of block b (host-computed, launch_common): xcd = b & xcd_mask, idx = b >> xcd_shift,
  //   tm = (xcd >> xn_log2) * rm + idx / rn,  tn = (xcd & ((1 << xn_log2) - 1)) * rn + idx % rn
  // (idx / rn as a multiplication by rn_magic).  XCD rectangles: xcd_mask 7, xcd_shift 3; plain row-major order: 0, 0, rn = tiles_n.
  uint32_t xcd_mask, xcd_shift, xn_log2, rm, rn, rn_magic;
};

// Kernel arguments are FLAT scalars (14 dwords), not a struct: with -mllvm -amdgpu-kernarg-preload-count=16 (build.py)
// the command processor hands them to the wavefront in SGPRs at launch, and the kernel no longer opens with two
// dependent scalar-load round trips to the kernarg segment (0.84 us from entry to the last prologue DMA issue with
// the struct, measured with the diagnostic build's stamps).
#define C4_GEMM_ARGS const uint16_t* __restrict__ a_x, const uint16_t* __restrict__ a_w, const float* __restrict__ a_bias, uint16_t* __restrict__ a_y, \
                     uint32_t a_m, uint32_t a_nk, uint32_t a_ld, uint32_t a_flags, uint32_t a_rmrn, uint32_t a_magic
#define C4_GEMM_UNPACK()                                                                                                              \
  GemmParams p;                                                                                                                        \
  p.x = a_x; p.w = a_w; p.bias = a_bias; p.y = a_y; p.M = a_m; p.N = a_nk & 0xFFFFu; p.K = a_nk >> 16; p.ldx = a_ld & 0xFFFFu;         \
  p.ldy = a_ld >> 16; p.relu = a_flags & 0xFFu; p.xn_log2 = (a_flags >> 8) & 0xFFu; p.xcd_mask = (a_flags >> 16) & 0xFFu;              \
  p.xcd_shift = a_flags >> 24; p.rm = a_rmrn & 0xFFFFu; p.rn = a_rmrn >> 16; p.rn_magic = a_magic

// Tile of block `blk` (see GemmParams).  ONE definition for the kernels and for the host-side map that the CPU tests walk
// (c4_linear_bf16_tile_map): idx / rn is a multiplication by rn_magic = 2^32 / rn + 1, exact while idx * rn < 2^32 -- except
// for rn == 1, whose magic does not fit 32 bits (the truncated magic sent every block but the first to tn = idx,
// past N): there the quotient is idx itself.  All operands are wavefront-uniform: scalar instructions on the device.
__host__ __device__ __forceinline__ void tile_of_block(uint32_t blk, uint32_t xcd_mask, uint32_t xcd_shift, uint32_t xn_log2, uint32_t rm, uint32_t rn,
                                                       uint32_t rn_magic, int& tm, int& tn) {
  const uint32_t xcd = blk & xcd_mask, idx = blk >> xcd_shift;
  const uint32_t qn = rn == 1u ? idx : (uint32_t)(((uint64_t)idx * rn_magic) >> 32);   // idx / rn
  tm = (int)((xcd >> xn_log2) * rm + qn);
  tn = (int)((xcd & ((1u << xn_log2) - 1u)) * rn + (idx - qn * rn));
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#if defined(C4_PHASE_STAMPS) && !defined(C4_GEMM_CLOCK)
#define C4_GEMM_CLOCK   // the diagnostic build carries the GEMM clock stamps; -DC4_GEMM_CLOCK alone = the product kernels + these stamps
#endif
#ifdef C4_GEMM_CLOCK
// Diagnostic builds only (tools/clock_probe.py, tools/bench_clock.py): (a) the shader clock a GEMM's main loop actually ran at -- shader cycles
// (s_memtime) over constant 100 MHz ticks (s_memrealtime) -- and (b) where a workgroup's time goes: stamps 0 entry,
// 1 prologue issued, 2 first k-tile landed (first barrier passed), 3 main loop done, 4 tail DMA drained, 5 bias arrived,
// 6 stores issued, 7 stores acknowledged; summed over the first wavefront of every workgroup, plus the earliest entry
// and the latest exit of all workgroups since the last reset.
__device__ unsigned long long c4_gemm_clk[16];   // {cycles, ticks, workgroups, -, phase ticks [7], -, min entry, max exit}
#define C4_CLK_DECL() unsigned long long clk_ts[8]; unsigned long long clk_c0 = 0, clk_c1 = 0
#define C4_GSTAMP(i) do { clk_ts[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define C4_CLK_BEGIN() do { clk_c0 = __builtin_amdgcn_s_memtime(); } while (0)
#define C4_CLK_END() do { clk_c1 = __builtin_amdgcn_s_memtime(); } while (0)
#ifdef C4_PHASE_STAMPS
#define C4_CLK_SAMPLE() true                       /* the diagnostic build: every workgroup, every phase */
#else
#define C4_CLK_SAMPLE() ((blockIdx.x & 31) == 5)   /* -DC4_GEMM_CLOCK alone (tools/bench_clock.py): one workgroup in 32, so that the atomics do not slow the run being measured */
#endif
#define C4_CLK_FLUSH()                                                                                       \
  do {                                                                                                       \
    if (threadIdx.x == 0 && C4_CLK_SAMPLE()) {                                                               \
      atomicAdd(&c4_gemm_clk[0], clk_c1 - clk_c0);                                                           \
      atomicAdd(&c4_gemm_clk[1], clk_ts[3] - clk_ts[1]);                                                     \
      atomicAdd(&c4_gemm_clk[2], 1ull);                                                                      \
      if (C4_CLK_PHASES) {                                                                                   \
        for (int i_ = 0; i_ < 7; i_++) atomicAdd(&c4_gemm_clk[4 + i_], clk_ts[i_ + 1] - clk_ts[i_]);         \
        atomicMin(&c4_gemm_clk[12], clk_ts[0]);                                                              \
        atomicMax(&c4_gemm_clk[13], clk_ts[7]);                                                              \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)
#ifdef C4_PHASE_STAMPS
#define C4_CLK_PHASES 1
#else
#define C4_CLK_PHASES 0
#endif
#else
#define C4_CLK_DECL() do { } while (0)
#define C4_GSTAMP(i) do { } while (0)
#define C4_CLK_BEGIN() do { } while (0)
#define C4_CLK_END() do { } while (0)
#define C4_CLK_FLUSH() do { } while (0)
#endif


typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- epilogue shared by both kernels.  The MFMA leaves a lane with 4 consecutive output features of one board (C/D map:
// row = 4 (lane >> 4) + reg, col = lane & 15), i.e. 8 bytes of bf16; stored like that, a wavefront needs TM x TN store
// instructions of 8 bytes per lane, and a launch's last microsecond is their ISSUE (measured with the phase stamps of
// the diagnostic build: 1.3 us of a 15.8 us workgroup).  Two 16-row tiles of the same features are therefore exchanged
// between the 16-lane rows of the wavefront (v_permlane16_swap_b32: odd rows of one register <-> even rows of the other),
// after which every lane holds 8 consecutive features = 16 bytes of ONE board: half the store instructions, twice the
// bytes per row segment.  Values are untouched (a permutation of registers): same bits as the 8-byte form.
// (Write-through `sc1` stores, so that the launch leaves no dirty lines for its end-of-kernel write-back, measured
// 2.4 % SLOWER in the bench, same box: plain stores stay.)
// max(v, 0) as ONE instruction.  fmaxf(v, 0.f) compiles to two (hipcc first canonicalises v with v_max v, v, v: sixteen instead of
// eight VALU instructions per stored tile pair in the epilogue, which is issue-bound); v_max_f32 in IEEE mode already is maxNum:
// the same result for every v, NaN -> 0 included.
__device__ __forceinline__ float relu1(float v) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
  return r;
}

template <int TM, int TN, bool RELU>
__device__ __forceinline__ void store_wave_tile_impl(const f32x4 (&acc)[TN][TM], const f32x4 (&bias_v)[TN], const GemmParams& p, int m_wave0, int n_wave0,
                                                     int li, int lg) {
  auto finish = [&](const f32x4& a, const f32x4& bv) __attribute__((always_inline)) {
    f32x4 v = a + bv;
    if (RELU) {
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = relu1(v[r]);
    }
    return __builtin_bit_cast(uint2, __builtin_convertvector(v, bf16x4));
  };
  // 16-byte stores: this lane writes board m16 + 32 pb, features n16 + 16 a .. + 7
  const int m16 = m_wave0 + 16 * (lg & 1) + li;
  uint16_t* y16 = p.y + (size_t)m16 * p.ldy + (n_wave0 + 8 * (lg >> 1));
  const size_t pair_step = (size_t)32 * p.ldy;
#pragma unroll
  for (int pb = 0; pb < TM / 2; pb++) {
#pragma unroll
    for (int a = 0; a < TN; a++) {
      const uint2 o0 = finish(acc[a][2 * pb], bias_v[a]), o1 = finish(acc[a][2 * pb + 1], bias_v[a]);
      const auto sx = __builtin_amdgcn_permlane16_swap(o0.x, o1.x, false, false);
      const auto sy = __builtin_amdgcn_permlane16_swap(o0.y, o1.y, false, false);
      const u32x4 v = {sx[0], sy[0], sx[1], sy[1]};
      if (m16 + 32 * pb < (int)p.M) *reinterpret_cast<u32x4*>(y16 + a * 16) = v;
    }
    y16 += pair_step;
  }
  if (TM & 1) {   // an odd tile left over (48-row wavefront tiles): 8 bytes per lane as the MFMA left them
    const int m8 = m_wave0 + 16 * (TM - 1) + li;
    uint16_t* y8 = p.y + (size_t)m8 * p.ldy + (n_wave0 + 4 * lg);
    if (m8 < (int)p.M) {
#pragma unroll
      for (int a = 0; a < TN; a++) *reinterpret_cast<uint2*>(y8 + a * 16) = finish(acc[a][TM - 1], bias_v[a]);
    }
  }
}

template <int TM, int TN>
__device__ __forceinline__ void store_wave_tile(const f32x4 (&acc)[TN][TM], const f32x4 (&bias_v)[TN], const GemmParams& p, int m_wave0, int n_wave0,
                                                int li, int lg) {
  if (p.relu) store_wave_tile_impl<TM, TN, true>(acc, bias_v, p, m_wave0, n_wave0, li, lg);   // one uniform branch, not one per tile
  else store_wave_tile_impl<TM, TN, false>(acc, bias_v, p, m_wave0, n_wave0, li, lg);
}

// NLOAD = 0: every wavefront computes AND issues its share of the DMA pieces (interleaved with its MFMAs).
// NLOAD > 0: wave specialisation -- WM x WN wavefronts compute (fragment reads + MFMAs only), NLOAD more wavefronts do nothing
// but issue the DMA pieces and wait for them; the k-tile barrier is the only thing the two kinds share.  Why: a CU accepts one
// 1 KB DMA piece per ~25 cycles (tools/feed_lab: 40 pieces per k-tile = 1 000 cycles with nothing else running) and an issuing
// wavefront is held 60-185 cycles per piece; in the NLOAD = 0 form those cycles come out of the wavefronts that should be
// issuing MFMAs (768 cycles of matrix work per k-tile and SIMD take 1 270-1 320).  Same LDS image, same fragment reads, same
// MFMA order: same bits.
// STAG = 1 (8 computing wavefronts, NLOAD = 0): wavefronts 4-7 -- the partners of wavefronts 0-3 on their SIMDs -- run half a k-tile
// behind: the second half's MFMAs of a k-tile are deferred past the next barrier (their fragments are in registers), so that
// right after a barrier one wavefront of every SIMD multiplies while its partner waits for its fragment reads.  Every
// accumulator still sees its MFMAs in the same order: same bits.
// WST = 2 (with NSTAGE = 3): a SPLIT ring -- three stages of the X operand, two of the W operand -- for the 256 x 192 tile, whose
// three full stages (168 KB) do not fit a CU's 160 KB: 3 x 32 + 2 x 24 = 144 KB.  Per k-tile a wavefront issues the W pieces of
// k-tile kt + 1 FIRST (they have the rest of this k-tile to land), then the X pieces of k-tile kt + 2.  Same LDS image per
// stage, same fragment reads, same MFMA order: same bits.
template <int BM, int BN, int WM, int WN, int NSTAGE, int MINW, int NLOAD = 0, int STAG = 0, int WST = 0>
__global__ __launch_bounds__(64 * (WM * WN + NLOAD), MINW) void c4_head_gemm_kernel(C4_GEMM_ARGS) {
  C4_GEMM_UNPACK();
  constexpr int kWaves = WM * WN;                             // computing wavefronts
  constexpr int kIssuers = NLOAD ? NLOAD : kWaves;            // wavefronts that issue DMA pieces
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;        // 16 x 16 output tiles per wavefront
  constexpr int kStageBytes = (BM + BN) * BK * 2;
  constexpr int kChunks = (BM + BN) / 8;                      // 1 KB DMA pieces (8 rows x 128 bytes) per k-tile
  constexpr int L = kChunks / kIssuers;                       // pieces per issuing wavefront per k-tile
  static_assert(kChunks % kIssuers == 0, "every issuing wavefront issues the same number of loads per k-tile (counted waits)");
  static_assert(BM % (16 * WM) == 0 && BN % (16 * WN) == 0, "wave tiles are multiples of 16");
  static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");
  constexpr bool kSplit = WST != 0;
  constexpr int kXStage = BM * BK * 2, kWStage = BN * BK * 2;
  constexpr int LX = (BM / 8) / kIssuers, LW = L - LX;        // split ring: a wavefront's pieces i < LX are X rows, the others W rows
  static_assert(!kSplit || (NLOAD == 0 && STAG == 0 && NSTAGE == 3 && WST == 2 && (BM / 8) % kIssuers == 0 && (BN / 8) % kIssuers == 0),
                "the split ring is written for three X stages + two W stages, every wavefront issuing whole shares of both");
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
  C4_CLK_DECL();
  C4_GSTAMP(0);
  C4_TL_BEGIN();

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = (wave / WM) % WN;            // (a loader wavefront's are unused)
  const int li = lane & 15, lg = lane >> 4;
  const bool is_loader = NLOAD != 0 && wave >= kWaves;
  const int issuer = NLOAD ? wave - kWaves : wave;            // this wavefront's index among the issuing ones

  // ---- which tile: blocks b and b + 8 share an XCD (round-robin dispatch; speed only).  Give each
  // XCD a rectangle of tiles so that its private L2 sees each X row block and W column block once.
  // (One straight-line formula, no division: the kernel's arguments are fetched by one scalar load at its top.)
  int tm, tn;
  tile_of_block(blockIdx.x, p.xcd_mask, p.xcd_shift, p.xn_log2, p.rm, p.rn, p.rn_magic, tm, tn);
  const int tm0 = tm * BM, tn0 = tn * BN;

  // ---- DMA source offsets (bytes from x / w) of this lane for its L pieces of a k-tile; a k-tile
  // later they are 128 bytes further.  Piece c < BM/8 is rows 8c..8c+7 of the X tile, else of the W tile.
  // Lane l of a piece fills LDS bytes 16 l..16 l + 15 = row l >> 3, slot l & 7 = k-group (l & 7) ^ (row & 7).
  uint32_t src_off[L];
  const int r8 = lane >> 3, slot = lane & 7;
#pragma unroll
  for (int i = 0; i < L; i++) {
    const int c = (issuer < 0 ? 0 : issuer) + kIssuers * i;
    if (c < BM / 8) {
      const int row = c * 8 + r8;
      int gr = tm0 + row;
      gr = gr < (int)p.M ? gr : (int)p.M - 1;                  // rows past the end re-read the last one (never stored)
      src_off[i] = (uint32_t)gr * p.ldx * 2u + (uint32_t)((slot ^ (row & 7)) * 16);
    } else {
      const int row = (c - BM / 8) * 8 + r8;
      src_off[i] = (uint32_t)(tn0 + row) * p.K * 2u + (uint32_t)((slot ^ (row & 7)) * 16);
    }
  }
  // buffer_load ... lds (MUBUF): descriptor + 32-bit per-lane offset + the k-tile's scalar offset.  (The
  // flat-encoded global_load_lds makes hipcc treat every later LDS wait as lgkmcnt(0); the MUBUF form
  // counts on vmcnt only, so the fragment reads below get counted lgkmcnt waits.)
  // num_records ends with the last row's K elements, not with its stride: X may be a column view of a wider tensor
  // (ldx > K), and the pieces issued past the last k-tile must then be clamped to zero by the bounds check instead of
  // reading beyond the allocation
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(((p.M - 1) * p.ldx + p.K) * 2u), 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)(p.N * p.K * 2u), 0x00020000);
  // Pieces are issued for every kt, also past the last k-tile (so that the count of outstanding pieces is the same in
  // every iteration: no branch around an issue, no tail cases in the counted waits) -- but those carry bit 31 in their
  // per-lane offset: beyond num_records (both operands are < 2 GiB), so the bounds check answers them with zeros
  // WITHOUT a memory access.  (The k-tile's own offset travels in the scalar offset, which the bounds check ignores:
  // without the flag the tail pieces really read two k-tiles past the operands' rows -- 10 % more L2 traffic, a wait
  // for them in front of the epilogue, and for the last row of a column view bytes beyond the allocation)
  const int KT = (int)p.K / BK;
  auto issue_one = [&](int kt, int i) __attribute__((always_inline)) {
    const int c = (issuer < 0 ? 0 : issuer) + kIssuers * i;
    uint8_t* dst = lds + (kt % NSTAGE) * kStageBytes + c * 1024;
    if (kSplit) dst = c < BM / 8 ? lds + (kt % NSTAGE) * kXStage + c * 1024 : lds + NSTAGE * kXStage + (kt % (kSplit ? WST : 1)) * kWStage + (c - BM / 8) * 1024;
    const uint32_t tail = kt >= KT ? 0x80000000u : 0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds((c < BM / 8) ? x_rsrc : w_rsrc, (__attribute__((address_space(3))) void*)dst, 16,
                                             (int)(src_off[i] | tail), kt * (BK * 2), 0, 0);
  };
  auto issue = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < L; i++) issue_one(kt, i);
  };
  if (NLOAD != 0 && is_loader) {
    // ---- a loader wavefront: k-tiles NSTAGE - 1 ahead of the computing wavefronts, one barrier per k-tile with them
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; s++) issue(s);
    for (int kt = 0; kt < KT; kt++) {
      wait_vmcnt<(NSTAGE - 2) * L>();                           // my pieces of k-tile kt have landed
      __builtin_amdgcn_s_barrier();                             // ... everybody's have; everybody has left buffer (kt - 1) % NSTAGE
      issue(kt + NSTAGE - 1);
    }
    wait_vmcnt<0>();                                            // nothing may land in LDS that was given away
    return;
  }

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; a++)
#pragma unroll
    for (int b = 0; b < TM; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets inside a stage: row = tile base (a multiple of 16) + li, so row & 7 == li & 7
  const uint32_t frag_off0 = (uint32_t)li * 128u + (uint32_t)(((0 + lg) ^ (li & 7)) * 16);
  const uint32_t frag_off1 = (uint32_t)li * 128u + (uint32_t)(((4 + lg) ^ (li & 7)) * 16);
  const uint32_t x_base = (uint32_t)(wm * (BM / WM)) * 128u;
  const uint32_t w_base = (uint32_t)(wn * (BN / WN)) * 128u;   // inside the stage's W part

  // this wavefront's biases: requested now, used after the last k-tile (the epilogue used to open with this round trip)
  f32x4 bias_v[TN];
#pragma unroll
  for (int a = 0; a < TN; a++) bias_v[a] = *reinterpret_cast<const f32x4*>(p.bias + tn0 + wn * (BN / WN) + a * 16 + 4 * lg);
  if (NLOAD == 0 && !kSplit) {
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; s++) issue(s);
  }
  if (kSplit) {   // X(0), W(0), X(1): in that order, so that "all but my youngest LX pieces" covers both operands of k-tile 0
#pragma unroll
    for (int i = 0; i < LX; i++) issue_one(0, i);
#pragma unroll
    for (int i = LX; i < L; i++) issue_one(0, i);
#pragma unroll
    for (int i = 0; i < LX; i++) issue_one(1, i);
  }

  C4_GSTAMP(1);
  C4_CLK_BEGIN();
  static_assert(!STAG || (NLOAD == 0 && WM * WN == 8 && NSTAGE >= 3), "the stagger is written for 8 computing wavefronts that issue their own pieces");
  bf16x8 afr[2][TN], bfr[2][TM];
  // One k-tile.  kLate: a wavefront that runs half a k-tile behind (STAG).
  auto body = [&](int kt, auto late_c) __attribute__((always_inline)) {
    constexpr bool kLate = decltype(late_c)::value;
    // k-tile kt must have landed; the NSTAGE - 2 younger ones stay in flight (the same count in every iteration:
    // see issue_one for the pieces past the last k-tile)
    if (kSplit) wait_vmcnt<LX>();                               // (split ring: only the X pieces of k-tile kt + 1 are younger than W(kt))
    else if (NLOAD == 0) wait_vmcnt<(NSTAGE - 2) * L>();
    __builtin_amdgcn_s_barrier();                               // everybody's pieces of kt landed; everybody left buffer (kt - 1) % NSTAGE
#ifdef C4_GEMM_CLOCK
    if (kt == 0) C4_GSTAMP(2);
#endif
    // The DMA pieces of k-tile kt + NSTAGE - 1 are NOT issued here in one burst (a wavefront would spend
    // hundreds of cycles queueing 1 KB requests before its first MFMA): they are handed out between the
    // MFMA groups below, one or two at a time, so the address pipe works under the matrix pipe.  With two
    // stages they all go out in the first half of the k-tile (the second half gives them time to land),
    // with three or more over the whole k-tile.
    const int lkt = kt + NSTAGE - 1;
    constexpr int kLoadSteps = NSTAGE == 2 ? TN : 2 * TN;
    constexpr int kLoadsPerStep = (L + kLoadSteps - 1) / kLoadSteps;
    const uint8_t* stx = kSplit ? lds + (kt % NSTAGE) * kXStage : lds + (kt % NSTAGE) * kStageBytes;
    const uint8_t* stw = kSplit ? lds + NSTAGE * kXStage + (kt % (kSplit ? WST : 1)) * kWStage : stx + kXStage;
    // Fragment reads of the first 32-deep half are requested up front (activations first: the first
    // MFMAs need all TM of them and one weight fragment); the second half's reads are issued between
    // the first half's MFMAs, so that at most ~14 LDS reads are outstanding (the counter holds 15)
    // and the MFMAs of a half never wait for more than the fragments they use.
    auto rd_x = [&](int kk, int b) __attribute__((always_inline)) {
      bfr[kk][b] = *reinterpret_cast<const bf16x8*>(stx + x_base + (kk ? frag_off1 : frag_off0) + b * 2048);
    };
    auto rd_w = [&](int kk, int a) __attribute__((always_inline)) {
      afr[kk][a] = *reinterpret_cast<const bf16x8*>(stw + w_base + (kk ? frag_off1 : frag_off0) + a * 2048);
    };
    auto issue_step = [&](int step) __attribute__((always_inline)) {   // the DMA pieces handed out behind MFMA group `step` of 2 TN
      if (NLOAD == 0 && (NSTAGE >= 3 || step < TN)) {
#pragma unroll
        for (int j = step * kLoadsPerStep; j < (step + 1) * kLoadsPerStep && j < L; j++) {
          if (!kSplit) issue_one(lkt, j);
          else if (j < LW) issue_one(kt + WST - 1, LX + j);     // W of the next k-tile first
          else issue_one(kt + NSTAGE - 1, j - LW);              // then X two k-tiles ahead
        }
      }
    };
#pragma unroll
    for (int b = 0; b < TM; b++) rd_x(0, b);
#pragma unroll
    for (int a = 0; a < TN; a++) rd_w(0, a);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int kPerStep = (TN + TM + TN - 1) / TN;           // second-half reads issued behind each weight row's MFMAs
    if (kLate) {
      // the deferred second half of k-tile kt - 1 (fragments in afr[1] / bfr[1]) runs while the reads above travel
#pragma unroll
      for (int a = 0; a < TN; a++) {
        if (kt > 0) {
#pragma unroll
          for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[1][a], bfr[1][b], acc[a][b], 0, 0, 0);
        }
        issue_step(a);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int a = 0; a < TN; a++) {
#pragma unroll
      for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0][a], bfr[0][b], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int r = a * kPerStep; r < (a + 1) * kPerStep && r < TN + TM; r++) {
        if (r < TM) rd_x(1, r); else rd_w(1, r - TM);
      }
      issue_step(kLate ? TN + a : a);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kLate) {
      // the second half's fragments must be in registers before the next barrier lets a DMA piece into this stage
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int a = 0; a < TN; a++) {
#pragma unroll
        for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[1][a], bfr[1][b], acc[a][b], 0, 0, 0);
        issue_step(TN + a);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if (STAG && wave >= kWaves / 2) {
    for (int kt = 0; kt < KT; kt++) body(kt, std::true_type{});
#pragma unroll
    for (int a = 0; a < TN; a++)                                // the last k-tile's deferred half
#pragma unroll
      for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[1][a], bfr[1][b], acc[a][b], 0, 0, 0);
  } else {
    for (int kt = 0; kt < KT; kt++) body(kt, std::false_type{});
  }

  C4_CLK_END();
  C4_GSTAMP(3);
  wait_vmcnt<0>();   // the pieces issued past the last k-tile must not land in an LDS allocation that was given away
  C4_GSTAMP(4);
  // ---- epilogue: lane holds features n0 + 4 lg + {0..3} of board m (C/D map: row = 4 (lane >> 4) + reg, col = lane & 15)
  // hipcc's wait-count pass does not see through the inline-asm wait above: it still believes the bias loads of the
  // prologue to be in flight and would wait for them (in order, i.e. also for the STORES issued meanwhile) in the
  // middle of the epilogue.  Using the values here puts its waits where everything has long arrived.
#pragma unroll
  for (int a = 0; a < TN; a++) asm volatile("" ::"v"(bias_v[a]));
  C4_GSTAMP(5);
  store_wave_tile<TM, TN>(acc, bias_v, p, tm0 + wm * (BM / WM), tn0 + wn * (BN / WN), li, lg);
  C4_GSTAMP(6);
#ifdef C4_GEMM_CLOCK
  if (C4_CLK_PHASES) wait_vmcnt<0>();
  C4_GSTAMP(7);
  C4_CLK_FLUSH();
#endif
  C4_TL_END(2, p.y);
}

// ------------------------------------------------------------------------------------------
// The same GEMM with 32-deep k-tiles (64-byte LDS rows), for tiles too big for three 64-deep stages:
// 256 x 192 pulls 30 % fewer operand bytes out of L2 per flop than 128 x 192 -- and L2 -> LDS bytes, not
// MFMA issue, are what bounds these layers (profiles/r03_evaluator_pmc.json) -- but a 64-deep stage of
// it is 56 KB.  Differences from the kernel above:
//   * a DMA piece is 16 rows x 64 bytes; slot g of row r lives at g ^ T[(r >> 2) & 3], T = {0, 2, 3, 1}
//     (conflict-free for every ds_read_b128 lane group: rows 4 apart share a bank row);
//   * one MFMA k-step per k-tile, so the fragment reads are skewed by a k-tile instead of by half of
//     one: after the barrier that publishes k-tile kt its fragments are requested into one register set
//     while the MFMAs of k-tile kt - 1 run from the other;
//   * the pieces of a k-tile need not divide evenly over the wavefronts (28 over 8): the first few
//     wavefronts issue one more, and each waits for its own count.
// Every output element still sees the same chain (k ascending, 32 at a time): same bits as above.
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int NSTAGE, int MINW = 1>
__global__ __launch_bounds__(64 * WM * WN, MINW) void c4_head_gemm32_kernel(C4_GEMM_ARGS) {
  C4_GEMM_UNPACK();
  constexpr int BKT = 32;
  constexpr int kWaves = WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int kStageBytes = (BM + BN) * BKT * 2;
  constexpr int kChunks = (BM + BN) / 16;                     // 1 KB pieces (16 rows x 64 bytes) per k-tile
  constexpr int L = (kChunks + kWaves - 1) / kWaves;          // pieces of the wavefronts that issue the most
  constexpr int R = kChunks % kWaves;                         // wavefronts 0 .. R - 1 issue L, the others L - 1 (R == 0: all L)
  static_assert(BM % (16 * WM) == 0 && BN % (16 * WN) == 0 && NSTAGE >= 3 && NSTAGE <= 5, "tile / ring");
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int li = lane & 15, lg = lane >> 4;
  const bool short_wave = R != 0 && wave >= R;                // issues L - 1 pieces per k-tile

  int tm, tn;
  tile_of_block(blockIdx.x, p.xcd_mask, p.xcd_shift, p.xn_log2, p.rm, p.rn, p.rn_magic, tm, tn);
  const int tm0 = tm * BM, tn0 = tn * BN;

  auto swz = [](int row) __attribute__((always_inline)) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; };   // T = {0, 2, 3, 1}
  uint32_t src_off[L];
  const int r16 = lane >> 2, slot = lane & 3;
#pragma unroll
  for (int i = 0; i < L; i++) {
    int c = wave + kWaves * i;
    c = c < kChunks ? c : kChunks - 1;                        // a short wavefront's surplus entry is never issued
    if (c < BM / 16) {
      const int row = c * 16 + r16;
      int gr = tm0 + row;
      gr = gr < (int)p.M ? gr : (int)p.M - 1;
      src_off[i] = (uint32_t)gr * p.ldx * 2u + (uint32_t)((slot ^ swz(row)) * 16);
    } else {
      const int row = (c - BM / 16) * 16 + r16;
      src_off[i] = (uint32_t)(tn0 + row) * p.K * 2u + (uint32_t)((slot ^ swz(row)) * 16);
    }
  }
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(((p.M - 1) * p.ldx + p.K) * 2u), 0x00020000);   // see above
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)(p.N * p.K * 2u), 0x00020000);
  const int KT = (int)p.K / BKT;
  auto issue_one = [&](int kt, int i) __attribute__((always_inline)) {
    if (i == L - 1 && short_wave) return;
    uint8_t* st = lds + (kt % NSTAGE) * kStageBytes;
    const int c = wave + kWaves * i;
    const uint32_t tail = kt >= KT ? 0x80000000u : 0u;          // past the last k-tile: out of bounds, answered with zeros without a memory access
    __builtin_amdgcn_raw_ptr_buffer_load_lds((c < BM / 16) ? x_rsrc : w_rsrc, (__attribute__((address_space(3))) void*)(st + c * 1024), 16,
                                             (int)(src_off[i] | tail), kt * (BKT * 2), 0, 0);
  };
  auto wait_tile = [&]() __attribute__((always_inline)) {     // all but the (NSTAGE - 2) youngest k-tiles of THIS wavefront have landed
    if (short_wave) wait_vmcnt<(NSTAGE - 2) * (L - 1)>(); else wait_vmcnt<(NSTAGE - 2) * L>();
    // the fragment reads of the previous k-tile (consumed only in THIS iteration) must have returned before the
    // barrier lets anybody's DMA overwrite their buffer; they were issued a whole MFMA group ago
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; a++)
#pragma unroll
    for (int b = 0; b < TM; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const uint32_t frag_off = (uint32_t)li * 64u + (uint32_t)((lg ^ swz(li)) * 16);   // row = tile base (multiple of 16) + li
  const uint32_t x_base = (uint32_t)(wm * (BM / WM)) * 64u;
  const uint32_t w_base = (uint32_t)(BM * BKT * 2) + (uint32_t)(wn * (BN / WN)) * 64u;

  f32x4 bias_v[TN];
#pragma unroll
  for (int a = 0; a < TN; a++) bias_v[a] = *reinterpret_cast<const f32x4*>(p.bias + tn0 + wn * (BN / WN) + a * 16 + 4 * lg);
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; s++) {
#pragma unroll
    for (int i = 0; i < L; i++) issue_one(s, i);
  }

  bf16x8 afr[2][TN], bfr[2][TM];
  auto read_frags = [&](int kt, int f) __attribute__((always_inline)) {
    const uint8_t* st = lds + (kt % NSTAGE) * kStageBytes;
#pragma unroll
    for (int b = 0; b < TM; b++) bfr[f][b] = *reinterpret_cast<const bf16x8*>(st + x_base + frag_off + b * 1024);
#pragma unroll
    for (int a = 0; a < TN; a++) afr[f][a] = *reinterpret_cast<const bf16x8*>(st + w_base + frag_off + a * 1024);
  };
  constexpr int kLoadsPerStep = (L + TN - 1) / TN;
  auto step = [&](int kt, int fcur, int fprev, bool do_mfma) __attribute__((always_inline)) {
    wait_tile();
    __builtin_amdgcn_s_barrier();                               // k-tile kt is in LDS for everybody; buffer (kt - 1) % NSTAGE is free
    read_frags(kt, fcur);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < TN; a++) {
      if (do_mfma) {
#pragma unroll
        for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[fprev][a], bfr[fprev][b], acc[a][b], 0, 0, 0);
      }
#pragma unroll
      for (int j = a * kLoadsPerStep; j < (a + 1) * kLoadsPerStep && j < L; j++) issue_one(kt + NSTAGE - 1, j);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  step(0, 0, 1, false);
  for (int kt = 1; kt + 1 < KT; kt += 2) {                      // KT is even (K % 64 == 0)
    step(kt, 1, 0, true);
    step(kt + 1, 0, 1, true);
  }
  step(KT - 1, 1, 0, true);
#pragma unroll
  for (int a = 0; a < TN; a++)                                  // the last k-tile's MFMAs
#pragma unroll
    for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[1][a], bfr[1][b], acc[a][b], 0, 0, 0);
  wait_vmcnt<0>();

#pragma unroll
  for (int a = 0; a < TN; a++) asm volatile("" ::"v"(bias_v[a]));   // see c4_head_gemm_kernel
  store_wave_tile<TM, TN>(acc, bias_v, p, tm0 + wm * (BM / WM), tn0 + wn * (BN / WN), li, lg);
}

// ------------------------------------------------------------------------------------------
// Round 5: the 256 x 192 tile on FOUR wavefronts, one per SIMD, 128 x 96 per wavefront, operands staged THROUGH REGISTERS.
// For the 64-channel net (K = 2 688: BASELINE configs 4 / 5), whose rounds are CU-time-bound and whose GEMMs are 56 % of that time.
// Why this shape: per 64-deep k-tile the 8-wavefront 256 x 192 kernel (64 x 96 per wavefront, config 7 / 48) moves 8 x 20 KB of
// fragments out of LDS + 56 KB of DMA into it and takes ~2 600 cycles where its MFMAs need 1 536; with 128 x 96 per wavefront the
// fragments are 4 x 28 KB -- and what killed round 4's 4-wavefront form (config 49: 3 400 cycles) was the ISSUE of its DMA: a
// wavefront that issues its own 14 pieces per k-tile is held 60-185 cycles per piece (MI355X_MICROARCH.md, LDS-DMA issue
// cost) in the same in-order stream as its 96 MFMAs.  A global_load_dwordx4 costs its issue slot only, and a wavefront with 512
// registers has room for the k-tile in flight (14 x 4 registers): load k-tile kt + 2 into registers, store k-tile kt + 1 from
// registers into the ring's other stage (ds_write_b128, the XOR swizzle on the LDS address), multiply k-tile kt out of this one.
// One barrier per k-tile: in front of it every wavefront has finished reading the stage that is written next and its own
// stores of the stage that is read next have landed.  Same LDS image, same fragment reads, same MFMA order per accumulator as
// every other configuration: same bits.
// ------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256, 1) void c4_head_gemm4_kernel(C4_GEMM_ARGS) {
  C4_GEMM_UNPACK();
  constexpr int WM = 2, WN = 2;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;          // 8 x 6 output tiles of 16 x 16 per wavefront
  constexpr int kStageBytes = (BM + BN) * BK * 2;              // 56 KB
  constexpr int kChunks = (BM + BN) / 8;                       // 1 KB pieces (8 rows x 128 bytes) per k-tile
  constexpr int L = kChunks / 4;                               // pieces per wavefront per k-tile
  static_assert(kChunks % 4 == 0 && (BM / 8) % 4 == 0, "every wavefront moves whole shares of both operands");
  constexpr int LX = (BM / 8) / 4;                             // a wavefront's pieces i < LX are X rows, the others W rows
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int li = lane & 15, lg = lane >> 4;
  int tm, tn;
  tile_of_block(blockIdx.x, p.xcd_mask, p.xcd_shift, p.xn_log2, p.rm, p.rn, p.rn_magic, tm, tn);
  const int tm0 = tm * BM, tn0 = tn * BN;

  // piece i of this wavefront: rows 8 c .. 8 c + 7 of the X tile (c < BM / 8) or of the W tile; lane l moves the 16 bytes of
  // row l >> 3, k-group l & 7, to LDS slot (l & 7) ^ (row & 7) of that row
  uint32_t src_off[L], dst_off[L];
  const int r8 = lane >> 3, slot = lane & 7;
#pragma unroll
  for (int i = 0; i < L; i++) {
    const bool is_x = i < LX;
    const int c = is_x ? wave + 4 * i : wave + 4 * (i - LX);
    const int row = c * 8 + r8;
    if (is_x) {
      int gr = tm0 + row;
      gr = gr < (int)p.M ? gr : (int)p.M - 1;                  // rows past the end re-read the last one (never stored)
      src_off[i] = (uint32_t)gr * p.ldx * 2u + (uint32_t)(slot * 16);
      dst_off[i] = (uint32_t)row * 128u + (uint32_t)((slot ^ (row & 7)) * 16);
    } else {
      src_off[i] = (uint32_t)(tn0 + row) * p.K * 2u + (uint32_t)(slot * 16);
      dst_off[i] = (uint32_t)(BM + row) * 128u + (uint32_t)((slot ^ (row & 7)) * 16);
    }
  }
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(((p.M - 1) * p.ldx + p.K) * 2u), 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)(p.N * p.K * 2u), 0x00020000);
  const int KT = (int)p.K / BK;
  u32x4 stg[L];                                                // the k-tile in flight between global memory and LDS
  auto load_tile = [&](int kt) __attribute__((always_inline)) {
    // past the last k-tile: bit 31 in the per-lane offset puts the request beyond num_records -> zeros without a memory access
    const uint32_t tail = kt >= KT ? 0x80000000u : 0u;
#pragma unroll
    for (int i = 0; i < L; i++)
      stg[i] = __builtin_amdgcn_raw_buffer_load_b128(i < LX ? x_rsrc : w_rsrc, (int)(src_off[i] | tail), kt * (BK * 2), 0);
  };
  auto store_tile = [&](int kt) __attribute__((always_inline)) {
    uint8_t* st = lds + (kt & 1) * kStageBytes;
#pragma unroll
    for (int i = 0; i < L; i++) *reinterpret_cast<u32x4*>(st + dst_off[i]) = stg[i];
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; a++)
#pragma unroll
    for (int b = 0; b < TM; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t frag_off0 = (uint32_t)li * 128u + (uint32_t)(((0 + lg) ^ (li & 7)) * 16);
  const uint32_t frag_off1 = (uint32_t)li * 128u + (uint32_t)(((4 + lg) ^ (li & 7)) * 16);
  const uint32_t x_base = (uint32_t)(wm * (BM / WM)) * 128u;
  const uint32_t w_base = (uint32_t)(BM + wn * (BN / WN)) * 128u;
  f32x4 bias_v[TN];
#pragma unroll
  for (int a = 0; a < TN; a++) bias_v[a] = *reinterpret_cast<const f32x4*>(p.bias + tn0 + wn * (BN / WN) + a * 16 + 4 * lg);

  load_tile(0);
  store_tile(0);                                               // (waits for the loads: the one exposed round trip of the kernel)
  load_tile(1);
  bf16x8 afr[2][TN], bfr[2][TM];
  for (int kt = 0; kt < KT; kt++) {
    // my stores of k-tile kt have landed (and my fragment reads of k-tile kt - 1 are long consumed); then everybody's have
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const uint8_t* st = lds + (kt & 1) * kStageBytes;
    auto rd_x = [&](int kk, int b) __attribute__((always_inline)) {
      bfr[kk][b] = *reinterpret_cast<const bf16x8*>(st + x_base + (kk ? frag_off1 : frag_off0) + b * 2048);
    };
    auto rd_w = [&](int kk, int a) __attribute__((always_inline)) {
      afr[kk][a] = *reinterpret_cast<const bf16x8*>(st + w_base + (kk ? frag_off1 : frag_off0) + a * 2048);
    };
    // first half's fragments leave first; the staging work of the NEXT k-tiles (14 ds_write_b128 of k-tile kt + 1 into the other
    // stage, 14 global loads of k-tile kt + 2) is issued while they travel
#pragma unroll
    for (int b = 0; b < TM; b++) rd_x(0, b);
#pragma unroll
    for (int a = 0; a < TN; a++) rd_w(0, a);
    __builtin_amdgcn_sched_barrier(0);
    store_tile(kt + 1);                                        // k-tile kt + 1 (in registers since the previous iteration) -> stage (kt + 1) & 1
    load_tile(kt + 2);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int kPerStep = (TN + TM + TN - 1) / TN;          // second-half reads issued behind each weight row's MFMAs
#pragma unroll
    for (int a = 0; a < TN; a++) {
#pragma unroll
      for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0][a], bfr[0][b], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int r = a * kPerStep; r < (a + 1) * kPerStep && r < TN + TM; r++) {
        if (r < TM) rd_x(1, r); else rd_w(1, r - TM);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < TN; a++) {
#pragma unroll
      for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[1][a], bfr[1][b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int a = 0; a < TN; a++) asm volatile("" ::"v"(bias_v[a]));   // see c4_head_gemm_kernel
  store_wave_tile<TM, TN>(acc, bias_v, p, tm0 + wm * (BM / WM), tn0 + wn * (BN / WN), li, lg);
}

// Tile order: of the factorizations xm * xn = 8 that divide the tile grid, the one whose rectangle pulls the fewest
// operand rows into an XCD's L2; none -> plain row-major order.
inline uint32_t set_tile_order(GemmParams& p, uint32_t tiles_m, uint32_t tiles_n, uint32_t bm, uint32_t bn) {
  p.xcd_mask = 0; p.xcd_shift = 0; p.xn_log2 = 0; p.rm = tiles_m; p.rn = tiles_n;
  uint64_t best = ~0ull;
  for (uint32_t xm = 1, l2 = 3; xm <= 8; xm *= 2, l2--) {
    const uint32_t xn = 8 / xm;
    if (tiles_m % xm || tiles_n % xn) continue;
    const uint64_t rows = (uint64_t)(tiles_m / xm) * bm + (uint64_t)(tiles_n / xn) * bn;
    if (rows < best) { best = rows; p.xcd_mask = 7; p.xcd_shift = 3; p.xn_log2 = l2; p.rm = tiles_m / xm; p.rn = tiles_n / xn; }
  }
  p.rn_magic = p.rn == 1 ? 0u : (uint32_t)((1ull << 32) / p.rn + 1);   // exact for idx * rn < 2^32; rn == 1: see tile_of_block
  return tiles_m * tiles_n;
}

template <int BM, int BN, typename K>
int launch_common(K k, GemmParams p, int threads, int lds_bytes, hipStream_t stream, int device) {
  if (lds_bytes > 64 * 1024) {
    const hipError_t e = c4host::opt_in_lds((const void*)k, lds_bytes, device);
    if (e != hipSuccess) return c4host::fail(C4_ERR_HIP, std::string("c4_linear_bf16: LDS opt-in: ") + hipGetErrorString(e));
  }
  const uint32_t tiles = set_tile_order(p, (p.M + BM - 1) / BM, p.N / BN, BM, BN);
  if (p.rm > 0xFFFFu || p.rn > 0xFFFFu)   // rm and rn travel as 16-bit halves of one kernel argument
    return c4host::fail(C4_ERR_BAD_ARG, "c4_linear_bf16: more than 65 535 tiles along one dimension of an XCD's rectangle (m too large for this tile)");
  k<<<dim3(tiles), dim3(threads), lds_bytes, stream>>>(p.x, p.w, p.bias, p.y, p.M, p.N | (p.K << 16), p.ldx | (p.ldy << 16),
                                                         (p.relu ? 1u : 0u) | (p.xn_log2 << 8) | (p.xcd_mask << 16) | (p.xcd_shift << 24), p.rm | (p.rn << 16), p.rn_magic);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return c4host::fail(C4_ERR_HIP, std::string("c4_linear_bf16 launch: ") + hipGetErrorString(e));
  return C4_OK;
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int MINW = 1>
int launch_gemm32(GemmParams p, hipStream_t stream, int device) {
  return launch_common<BM, BN>(c4_head_gemm32_kernel<BM, BN, WM, WN, NSTAGE, MINW>, p, 64 * WM * WN, NSTAGE * (BM + BN) * 32 * 2, stream, device);
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int MINW, int NLOAD = 0, int STAG = 0, int WST = 0>
int launch_gemm(GemmParams p, hipStream_t stream, int device) {
  return launch_common<BM, BN>(c4_head_gemm_kernel<BM, BN, WM, WN, NSTAGE, MINW, NLOAD, STAG, WST>, p, 64 * (WM * WN + NLOAD),
                               WST ? (NSTAGE * BM + WST * BN) * BK * 2 : NSTAGE * (BM + BN) * BK * 2, stream, device);
}

}  // namespace

#ifdef C4_GEMM_CLOCK
// diagnostic builds only: mean shader clock (GHz) and main-loop duration (us) of the c4_head_gemm_kernel workgroups since the last reset
extern "C" int c4_debug_gemm_clock(double* ghz, double* loop_us, uint64_t* n_workgroups, int reset) {
  unsigned long long h[16];
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(c4_gemm_clk), sizeof h) != hipSuccess) return C4_ERR_HIP;
  if (ghz) *ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
  if (loop_us) *loop_us = h[2] ? (double)h[1] / (double)h[2] * 0.01 : 0.0;
  if (n_workgroups) *n_workgroups = h[2];
  if (reset) { unsigned long long z[16] = {0}; z[12] = ~0ull; if (hipMemcpyToSymbol(HIP_SYMBOL(c4_gemm_clk), z, sizeof z) != hipSuccess) return C4_ERR_HIP; }
  return C4_OK;
}
// mean time per phase (us, 7 phases between the 8 stamps) of the workgroups since the last reset, and the span from the earliest
// workgroup's entry to the latest one's exit (us; meaningful for ONE launch)
extern "C" int c4_debug_gemm_phases(double* phase_us, double* span_us, int reset) {
  unsigned long long h[16];
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(c4_gemm_clk), sizeof h) != hipSuccess) return C4_ERR_HIP;
  for (int i = 0; i < 7; i++) phase_us[i] = h[2] ? (double)h[4 + i] / (double)h[2] * 0.01 : 0.0;
  if (span_us) *span_us = h[13] > h[12] ? (double)(h[13] - h[12]) * 0.01 : 0.0;
  if (reset) { unsigned long long z[16] = {0}; z[12] = ~0ull; if (hipMemcpyToSymbol(HIP_SYMBOL(c4_gemm_clk), z, sizeof z) != hipSuccess) return C4_ERR_HIP; }
  return C4_OK;
}
#endif

C4_TL_SETTER(c4_debug_timeline_gemm)

// The block -> tile map of a launch, computed on the host with the kernels' own formula (tile_of_block): tiles_out[2 b] = tm,
// tiles_out[2 b + 1] = tn of block b.  No device is touched: the CPU suite walks every tile grid c4_linear_bf16 accepts and
// checks that the map is a bijection onto the grid (tests/test_abi_and_results.py).
extern "C" int c4_linear_bf16_tile_map(uint32_t m, uint32_t n, uint32_t bm, uint32_t bn, uint32_t* tiles_out, uint32_t cap_blocks, uint32_t* n_blocks) {
  if (!n_blocks || bm == 0 || bn == 0 || m == 0 || n == 0 || n % bn) return c4host::fail(C4_ERR_BAD_ARG, "c4_linear_bf16_tile_map: bad argument");
  GemmParams p{};
  p.M = m; p.N = n;
  const uint32_t tiles = set_tile_order(p, (m + bm - 1) / bm, n / bn, bm, bn);
  *n_blocks = tiles;
  if (p.rm > 0xFFFFu || p.rn > 0xFFFFu) return c4host::fail(C4_ERR_BAD_ARG, "c4_linear_bf16_tile_map: more than 65 535 tiles along one dimension");
  if (!tiles_out) return C4_OK;
  if (cap_blocks < tiles) return c4host::fail(C4_ERR_BAD_ARG, "c4_linear_bf16_tile_map: output too small");
  for (uint32_t b = 0; b < tiles; b++) {
    int tm, tn;
    tile_of_block(b, p.xcd_mask, p.xcd_shift, p.xn_log2, p.rm, p.rn, p.rn_magic, tm, tn);
    tiles_out[2 * b] = (uint32_t)tm; tiles_out[2 * b + 1] = (uint32_t)tn;
  }
  return C4_OK;
}

extern "C" int c4_linear_bf16(const void* x_dev, const void* w_dev, const float* bias_dev, void* y_dev, uint32_t m, uint32_t n,
                              uint32_t k, uint32_t ldx, uint32_t ldy, uint32_t relu, uint32_t config, void* stream) {
  if (!x_dev || !w_dev || !bias_dev || !y_dev) return c4host::fail(C4_ERR_BAD_ARG, "c4_linear_bf16: null argument");
  if (k == 0 || k % BK) return c4host::fail(C4_ERR_BAD_ARG, "c4_linear_bf16: K must be a positive multiple of 64");
  if (n == 0 || n % 192) return c4host::fail(C4_ERR_BAD_ARG, "c4_linear_bf16: N must be a positive multiple of 192 (42 x C features, C a multiple of 32)");
  if (ldx < k || ldy < n || ldx % 8 || ldy % 8 || ((uintptr_t)y_dev & 15) || ((uintptr_t)x_dev & 15))
    return c4host::fail(C4_ERR_BAD_ARG, "c4_linear_bf16: row strides must cover a row and keep rows 16-byte aligned (x and y)");
  if ((uint64_t)m * ldx * 2 >= (1ull << 31) || (uint64_t)n * k * 2 >= (1ull << 31)) return c4host::fail(C4_ERR_BAD_ARG, "c4_linear_bf16: operands are addressed with 31-bit byte offsets (< 2 GiB each)");
  if (n >= 65536 || k >= 65536 || ldx >= 65536 || ldy >= 65536) return c4host::fail(C4_ERR_BAD_ARG, "c4_linear_bf16: n, k and the row strides must be below 65 536 (packed kernel arguments)");
  if (m == 0) return C4_OK;
  const int device = c4host::stream_device((hipStream_t)stream);
  c4host::DeviceGuard guard(device);
  if (guard.error() != hipSuccess) return c4host::fail(C4_ERR_HIP, std::string("c4_linear_bf16: hipSetDevice: ") + hipGetErrorString(guard.error()));
  GemmParams p{(const uint16_t*)x_dev, (const uint16_t*)w_dev, bias_dev, (uint16_t*)y_dev, m, n, k, ldx, ldy, relu, 0, 0, 0, 0, 0, 0};
  hipStream_t st = (hipStream_t)stream;
  // Automatic choice, measured on MI355X IN THE BENCH (two sessions' kernels sharing the chip), not alone
  // (profiles/r03_gemm_configs.txt; every configuration gives the same bits): the 128 x 192 tile with
  // 8 wavefronts and a 3-deep ring wins for the wide AND the narrow layers -- it pulls the fewest operand
  // bytes out of L2 per flop of all tiles that still give every layer >= 112 workgroups, and the CUs its
  // narrow-layer launches leave free go to the other session's kernels.  Finer tilings that look better
  // alone (128 x 96: 13.5 vs 16.6 us for a 2048 x 1344 x 1344 layer) lose 8-13 % of games/s in the mix.
  // Small batches (callback mode, tails, jobs of a few hundred games) want many small tiles: there a
  // layer is a latency chain, not CU-time (alone, us: M = 512 cfg 9 8.0 / 7.5 for the wide / narrow
  // layer against cfg 11's 16.4 / 15.5; M = 1024 cfg 10 12.3 wide, cfg 9 7.9 narrow).
  // Round 3, alone, us (tools/gemm_probe.py, wide / narrow): M = 256 cfg 27 6.6 / 6.2 (cfg 9 7.7 / 7.3); M = 512 cfg 9
  // 8.2 wide, cfg 27 6.6 narrow; M = 768 cfg 23 10.0 wide (cfg 10 11.7), cfg 9 7.5 narrow.  Between 1 025 and 1 728
  // rows a caller that has the chip to itself asks for cfg 23 (96 x 96, 72 KB: <= 2 x 256 workgroups, two per
  // CU: 18.0 / 12.2 against 19.4 / 12.5 at 1 700 rows; c4a0_amd/nn.py latency_mode).
  if (config == 0) {
    const bool wide = n > k;
    // Above 1 024 rows every layer runs the 128 x 192 tile on 8 wavefronts (config 11).  (Mid round 4 the 2F-wide layer ran it on
    // FOUR wavefronts, config 6 -- 64 x 96 each, one per SIMD, 29 % fewer fragment bytes out of LDS: slower alone, 20.3 against
    // 19.8 us at 2 048 rows, but +3.6 % games/s in the bench of that moment because the other session's small step-kernel
    // workgroups fitted beside it.  Since the output layers and the step are ONE launch of 6-wavefront workgroups that
    // advantage is gone: same box, 11 for every layer 29.6-30.2 k games/s against 29.4-29.8 k, profiles/r04_gemm_configs.txt (4).)
    // ... and up to 1 024 rows, where a layer is a latency chain, the wave-specialised forms of round 3's small tiles (two or four
    // wavefronts that only issue the DMA pieces: 3-15 % less time alone, tools/gemm_small_ab.sh in profiles/r04_gemm_configs.txt)
    // (Round 5, ADVICE r4: those forms were chosen with each GEMM ALONE; in the paired graph -- two sessions sharing the chip, 2 x 512 and
    // 2 x 1 024 rows -- round 3's tiles without loader wavefronts are 5 % FASTER, 11 928 against 11 345 and 20 960 against 19 850 games/s,
    // profiles/r05_small_rows_ab.txt: a 6- or 8-wavefront workgroup leaves no room beside the other session's output + step launch.  The
    // automatic choice is therefore round 3's table again; a caller that has the chip to itself asks for 41 / 42 / 44 / 43 by number,
    // c4a0_amd/nn.py latency_mode.)
    config = m <= 384 ? 27 : m <= 640 ? (wide ? 9 : 27) : m <= 896 ? (wide ? 23 : 9) : m <= 1024 ? (wide ? 10 : 9) : 11;
    // The 64-channel net (K = 2 688, N = 5 376 / 2 688: four times the flops per layer): there the 256 x 192 tile
    // (64 x 96 per wavefront, 2-deep ring, 112 KB) pays -- BASELINE config 4 (2 x 2 048 rows) 862 -> 925-939 games/s,
    // config 5's per-GPU share (2 x 4 096 rows) 3 271 -> 3 671 (hipBLASLt: 911-919 / 3 736), profiles/r03_gemm_configs.txt
    // Below that its layers want one size up from the 32-channel table (alone, us, wide / narrow: M = 256 cfg 9 13.8 /
    // cfg 27 11.8; M = 512 cfg 10 21.4 / cfg 9 13.6; M = 1 024 cfg 11 30.9 / cfg 10 21.0).
    if (k >= 2048) config = m <= 384 ? (wide ? 9 : 27) : m <= 640 ? (wide ? 10 : 9) : m <= 1024 ? (wide ? 11 : 10) : 7;
  }
  switch (config) {
    case 1: return launch_gemm<128, 192, 2, 2, 2, 2>(p, st, device);   // 4 wavefronts (64 x 96 each), 80 KB: two workgroups per CU
    case 2: return launch_gemm<128, 192, 2, 4, 4, 1>(p, st, device);   // 8 wavefronts (64 x 48), 4-deep ring, the whole LDS
    case 3: return launch_gemm<256, 192, 2, 4, 2, 1>(p, st, device);   // 8 wavefronts (128 x 48), 112 KB
    case 4: return launch_gemm<128, 96, 2, 2, 2, 2>(p, st, device);    // 4 wavefronts (64 x 48), 56 KB
    case 5: return launch_gemm<64, 192, 1, 4, 2, 2>(p, st, device);    // 4 wavefronts (64 x 48), 64 KB
    case 6: return launch_gemm<128, 192, 2, 2, 3, 1>(p, st, device);   // 4 wavefronts, 3-deep ring, 120 KB: one workgroup per CU
    case 7: return launch_gemm<256, 192, 4, 2, 2, 1>(p, st, device);   // 8 wavefronts (64 x 96), 112 KB
    case 8: return launch_gemm<128, 96, 2, 2, 4, 1>(p, st, device);    // 4 wavefronts (64 x 48), 4-deep ring, 112 KB
    case 9: return launch_gemm<64, 96, 2, 2, 4, 2>(p, st, device);     // 4 wavefronts (32 x 48), 80 KB: small batches
    case 10: return launch_gemm<128, 96, 2, 2, 3, 1>(p, st, device);   // 4 wavefronts (64 x 48), 3-deep ring, 84 KB
    case 11: return launch_gemm<128, 192, 2, 4, 3, 1>(p, st, device);  // 8 wavefronts (64 x 48), 3-deep ring, 120 KB
    case 12: return launch_gemm<128, 192, 1, 4, 3, 1>(p, st, device);  // 4 wavefronts (128 x 48), 3-deep ring, 120 KB
    case 13: return launch_gemm<128, 192, 2, 2, 4, 1>(p, st, device);  // 4 wavefronts (64 x 96), 4-deep ring, 160 KB
    case 14: return launch_gemm<256, 96, 2, 2, 3, 1>(p, st, device);   // 4 wavefronts (128 x 48), 3-deep ring, 132 KB
    case 15: return launch_gemm<64, 192, 1, 4, 3, 1>(p, st, device);   // 4 wavefronts (64 x 48), 3-deep ring, 96 KB
    case 16: return launch_gemm<64, 192, 2, 4, 3, 1>(p, st, device);   // 8 wavefronts (32 x 48), 3-deep ring, 96 KB
    case 17: return launch_gemm32<256, 192, 2, 4, 4>(p, st, device);   // 8 wavefronts (128 x 48), 32-deep k-tiles, 4-deep ring, 112 KB
    case 18: return launch_gemm32<256, 192, 2, 4, 3>(p, st, device);   // ... 3-deep ring, 84 KB
    case 19: return launch_gemm32<256, 192, 4, 2, 4>(p, st, device);   // 8 wavefronts (64 x 96), 4-deep ring
    case 20: return launch_gemm<128, 192, 2, 4, 2, 2>(p, st, device);  // 8 wavefronts (64 x 48), 2-deep ring, 80 KB: two workgroups per CU
    case 21: return launch_gemm<96, 192, 2, 2, 3, 1>(p, st, device);   // 4 wavefronts (48 x 96), 108 KB: 18 x 14 = 252 workgroups at 1 700 rows
    case 22: return launch_gemm<96, 96, 2, 2, 4, 1>(p, st, device);    // 4 wavefronts (48 x 48), 4-deep ring, 96 KB: 252 workgroups for the F-wide layers
    case 23: return launch_gemm<96, 96, 2, 2, 3, 1>(p, st, device);    // ... 3-deep ring, 72 KB
    case 24: return launch_gemm<96, 192, 2, 2, 4, 1>(p, st, device);   // 4-deep ring, 144 KB
    case 25: return launch_gemm<96, 96, 2, 3, 3, 1>(p, st, device);    // 6 wavefronts (48 x 32), 72 KB
    case 26: return launch_gemm<192, 96, 2, 2, 2, 1>(p, st, device);   // 4 wavefronts (96 x 48), 2-deep ring, 72 KB
    case 27: return launch_gemm<64, 64, 2, 2, 4, 2>(p, st, device);    // 4 wavefronts (32 x 32), 4-deep ring, 64 KB
    case 28: return launch_gemm<96, 64, 2, 2, 3, 1>(p, st, device);    // 4 wavefronts (48 x 32), 60 KB
    case 29: return launch_gemm<192, 192, 2, 4, 3, 1>(p, st, device);  // 8 wavefronts (96 x 48), 3-deep ring, 144 KB: 20 % fewer operand bytes per flop than 128 x 192
    case 30: return launch_gemm<192, 192, 4, 2, 3, 1>(p, st, device);  // 8 wavefronts (48 x 96)
    case 31: return launch_gemm32<256, 192, 4, 2, 5>(p, st, device);   // 8 wavefronts (64 x 96), 32-deep k-tiles, 5-deep ring, 140 KB
    case 32: return launch_gemm32<256, 192, 2, 4, 5>(p, st, device);   // 8 wavefronts (128 x 48), 5-deep ring
    case 33: return launch_gemm<192, 96, 2, 2, 4, 1>(p, st, device);   // 4 wavefronts (96 x 48), 4-deep ring, 144 KB
    case 34: return launch_gemm<192, 192, 2, 2, 3, 1>(p, st, device);  // 4 wavefronts (96 x 96), 3-deep ring, 144 KB
    case 35: return launch_gemm<128, 192, 2, 4, 3, 1, 4>(p, st, device);   // config 11 wave-specialised: 8 computing + 4 loading wavefronts
    case 36: return launch_gemm<128, 192, 2, 4, 4, 1, 4>(p, st, device);   // ... 4-deep ring (160 KB)
    case 37: return launch_gemm<128, 192, 2, 2, 3, 1, 4>(p, st, device);   // config 6 wave-specialised: 4 computing + 4 loading wavefronts
    case 38: return launch_gemm<128, 192, 2, 4, 3, 1, 2>(p, st, device);   // 8 computing + 2 loading wavefronts
    case 39: return launch_gemm<128, 192, 2, 4, 3, 1, 8>(p, st, device);   // 8 computing + 8 loading wavefronts
    case 40: return launch_gemm<96, 96, 2, 2, 3, 1, 2>(p, st, device);     // config 23 wave-specialised: 4 computing + 2 loading wavefronts
    case 41: return launch_gemm<64, 64, 2, 2, 4, 2, 2>(p, st, device);     // config 27 wave-specialised
    case 42: return launch_gemm<64, 96, 2, 2, 4, 2, 2>(p, st, device);     // config 9 wave-specialised
    case 43: return launch_gemm<128, 96, 2, 2, 3, 1, 2>(p, st, device);    // config 10 wave-specialised
    case 44: return launch_gemm<96, 96, 2, 2, 3, 1, 4>(p, st, device);     // config 23 with 4 loading wavefronts
    case 45: return launch_gemm<64, 64, 2, 2, 4, 2, 4>(p, st, device);     // config 27 with 4 loading wavefronts
    case 46: return launch_gemm<128, 192, 2, 4, 3, 1, 0, 1>(p, st, device); // config 11 with wavefronts 4-7 half a k-tile behind
    case 47: return launch_gemm<128, 192, 2, 4, 4, 1, 0, 1>(p, st, device); // ... 4-deep ring
    case 48: return launch_gemm<256, 192, 4, 2, 3, 1, 0, 0, 2>(p, st, device); // 256 x 192 on 8 wavefronts (64 x 96 each), split ring 3 X + 2 W stages = 144 KB: 112 workgroups for the 2F-wide layer at 2 048 rows
    case 49: return launch_gemm<256, 192, 2, 2, 3, 1, 0, 0, 2>(p, st, device); // ... on 4 wavefronts (128 x 96 each)
    // round 5: the 128 x 192 tile in 80 KB or less (32-deep k-tiles), so that TWO workgroups -- one of each session's GEMMs -- share a CU
    case 50: return launch_gemm32<128, 192, 2, 4, 4, 4>(p, st, device);   // 8 wavefronts (64 x 48), 4-deep ring of 32-deep k-tiles, 80 KB, <= 128 registers
    case 51: return launch_gemm32<128, 192, 2, 4, 3, 4>(p, st, device);   // ... 3-deep ring, 60 KB
    case 52: return launch_gemm32<128, 192, 2, 2, 4, 2>(p, st, device);   // 4 wavefronts (64 x 96), 4-deep ring, 80 KB, <= 256 registers
    case 54: return launch_gemm<96, 192, 2, 2, 3, 1, 4>(p, st, device);    // config 21 wave-specialised: 4 computing (48 x 96) + 4 loading wavefronts, 108 KB: 252 workgroups at 1 700 rows, 25 % fewer operand bytes per CU than 96 x 96
    case 55: return launch_gemm<96, 192, 2, 4, 3, 1, 4>(p, st, device);    // ... 8 computing (48 x 48) + 4 loading wavefronts
    case 56: return launch_gemm<192, 96, 2, 2, 3, 1, 4>(p, st, device);    // 192 x 96: 4 computing (96 x 48) + 4 loading wavefronts
    case 57: return launch_gemm<192, 96, 4, 2, 3, 1, 4>(p, st, device);    // ... 8 computing (48 x 48) + 4 loading wavefronts
    case 58: return launch_gemm<192, 96, 2, 2, 4, 1, 4>(p, st, device);    // ... 4 + 4, 4-deep ring (144 KB)
    case 59: return launch_gemm<192, 96, 2, 2, 3, 1, 2>(p, st, device);    // ... 4 computing + 2 loading wavefronts
    case 53: return launch_common<256, 192>(c4_head_gemm4_kernel<256, 192>, p, 256, 2 * (256 + 192) * BK * 2, st, device);   // round 5: 4 wavefronts (128 x 96 each), operands staged through registers, 2-deep ring, 112 KB
    default: return c4host::fail(C4_ERR_BAD_ARG, "c4_linear_bf16: unknown config");
  }
}
