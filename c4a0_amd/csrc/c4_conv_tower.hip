// c4_conv_tower.hip -- the convolutional tower of ConnectFourNet (reference src/c4a0/nn.py:64-70,
// 184-195) as ONE hand-written MFMA kernel for gfx950.
//
//   x0 = Conv3x3(2 -> C)(planes)                       (no BN / ReLU, nn.py:65)
//   x_{i+1} = x_i + ReLU(BN(Conv3x3(Conv3x3(x_i))))    (ResidualBlock, nn.py:184-195; BN folded)
//
// The stock PyTorch path runs this as 9 MIOpen implicit-GEMM launches plus ~25 elementwise
// launches (bias, ReLU, residual add, layout copies), each streaming the [G,C,6,7] activation
// through HBM.  A board is only 42 cells x C channels (2.7 KB at C = 32), so here a workgroup
// keeps NB boards in LDS for the WHOLE tower: HBM sees 168 B in and 42*C*2 B out per board.
//
// Mapping (C = 32 shown; C = 64 doubles the channel groups and halves NB):
//   * LDS activation image: [channel group of 8][board][padded cell] 16-byte slots.  Boards are
//     padded to 8 columns (one shared zero column between rows) plus a zero row above and below
//     (shared between consecutive boards), so a 3x3 tap is a constant slot offset 8*dr + dc and
//     needs no bounds test.
//   * GEMM orientation: D[co, cell] = sum_k W[co, k] * X[k, cell] with v_mfma_f32_16x16x32_bf16:
//     A = weights (held in registers for the whole layer), B = activations, one k-step = one tap
//     x 32 input channels = one ds_read_b128 per lane.  A tile is 16 consecutive padded cells
//     (two board rows): 16 distinct 16-byte slots per channel group -> bank-conflict free.
//   * D leaves the MFMA as 4 consecutive output channels per lane for one cell: bias, ReLU and
//     the residual add happen in registers and go back to LDS as one 8-byte store.
//   * 8 waves per workgroup (two per SIMD).  C = 32: each owns 1/8 of the tiles of every layer and walks
//     them software-pipelined (next tile's fragments and the previous tile's epilogue under the
//     current tile's MFMAs, see tower_layer).  C = 64: the waves work in pairs that share 1/4 of the
//     tiles and split the output channels, so a wave holds half of a layer's weights (144 registers).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <type_traits>

#include "../../include/c4a0_hip.h"
#include "c4_head_out.hpp"
#include "c4_host.hpp"
#include "c4_timeline.hpp"

#ifndef C4_TOWER_PAIR_SAME_SIMD
#define C4_TOWER_PAIR_SAME_SIMD 0
#endif
#ifndef C4_TOWER_ALT_PRIO
#define C4_TOWER_ALT_PRIO 0
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A board's padded image uses slots 0..65 of its channel-group plane region; consecutive boards are
// 56 slots apart, so a board's zero row below row 5 IS the next board's zero row above row 0 (slots
// 57..64 = next board's 1..8).  The two other shared slots are harmless: a board's slot 0 (= the
// previous board's last cell) is read only as the (-1,-1) tap of a halo-column output that is never
// stored, and its slot 65 is the next board's always-zero halo column.
constexpr int kBS = 56;
constexpr int kTilesPerBoard = 3;

// slot of board cell (row r, column c): rows 1..6 of an 8-wide padded image, +1 so that the
// (-1,-1) tap of the first tile lane stays inside the board's own region
__device__ __forceinline__ int cell_slot(int r, int c) { return (r + 1) * 8 + (c + 1) + 1; }

template <int C, int NB>
struct Geo {
  static constexpr int KG = C / 8;                 // channel groups of 8 (16-byte slots)
  static constexpr int MT = C / 16;                // output-channel tiles of 16
  static constexpr int KC = C / 32;                // k-steps per tap
  static constexpr int kPlane = (NB * kBS + 10 + 15) / 16 * 16;   // slots per channel-group plane
  static constexpr int kBufSlots = KG * kPlane;    // slots per activation buffer
  // C = 32: one layer's weight fragments (18 KB) are staged in LDS by DMA a layer ahead; at C = 64
  // a layer's weights are 72 KB and go from global memory straight to registers
  static constexpr bool kStageW = (C == 32);
  static constexpr int kWFrags = 9 * MT * KC;      // 1 KB fragments (64 lanes x 16 bytes) per C -> C layer
  static constexpr int kLdsBytes = 2 * kBufSlots * 16 + (kStageW ? 2 * kWFrags * 1024 : 0);   // two stages: layers alternate
  static constexpr int kTiles = NB * kTilesPerBoard;
  static_assert((kPlane * 16) % 256 == 0, "channel-group planes must keep the bank phase");
  static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
};

#ifdef C4_PHASE_STAMPS
// Diagnostic build only (tools/tower_phases.py): where a tower workgroup's time goes.  Stamps (100 MHz device clock, first
// wavefront of every workgroup): 0 entry, 1 input staged (images zeroed, planes in), 2 conv0 done, 2 + i residual layer i done
// (barrier passed; the last layer: its stores acknowledged); summed over workgroups, plus the earliest entry and latest exit.
// (round 5: room for three stamps per layer of the 16-layer 64-channel tower -- k-loop done, epilogue done, pair hand-over passed)
__device__ unsigned long long c4_tower_clk[80];   // [0] workgroups, [1..60] phase ticks, [78] min entry, [79] max exit
#define C4_TSTAMP(i) do { if (threadIdx.x == 0) tw_ts[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define C4_TSTAMP(i) do { } while (0)
#endif

struct TowerParams {
  const uint16_t* planes;   // [G][2][42] bf16
  const bf16x8* w0;         // conv0 fragments  [3 steps][MT][64 lanes]
  const bf16x8* w;          // block conv fragments [2*n_blocks][9 taps][MT][KC][64 lanes]
  const float* bias;        // [1 + 2*n_blocks][C]
  uint16_t* out;            // [G][42][C] bf16
  uint32_t n_boards;
  uint32_t n_blocks;
};

// One conv layer for this wavefront's G cell tiles.  kConv0: the 2-channel input layer (3 k-steps whose
// k-groups are taps); otherwise a C -> C layer (9 taps x C/32 k-steps).  kSecond: the second conv of a
// residual block (dst += ReLU(.)).
//
// Schedule per wavefront (round 3; the round-2 kernel read 18 fragments, waited for ALL of them and for
// the weight DMA, ran 36 MFMAs back to back and then a serial epilogue of four dependent LDS round
// trips: the matrix pipe was busy 43 % of the time, profiles/r03_evaluator_pmc.json):
//   * one tile (MTW accumulators, started at the bias) per step, software-pipelined over the tiles:
//     while tile g's MFMAs run, tile g + 1's B fragments are already on their way from LDS (kPrefetch:
//     a second fragment set, C = 32) and the epilogue of tile g - 1 -- bias/ReLU/residual/convert and
//     its 8-byte LDS stores -- is issued between them by the scheduler (same basic block, no branch:
//     the residual was read a step earlier, halo lanes are masked at the store only);
//   * counted lgkmcnt waits: an MFMA waits for the fragments it uses, not for everything in flight (the
//     weight staging uses the MUBUF form of LDS-DMA: with the flat-encoded global_load_lds hipcc drains
//     vmcnt AND lgkmcnt to zero in front of the first MFMA of every tile);
//   * weights: C = 32 -- the workgroup fetches each layer's 18 KB ONCE, by global->LDS DMA into one of
//     two stages a whole layer ahead, and every wavefront copies them LDS->registers at the start of
//     the layer.  C = 64 -- requested from global memory after this layer's last epilogue.
//   * kLast (the tower's final layer): the epilogue stores to out[g][cell][C] in global memory instead of
//     the LDS image -- no output pass, no barrier before it (8 bytes per lane, a cell's C channels are
//     one 64- or 128-byte run written by 4 (lg) x MTW stores of neighbouring lanes).
template <int C, int NB, bool kConv0, bool kSecond, bool kLast, int G_TILES, int MTW, bool kPrefetch, typename WF, typename Hook>
__device__ __forceinline__ void tower_layer(const uint4* __restrict__ src, uint4* __restrict__ dst, WF& wf,
                                            const float* __restrict__ bias, int tile_lo, int m0, int lane, Hook&& after_last_tile,
                                            uint16_t* __restrict__ out = nullptr, uint32_t board0 = 0, uint32_t n_boards = 0) {
  using G = Geo<C, NB>;
  constexpr int kSteps = kConv0 ? 3 : 9 * G::KC;        // MFMA k-steps (= B fragments) per tile
  constexpr int NF = kPrefetch ? 2 : 1;                 // fragment sets
  const int li = lane & 15, lg = lane >> 4;

  f32x4 bias4[MTW];   // this wavefront's output-channel tiles are m0 .. m0 + MTW - 1
#pragma unroll
  for (int m = 0; m < MTW; m++) {
    const float* bp = bias + 16 * (m0 + m) + 4 * lg;
    bias4[m] = f32x4{bp[0], bp[1], bp[2], bp[3]};
  }

  uint4 fr[NF][kSteps];
  f32x4 acc[2][MTW];
  uint2 old[2][MTW];
  int e_slot[2], e_cell[2];     // of the tile whose accumulators sit in acc[.]: 16-byte slot index in a plane, padded cell
  int e_board[2];               // ... and its board within the workgroup (kLast)

  auto tile_cell = [&](int g, int& cell) __attribute__((always_inline)) {   // plane-relative slot of this lane's cell of tile g
    const int tl = tile_lo + g;
    const int bidx = tl / kTilesPerBoard;
    const int slot = 9 + 16 * (tl - bidx * kTilesPerBoard) + li;            // cell_slot of the tile's first cell is 9 + 16 j
    cell = slot;
    return bidx * kBS + slot;
  };
  auto load_frags = [&](int g, int f) __attribute__((always_inline)) {
    int cell;
    const int base = tile_cell(g, cell);
#pragma unroll
    for (int k = 0; k < kSteps; k++) {
      int d, plane;
      if (kConv0) {
        // k-step k, k-group lg <-> tap 4 k + lg (taps >= 9 have zero weights); the input image is
        // channel group 0: 8 "channels" per cell, 2 real
        int tap = 4 * k + lg;
        tap = tap < 9 ? tap : 4;
        d = 8 * (tap / 3 - 1) + (tap % 3 - 1);
        plane = 0;
      } else {
        const int t = k / G::KC, kc = k % G::KC;
        d = 8 * (t / 3 - 1) + (t % 3 - 1);
        plane = 4 * kc + lg;
      }
      fr[f][k] = src[plane * G::kPlane + base + d];
    }
  };
  // 8-byte half of the 16-byte slot of channel group 2 (m0 + m) + lg / 2 for this lane's cell
  auto out_ptr = [&](int slot_in_plane, int m) __attribute__((always_inline)) {
    return reinterpret_cast<uint2*>(&dst[(2 * (m0 + m) + (lg >> 1)) * G::kPlane + slot_in_plane]) + (lg & 1);
  };
  auto start_tile = [&](int g, int a) __attribute__((always_inline)) {     // geometry + residual of tile g, accumulators at the bias
    e_slot[a] = tile_cell(g, e_cell[a]);
    e_board[a] = (tile_lo + g) / kTilesPerBoard;
#pragma unroll
    for (int m = 0; m < MTW; m++) {
      acc[a][m] = bias4[m];
      if (kSecond && kPrefetch) old[a][m] = *out_ptr(e_slot[a], m);   // read now, used one step later: only this lane ever rewrites these bytes
    }
  };
  auto mfmas = [&](int f, int a) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < kSteps; k++) {
#pragma unroll
      for (int m = 0; m < MTW; m++)
        acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[k][m], __builtin_bit_cast(bf16x8, fr[f][k]), acc[a][m], 0, 0, 0);
    }
  };
  // lane holds output channels 16 m + 4 lg + {0..3} of its cell
  auto epilogue = [&](int a) __attribute__((always_inline)) {
    const bool valid = ((e_cell[a] - 1) & 7) != 0;       // padded column 0 is halo: computed, never stored
#pragma unroll
    for (int m = 0; m < MTW; m++) {
      f32x4 v = acc[a][m];
      if (kSecond) {
        const uint2 o = kPrefetch ? old[a][m] : *out_ptr(e_slot[a], m);   // bf16 x4: widen by shifting into the f32 exponent/mantissa
        const float o0 = __uint_as_float(o.x << 16), o1 = __uint_as_float(o.x & 0xffff0000u);
        const float o2 = __uint_as_float(o.y << 16), o3 = __uint_as_float(o.y & 0xffff0000u);
        v[0] = o0 + fmaxf(v[0], 0.f); v[1] = o1 + fmaxf(v[1], 0.f);
        v[2] = o2 + fmaxf(v[2], 0.f); v[3] = o3 + fmaxf(v[3], 0.f);
      }
      const bf16x4 o = __builtin_convertvector(v, bf16x4);
      if (kLast) {
        // padded cell s: s - 1 = 8 (row + 1) + (col + 1)  ->  board cell 7 row + col
        const int rc = e_cell[a] - 1, cell = 7 * ((rc >> 3) - 1) + (rc & 7) - 1;
        const uint32_t gb = board0 + (uint32_t)e_board[a];
        if (valid && gb < n_boards)
          *reinterpret_cast<uint2*>(out + ((size_t)gb * 42 + cell) * C + 16 * (m0 + m) + 4 * lg) = __builtin_bit_cast(uint2, o);
      } else if (valid) {
        *out_ptr(e_slot[a], m) = __builtin_bit_cast(uint2, o);
      }
    }
  };

  load_frags(0, 0);
  if (kPrefetch) {
#pragma unroll
    for (int g = 0; g < G_TILES; g++) {
      const int a = g & 1, f = g & 1;
      start_tile(g, a);
      if (g + 1 < G_TILES) load_frags(g + 1, f ^ 1);                  // in flight under this tile's MFMAs
      mfmas(f, a);
      if (g >= 1) epilogue(a ^ 1);                                    // the previous tile's: scheduled among the MFMAs above
    }
    epilogue((G_TILES - 1) & 1);
  } else {
    // C = 64: a wavefront's weights alone are 144 registers -- one fragment set, one accumulator set
#pragma unroll
    for (int g = 0; g < G_TILES; g++) {
      start_tile(g, 0);
      mfmas(0, 0);
      epilogue(0);
      if (g + 1 < G_TILES) load_frags(g + 1, 0);
    }
  }
  after_last_tile();                                                  // after the last epilogue: fewer live registers
}

// The same layer with the loops the other way round, for C = 64 (round 3): k-steps OUTSIDE, the wavefront's cell tiles
// inside.  Every accumulator still sums its k-steps in tap order (same bits as tower_layer), but a k-step's weight
// fragments are needed once per layer, so they are STREAMED from global memory (L1/L2: all wavefront pairs of the
// workgroup ask for the same lines) through a ring of kDepth k-steps instead of being held for the whole layer:
// 48 registers instead of 144, and -- what this is about -- no wavefront ever waits for a layer's 36 KB after the
// previous layer's last tile (counters, profiles/r03_tower64_pmc.json: half of the wave time was waiting, 4.4 us per
// layer with the matrix pipe idle, 123 us per 2 048 boards of which 53 are MFMA time).  The ring runs across layer
// boundaries (the layers' fragments are contiguous in global memory; kSteps % kDepth == 0 keeps the phase), so the
// first k-steps of layer L + 1 are requested under the last MFMAs of layer L.  B fragments of k-step s + 1 (one per
// tile) are read from LDS under the MFMAs of k-step s; the accumulators of all G_TILES tiles live in registers and the
// epilogue (bias already in, ReLU / residual / convert / store) runs once per layer.
struct NoStamp { __device__ __forceinline__ void operator()(int) const {} };
// Where a streamed layer's weight fragments come from.  NoFeed: each wavefront asks global memory (L1 / L2) for its own, kDepth
// k-steps ahead.  RingFeed (round 5, the 8-wavefront 64-channel kernel): the WORKGROUP fetches every fragment once, by global -> LDS
// DMA, into a ring of four stages of one tap each (KC k-steps x MT fragments = 8 KB at C = 64), FOUR taps ahead of its use; the
// wavefronts read their MTW fragments of k-step s + 2 out of the ring under the MFMAs of k-step s.  Why: with the weights
// requested from global memory three k-steps ahead -- all the register ring has room for -- a request has 576 cycles of the
// wavefront's own MFMAs (~1 150 with its SIMD partner's) to come back, less than an L2 hit takes once 256 workgroups stream: the
// k-loop ran at half the matrix pipe's rate (tools/tower_phases.py) and 4.6 MB of fragments per workgroup and launch went
// through the CU's L1.  One workgroup barrier per tap replaces the wavefront pairs' hand-over between layers:
//   top of stage t (tap t of the tower, counted across layers):  my DMA piece of stage t + 1 has landed, my LDS reads and
//   stores so far are done | barrier | stage t's slot (whose fragments everybody read during stage t - 1) takes stage t + 4.
struct NoFeed { static constexpr bool kRing = false; };
template <int C>
struct RingFeed {
  static constexpr bool kRing = true;
  static constexpr int kStageSlots = (C / 16) * (C / 32) * 64;     // 16-byte slots per stage: MT x KC fragments of 64 lanes
  const uint4* ring;                                                // LDS, 4 stages
  __amdgpu_buffer_rsrc_t rsrc;                                      // all residual layers' fragments, contiguous
  int t, n_stages, wave, lane;
  __device__ __forceinline__ void issue(int st) {                   // this wavefront's 1 KB piece of stage st (past the tower's end: zeros, no memory access)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(ring + (st & 3) * kStageSlots + wave * 64), 16,
                                             (int)((uint32_t)(lane * 16) | (st >= n_stages ? 0x80000000u : 0u)), st * (kStageSlots * 16) + wave * 1024, 0, 0);
  }
  // (Wavefronts 0-3 -- one per SIMD -- issuing two pieces each, so that a SIMD's other wavefront multiplies under the issue: 84.7 -> 86.8 us
  // alone, 110-112 -> 114.6 at 2 048 boards.  Not the issue cost, then.)
  __device__ __forceinline__ void stage_top() {
    t += 1;
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");     // in flight behind stage t + 1: my pieces of stages t + 2, t + 3
    __builtin_amdgcn_s_barrier();
    issue(t + 4);
  }
  template <typename WQ, int MTW>
  __device__ __forceinline__ void read(WQ& wq, int s, int kc, int m0) const {   // k-step `kc` of stage t + 1 -> wq[s % 2]
    const uint4* st = ring + ((t + 1) & 3) * kStageSlots + lane;
#pragma unroll
    for (int m = 0; m < MTW; m++) wq[s % 2][m] = __builtin_bit_cast(bf16x8, st[((m0 + m) * (C / 32) + kc) * 64]);
  }
};
// kMode (round 5): -1 = kSecond / kLast are run-time flags and the epilogue walks its items one by one (the 8-wavefront kernel: one
// inlined copy of the layer, 253 registers); 0 / 1 / 2 = first conv of a block / second conv / the tower's last layer as
// COMPILE-TIME variants whose epilogue requests every residual first and has no branch per item (the 4-wavefront kernel, which
// has the registers: with the flags at run time each of a wavefront's 24 items was "branch, ds_read_b64, s_waitcnt lgkmcnt(0),
// 12 vector instructions, branch" -- 1.5 / 2.5 us per layer of serial LDS round trips that a lone wavefront on its SIMD cannot hide).
// max(v, 0) as ONE instruction (see c4_head_gemm.hip relu1: fmaxf costs a canonicalising v_max first; same result for every v)
__device__ __forceinline__ float tower_relu1(float v) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
  return r;
}
template <int C, int NB, int G_TILES, int MTW, int kDepth, typename WQ, typename Stamp = NoStamp, int kMode = -1, typename Feed = NoFeed>
__device__ __forceinline__ void tower_layer_stream(const bool kSecond, const bool kLast, const uint4* __restrict__ src, uint4* __restrict__ dst, WQ& wq,
                                                   const bf16x8* __restrict__ w_layer, bool has_next, const float* __restrict__ bias,
                                                   int tile_lo, int m0, int lane, uint16_t* __restrict__ out = nullptr,
                                                   uint32_t board0 = 0, uint32_t n_boards = 0, Stamp&& stamp = NoStamp{}, Feed&& feed = NoFeed{}) {
  using G = Geo<C, NB>;
  using FeedT = typename std::remove_reference<Feed>::type;
  constexpr int kSteps = 9 * G::KC;
  static_assert(kSteps % kDepth == 0, "the weight ring keeps its phase from layer to layer");
  static_assert(!FeedT::kRing || (kDepth == 2 && G::KC == 2), "the LDS ring is read one stage (= one tap = two k-steps) ahead");
  const int li = lane & 15, lg = lane >> 4;

  // this layer's fragment of k-step s (tap s / KC, channel half s % KC), output-channel tile m0 + m; s >= kSteps
  // continues into the next layer (same formula, kWFrags fragments further)
  auto request = [&](int s) __attribute__((always_inline)) {
    if constexpr (FeedT::kRing) { feed.template read<WQ, MTW>(wq, s, s % G::KC, m0); return; }
    const int sl = s % kSteps;
    const bf16x8* wl = w_layer + (s >= kSteps ? G::kWFrags * 64 : 0);
#pragma unroll
    for (int m = 0; m < MTW; m++) wq[s % kDepth][m] = wl[(((sl / G::KC) * G::MT + m0 + m) * G::KC + (sl % G::KC)) * 64 + lane];
  };

  int base[G_TILES], cell[G_TILES];
  f32x4 acc[G_TILES][MTW];
#pragma unroll
  for (int g = 0; g < G_TILES; g++) {
    const int tl = tile_lo + g;
    const int bidx = tl / kTilesPerBoard;
    cell[g] = 9 + 16 * (tl - bidx * kTilesPerBoard) + li;            // cell_slot of the tile's first cell is 9 + 16 j
    base[g] = bidx * kBS + cell[g];
#pragma unroll
    for (int m = 0; m < MTW; m++) {
      const float* bp = bias + 16 * (m0 + m) + 4 * lg;
      acc[g][m] = f32x4{bp[0], bp[1], bp[2], bp[3]};
    }
  }
  // B fragments.  The three taps of a board row are ONE fragment shifted by a lane: lane (lg, li) holds channels
  // 8 lg .. 8 lg + 7 of padded cell c0 + li, the tap to the right needs cell c0 + li + 1 = lane li + 1's registers, the
  // tap to the left lane li - 1's -- a DPP row shift (rows of 16 lanes = one lg), zero-filled at the row's ends, which
  // is exactly right: cell c0 + 16 is the next tile's halo column (always zero) and lane 0 is itself a halo column whose
  // output is never stored.  So LDS is read once per (tile, board row, channel half) instead of once per tap: a third
  // of the fragment traffic -- the LDS port, loaded as heavily as the matrix pipe by the nine-reads-per-tile loop
  // (8 wavefronts x 6 KB per k-step at 128 B per clock = 2 x 12 MFMAs x 16 cycles per SIMD), stops being a bound.
  uint4 fr[2][G_TILES][G::KC];
  auto read_row = [&](int r, int f) __attribute__((always_inline)) {           // board row r - 1 relative to the output cell, tap column 0
    const int d = 8 * (r - 1);
#pragma unroll
    for (int g = 0; g < G_TILES; g++)
#pragma unroll
      for (int kc = 0; kc < G::KC; kc++) fr[f][g][kc] = src[(4 * kc + lg) * G::kPlane + base[g] + d];
  };
  auto shifted = [&](const uint4& v, int dc) __attribute__((always_inline)) {   // dc = 0: left neighbour's cell, 2: right neighbour's
    if (dc == 1) return v;
    uint4 o;
    if (dc == 2) {          // row_shl:1 -- lane li reads lane li + 1
      o.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x101, 0xF, 0xF, true); o.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x101, 0xF, 0xF, true);
      o.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x101, 0xF, 0xF, true); o.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x101, 0xF, 0xF, true);
    } else {                // row_shr:1 -- lane li reads lane li - 1
      o.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x111, 0xF, 0xF, true); o.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x111, 0xF, 0xF, true);
      o.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x111, 0xF, 0xF, true); o.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x111, 0xF, 0xF, true);
    }
    return o;
  };
  if constexpr (FeedT::kRing) feed.stage_top();                       // (also the hand-over from the previous layer: everybody's epilogue stores are done)
  read_row(0, 0);
#pragma unroll
  for (int r = 0; r < 3; r++) {
    // The fences pin the order the source states (hipcc's scheduler otherwise sinks every LDS read and every weight
    // request to just in front of its first use and waits for it there: one round trip per pair of MFMAs).
    if (r + 1 < 3) read_row(r + 1, (r + 1) & 1);                    // under this row's 6 KC k-steps
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dc = 0; dc < 3; dc++) {
      if constexpr (FeedT::kRing) {
        if (3 * r + dc > 0) { feed.stage_top(); __builtin_amdgcn_sched_barrier(0); }
      }
#pragma unroll
      for (int kc = 0; kc < G::KC; kc++) {
        const int s = (3 * r + dc) * G::KC + kc;                    // k-steps in tap order, as tower_layer sums them
#pragma unroll
        for (int g = 0; g < G_TILES; g++) {
          const uint4 bfrag = shifted(fr[r & 1][g][kc], dc);
#pragma unroll
          for (int m = 0; m < MTW; m++)
            acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[s % kDepth][m], __builtin_bit_cast(bf16x8, bfrag), acc[g][m], 0, 0, 0);
        }
        if (FeedT::kRing || s + kDepth < kSteps || has_next) request(s + kDepth);    // this ring slot is free again (the LDS ring: past the tower's end it holds zeros)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  stamp(1);                                                         // (diagnostic build) the k-loop is done
  if constexpr (kMode >= 0) {
    // ---- epilogue, compile-time variants: every residual requested up front, no branch per item
    constexpr bool kSec = kMode >= 1, kLst = kMode == 2;
    auto optr = [&](int g, int m) __attribute__((always_inline)) {
      return reinterpret_cast<uint2*>(&dst[(2 * (m0 + m) + (lg >> 1)) * G::kPlane + base[g]]) + (lg & 1);
    };
    uint2 res[G_TILES][MTW];
    if (kSec) {
#pragma unroll
      for (int g = 0; g < G_TILES; g++)
#pragma unroll
        for (int m = 0; m < MTW; m++) res[g][m] = *optr(g, m);
    }
    // padded column 0 is halo: (cell - 1) & 7 == li & 7 in every tile; the LDS image takes the halo lanes' result as zeros over zeros
    const bool valid = (li & 7) != 0;
    const uint32_t vmask = valid ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int g = 0; g < G_TILES; g++) {
#pragma unroll
      for (int m = 0; m < MTW; m++) {
        f32x4 v = acc[g][m];
        if (kSec) {
          const uint2 o = res[g][m];                                 // the residual stream: bf16 x4, widened by a shift
          v[0] = __uint_as_float(o.x << 16) + tower_relu1(v[0]); v[1] = __uint_as_float(o.x & 0xffff0000u) + tower_relu1(v[1]);
          v[2] = __uint_as_float(o.y << 16) + tower_relu1(v[2]); v[3] = __uint_as_float(o.y & 0xffff0000u) + tower_relu1(v[3]);
        }
        uint2 ob = __builtin_bit_cast(uint2, __builtin_convertvector(v, bf16x4));
        if (kLst) {
          const int rc = cell[g] - 1, bcell = 7 * ((rc >> 3) - 1) + (rc & 7) - 1;
          const uint32_t gb = board0 + (uint32_t)((tile_lo + g) / kTilesPerBoard);
          if (valid && gb < n_boards) *reinterpret_cast<uint2*>(out + ((size_t)gb * 42 + bcell) * C + 16 * (m0 + m) + 4 * lg) = ob;
        } else {
          ob.x &= vmask; ob.y &= vmask;
          *optr(g, m) = ob;
        }
      }
    }
    stamp(2);
    return;
  }
  // ---- epilogue: lane holds output channels 16 (m0 + m) + 4 lg + {0..3} of its cell of every tile
#pragma unroll
  for (int g = 0; g < G_TILES; g++) {
    const bool valid = ((cell[g] - 1) & 7) != 0;                     // padded column 0 is halo: computed, never stored
#pragma unroll
    for (int m = 0; m < MTW; m++) {
      uint2* op = reinterpret_cast<uint2*>(&dst[(2 * (m0 + m) + (lg >> 1)) * G::kPlane + base[g]]) + (lg & 1);
      f32x4 v = acc[g][m];
      if (kSecond) {
        const uint2 o = *op;                                         // the residual stream: bf16 x4, widened by a shift
        v[0] = __uint_as_float(o.x << 16) + fmaxf(v[0], 0.f); v[1] = __uint_as_float(o.x & 0xffff0000u) + fmaxf(v[1], 0.f);
        v[2] = __uint_as_float(o.y << 16) + fmaxf(v[2], 0.f); v[3] = __uint_as_float(o.y & 0xffff0000u) + fmaxf(v[3], 0.f);
      }
      const bf16x4 o = __builtin_convertvector(v, bf16x4);
      if (kLast) {
        const int rc = cell[g] - 1, bcell = 7 * ((rc >> 3) - 1) + (rc & 7) - 1;
        const uint32_t gb = board0 + (uint32_t)((tile_lo + g) / kTilesPerBoard);
        if (valid && gb < n_boards)
          *reinterpret_cast<uint2*>(out + ((size_t)gb * 42 + bcell) * C + 16 * (m0 + m) + 4 * lg) = __builtin_bit_cast(uint2, o);
      } else if (valid) {
        *op = __builtin_bit_cast(uint2, o);
      }
    }
  }
  stamp(2);                                                         // ... the epilogue
}

// MS = 1: every wavefront computes all C/16 output-channel tiles of its cell tiles.  MS = 2 (C = 64):
// the output-channel tiles are split over two wavefronts that share the cell tiles, which halves the
// weights a wavefront holds (144 instead of 288 registers) so that two wavefronts fit on a SIMD.
// One layer's weight fragments, global -> LDS by DMA in the MUBUF form (buffer_load ... lds): KW pieces of
// 1 KB shared out over the workgroup's wavefronts.  (A function of its own, not a lambda in the kernel:
// with the buffer-resource builtins inside a kernel-body lambda hipcc 7.2 silently drops the kernel's host stub.)
template <int KW, int NWAVES>
__device__ __forceinline__ void dma_weights(const void* w, uint32_t bytes, uint4* lds_dst, int first_frag, int wave, int lane) {
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, (int)bytes, 0x00020000);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  for (int c = wave_u; c < KW; c += NWAVES)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds_dst + c * 64), 16, lane * 16, (first_frag + c) * 1024, 0, 0);
}

// ST (C = 64): the residual layers run as tower_layer_stream (weights streamed through a ring, k-steps outside).
// KD (ST only): k-steps of weights in flight in the register ring.
// RG (ST, MS = 2, 8 wavefronts): the weights through the workgroup's LDS ring (RingFeed) instead of every wavefront's own global loads.
template <int C, int NB, int NT, int MS, bool ST = false, int KD = 3, bool RG = false>
__global__ __launch_bounds__(NT) void c4_conv_tower_kernel(const uint16_t* __restrict__ a_planes, const bf16x8* __restrict__ a_w0, const bf16x8* __restrict__ a_w,
                                                             const float* __restrict__ a_bias, uint16_t* __restrict__ a_out, uint32_t a_n_boards, uint32_t a_n_blocks) {
  // flat scalar arguments (12 dwords): preloaded into SGPRs at wavefront launch (build.py: -amdgpu-kernarg-preload-count)
  const TowerParams p{a_planes, a_w0, a_w, a_bias, a_out, a_n_boards, a_n_blocks};
  C4_TL_BEGIN();
#ifdef C4_PHASE_STAMPS
  unsigned long long tw_ts[64];
  const unsigned long long wave_c0 = __builtin_amdgcn_s_memtime();   // shader cycles against 100 MHz ticks: the clock this workgroup ran at
  const unsigned long long wave_r0 = __builtin_amdgcn_s_memrealtime();
  auto tw_flush = [&](int last) {
    if (threadIdx.x == 0) {
      atomicAdd(&c4_tower_clk[70], __builtin_amdgcn_s_memtime() - wave_c0);
      atomicAdd(&c4_tower_clk[71], __builtin_amdgcn_s_memrealtime() - wave_r0);
      atomicAdd(&c4_tower_clk[0], 1ull);
      for (int i = 0; i < last; i++) atomicAdd(&c4_tower_clk[1 + i], tw_ts[i + 1] - tw_ts[i]);
      atomicMin(&c4_tower_clk[78], tw_ts[0]);
      atomicMax(&c4_tower_clk[79], tw_ts[last]);
    }
  };
#endif
  C4_TSTAMP(0);
#ifdef C4_PHASE_STAMPS
  const unsigned long long wave_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  using G = Geo<C, NB>;
  constexpr int MTW = G::MT / MS;
  static_assert(G::MT % MS == 0 && (!G::kStageW || MS == 1), "co-tile split");

  extern __shared__ __attribute__((aligned(256))) uint8_t lds_raw[];
  uint4* X = reinterpret_cast<uint4*>(lds_raw);    // block input / residual stream
  uint4* T = X + G::kBufSlots;                     // intermediate (and the conv0 input image)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // Which wavefronts form an MS-group (they share cell tiles and split the output channels).  C4_TOWER_PAIR_SAME_SIMD (round 5
  // experiment): group = wavefronts w and w + NT/64/MS, i.e. (with 8 wavefronts dispatched to SIMDs 0, 2, 1, 3, 0, 2, 1, 3) the
  // two wavefronts of ONE SIMD, instead of the neighbours w, w + 1 that sit on different SIMDs.
#if C4_TOWER_PAIR_SAME_SIMD
  constexpr int kGroups = NT / 64 / MS;
  const int grp_index = wave % kGroups, grp_member = wave / kGroups;
#else
  const int grp_index = wave / MS, grp_member = wave % MS;
#endif
  const int m0 = grp_member * MTW;                 // first output-channel tile of this wavefront
  const uint32_t board0 = blockIdx.x * NB;

  // A fragments (weights) of the current layer, in registers; conv0 uses the first 3 k-steps
  constexpr int kWSteps = 9 * G::KC;
  bf16x8 wf[kWSteps][MTW];
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int m = 0; m < MTW; m++) wf[k][m] = p.w0[(k * G::MT + m0 + m) * 64 + lane];
  const int n_layers = 2 * (int)p.n_blocks;
  if constexpr (ST && RG) {   // the ring's first four stages: requested first, they land under the input staging and conv0
    RingFeed<C> f0{reinterpret_cast<const uint4*>(lds_raw) + 2 * G::kBufSlots,
                   __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(p.w), 0, (int)(2u * p.n_blocks * G::kWFrags * 1024u), 0x00020000), -1, 9 * n_layers,
                   __builtin_amdgcn_readfirstlane(wave), lane};
#pragma unroll
    for (int st = 0; st < 4; st++) f0.issue(st);
  }
  auto load_layer_weights = [&](int layer) __attribute__((always_inline)) {   // layer >= 1: [t][m][kc][lane]
    const bf16x8* wl = p.w + (size_t)(layer - 1) * G::kWFrags * 64;
#pragma unroll
    for (int k = 0; k < kWSteps; k++)
#pragma unroll
      for (int m = 0; m < MTW; m++) wf[k][m] = wl[(((k / G::KC) * G::MT + m0 + m) * G::KC + (k % G::KC)) * 64 + lane];
  };
  // C = 32: LDS stage for one layer's fragments, same [t][m][kc][lane] order as global memory
  uint4* Wst = T + G::kBufSlots;
  auto stage_layer_weights = [&](int layer) __attribute__((always_inline)) {  // global -> LDS DMA, 1 KB per instruction
    if (layer <= n_layers)
      dma_weights<G::kWFrags, NT / 64>(p.w, 2u * p.n_blocks * G::kWFrags * 1024u, Wst + (layer & 1) * G::kWFrags * 64, (layer - 1) * G::kWFrags, wave, lane);
  };
  auto fetch_staged_weights = [&](int layer) __attribute__((always_inline)) {
    const uint4* ws = Wst + (layer & 1) * G::kWFrags * 64;
#pragma unroll
    for (int k = 0; k < kWSteps; k++)
#pragma unroll
      for (int m = 0; m < MTW; m++)
        wf[k][m] = __builtin_bit_cast(bf16x8, ws[(((k / G::KC) * G::MT + m0 + m) * G::KC + (k % G::KC)) * 64 + lane]);
  };
  if (G::kStageW && !ST) stage_layer_weights(1);

  // ---- the input planes are requested first (a global round trip), then both images are zeroed under
  // that latency (halo cells stay zero for the whole kernel), then the planes go in: channel group 0 of T
  // holds {plane0, plane1, 0 x6} per cell ----
  constexpr int kIn = (NB * 42 + NT - 1) / NT;
  uint32_t in_v[kIn];
#pragma unroll
  for (int j = 0; j < kIn; j++) {
    const int i = tid + j * NT;
    const int b = i / 42, cell = i - b * 42;
    const uint32_t g = board0 + b;
    in_v[j] = 0;
    if (i < NB * 42 && g < p.n_boards) in_v[j] = (uint32_t)p.planes[(size_t)g * 84 + cell] | ((uint32_t)p.planes[(size_t)g * 84 + 42 + cell] << 16);
  }
  for (int i = tid; i < 2 * G::kBufSlots; i += NT) X[i] = make_uint4(0, 0, 0, 0);
  int* pair_ctr = reinterpret_cast<int*>(lds_raw + G::kLdsBytes);   // ST only: one counter per MS-group of wavefronts
  if (ST && !RG && tid < 16) pair_ctr[tid] = 0;   // (RG: no hand-over counters -- and the ring's first stage, already on its way, lives there)
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kIn; j++) {
    const int i = tid + j * NT;
    const int b = i / 42, cell = i - b * 42;
    if (i < NB * 42) T[b * kBS + cell_slot(cell / 7, cell % 7)] = make_uint4(in_v[j], 0, 0, 0);
  }
  __syncthreads();
  C4_TSTAMP(1);

  // tiles of this wave: a contiguous range
  constexpr int kWaves = NT / 64 / MS;             // wavefronts (or MS-groups of them) that share out the cell tiles
  static_assert(G::kTiles % kWaves == 0, "every wavefront owns the same number of cell tiles");
  constexpr int kTilesPerWave = G::kTiles / kWaves;
  constexpr bool kPrefetch = (C == 32);            // a second fragment set: 36 more registers at C = 32, 72 at C = 64 (too many)
  constexpr bool kFuseOut = (C == 32);             // C = 64 has no registers left for the global address arithmetic (it would spill)
  const int tile_lo = grp_index * kTilesPerWave;

  if constexpr (ST && RG) {
    static_assert(NT == 512 && MS == 2 && C == 64, "one 1 KB DMA piece per wavefront and stage");
    // LDS behind the two images: the ring (4 stages x 8 KB), then the residual layers' biases (the layers read them at their top;
    // out of global memory that load would sit behind the DMA pieces in flight and wait for all of them)
    const uint4* ring = T + G::kBufSlots;
    float* bias_lds = reinterpret_cast<float*>(lds_raw + 2 * G::kBufSlots * 16 + 4 * RingFeed<C>::kStageSlots * 16);
    RingFeed<C> feed{ring, __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(p.w), 0, (int)(2u * p.n_blocks * G::kWFrags * 1024u), 0x00020000),
                     -1, 9 * n_layers, __builtin_amdgcn_readfirstlane(wave), lane};
    // (the first four stages were requested at the kernel's top -- see below -- and have had conv0's time to land)
    for (int i = tid; i < n_layers * C; i += NT) bias_lds[i] = p.bias[C + i];
    bf16x8 wq[2][MTW];
    tower_layer<C, NB, true, false, false, kTilesPerWave, MTW, false>(T, X, wf, p.bias, tile_lo, m0, lane, [&]() __attribute__((always_inline)) {});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // stage 0 is in the ring (everybody's piece), the biases are in LDS
    feed.template read<decltype(wq), MTW>(wq, 0, 0, m0);   // t = -1: stage 0's two k-steps
    feed.template read<decltype(wq), MTW>(wq, 1, 1, m0);
    for (int layer = 1; layer <= n_layers; layer++) {
      const bool second = (layer & 1) == 0;
      const float* bl = bias_lds + (size_t)(layer - 1) * C;
      // the three compile-time copies of the layer (kMode: every residual requested up front, no branch per item): with the register ring at two
      // k-steps this kernel has the registers for them (252, no spill; the 8-wavefront kernel that streams from global memory spills: 201 us)
      if (layer == n_layers)
        tower_layer_stream<C, NB, kTilesPerWave, MTW, 2, decltype(wq), NoStamp, 2, RingFeed<C>&>(true, true, T, X, wq, nullptr, false, bl, tile_lo, m0, lane, p.out, board0, p.n_boards, NoStamp{}, feed);
      else if (second)
        tower_layer_stream<C, NB, kTilesPerWave, MTW, 2, decltype(wq), NoStamp, 1, RingFeed<C>&>(true, false, T, X, wq, nullptr, true, bl, tile_lo, m0, lane, p.out, board0, p.n_boards, NoStamp{}, feed);
      else
        tower_layer_stream<C, NB, kTilesPerWave, MTW, 2, decltype(wq), NoStamp, 0, RingFeed<C>&>(false, false, X, T, wq, nullptr, true, bl, tile_lo, m0, lane, p.out, board0, p.n_boards, NoStamp{}, feed);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may land in LDS that was given away (the pieces past the tower's end are zero fills)
    if (n_layers > 0) return;                          // (the last layer stored the tower's output; without residual blocks conv0's image is copied out below)
  } else if constexpr (ST) {
    constexpr int kDepth = KD;                       // k-steps of weights in flight: KD x (G_TILES x MTW) MFMAs x 16 cycles ahead of their use
    bf16x8 wq[kDepth][MTW];
    if (n_layers >= 1) {                             // layer 1's first k-steps travel under conv0
#pragma unroll
      for (int sdx = 0; sdx < kDepth; sdx++)
#pragma unroll
        for (int m = 0; m < MTW; m++) wq[sdx][m] = p.w[(((sdx / G::KC) * G::MT + m0 + m) * G::KC + (sdx % G::KC)) * 64 + lane];
    }
    tower_layer<C, NB, true, false, false, kTilesPerWave, MTW, false>(T, X, wf, p.bias, tile_lo, m0, lane, [&]() __attribute__((always_inline)) {});
    // MS == 1 with whole boards per wavefront: a wavefront's layer L + 1 reads only what its own layer L wrote (and
    // zero rows), LDS operations of one wavefront execute in order -- no barrier between layers, the two wavefronts of
    // a SIMD drift apart and one's epilogue runs under the other's MFMAs
    constexpr bool kBarrier = !(MS == 1 && kTilesPerWave % kTilesPerBoard == 0);
    if (kBarrier) __syncthreads();
    C4_TSTAMP(2);
    for (int layer = 1; layer <= n_layers; layer++) {
      const bf16x8* wl = p.w + (size_t)(layer - 1) * G::kWFrags * 64;
      const bool has_next = layer < n_layers;
      const float* bl = p.bias + (size_t)layer * C;
      const bool second = (layer & 1) == 0;          // second conv of a block: T -> X, += residual; the last one stores the tower's output
#if C4_TOWER_ALT_PRIO
      // experiment: the older wavefront of a SIMD wins every arbitration and runs ahead (it finishes the tower at 74 of the
      // launch's 108 us, tools/tower_phases.py); alternate who has priority, layer by layer
      if (((layer ^ (__builtin_amdgcn_readfirstlane(wave) >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
#ifdef C4_PHASE_STAMPS
      auto st_stamp = [&](int q) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (layer <= 19) C4_TSTAMP(2 + 3 * (layer - 1) + q);
      };
#elif defined(C4_TOWER64_FENCE)
      // experiment (round 5): the diagnostic build's stamps made the 64-channel tower FASTER; which part of a stamp does it?
      //   1 = compiler fence only, 2 = + s_waitcnt lgkmcnt(0), 3 = + vmcnt(0) (what a stamp executes, minus the clock read)
      auto st_stamp = [&](int q) __attribute__((always_inline)) {
        if (C4_TOWER64_FENCE == 1) asm volatile("" ::: "memory");
        if (C4_TOWER64_FENCE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (C4_TOWER64_FENCE == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      };
#else
      NoStamp st_stamp;
#endif
      if constexpr (MS == 1) {   // one wavefront per SIMD, registers to spare: three compile-time copies of the layer (see kMode)
        if (layer == n_layers)
          tower_layer_stream<C, NB, kTilesPerWave, MTW, kDepth, decltype(wq), decltype(st_stamp)&, 2>(true, true, T, X, wq, wl, false, bl, tile_lo, m0, lane, p.out, board0, p.n_boards, st_stamp);
        else if (second)
          tower_layer_stream<C, NB, kTilesPerWave, MTW, kDepth, decltype(wq), decltype(st_stamp)&, 1>(true, false, T, X, wq, wl, true, bl, tile_lo, m0, lane, p.out, board0, p.n_boards, st_stamp);
        else
          tower_layer_stream<C, NB, kTilesPerWave, MTW, kDepth, decltype(wq), decltype(st_stamp)&, 0>(false, false, X, T, wq, wl, true, bl, tile_lo, m0, lane, p.out, board0, p.n_boards, st_stamp);
      } else
      tower_layer_stream<C, NB, kTilesPerWave, MTW, kDepth>(second, layer == n_layers, second ? T : X, second ? X : T, wq, wl, has_next, bl, tile_lo, m0, lane,
                                                            p.out, board0, p.n_boards, st_stamp);
#ifdef C4_PHASE_STAMPS
      if (layer == n_layers) {
        st_stamp(3);
        if (layer <= 19) tw_flush(2 + 3 * layer);
        // lifetime of EVERY wavefront of the workgroup, by wavefront number (entry of the workgroup's first wavefront -> this one's exit)
        if (lane == 0) atomicAdd(&c4_tower_clk[62 + wave], __builtin_amdgcn_s_memrealtime() - wave_t0);
      }
#endif
      if (layer == n_layers) return;
      if (kBarrier) {
        // Only the MS wavefronts that share cell tiles must meet between layers (each wrote its share of the output
        // channels of the group's boards; a group's taps reach nothing else that is ever stored).  A workgroup barrier
        // would also line up the two wavefronts of every SIMD -- they belong to different groups -- so that both run
        // their once-per-layer epilogue at the same time with the matrix pipe idle.  Instead each group counts arrivals
        // in LDS: LDS operations of a wavefront execute in order, so whoever sees the count sees the stores before it.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        int* ctr = pair_ctr + grp_index;
        if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < MS * layer) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
#if defined(C4_PHASE_STAMPS) || defined(C4_TOWER64_FENCE)
      st_stamp(3);
#endif
    }
  } else {
  // conv0: input image (T) -> X
  tower_layer<C, NB, true, false, false, kTilesPerWave, MTW, kPrefetch>(T, X, wf, p.bias, tile_lo, m0, lane, [&]() __attribute__((always_inline)) {
    if (!G::kStageW && n_layers >= 1) load_layer_weights(1);
  });
  if (G::kStageW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's share of the staged layer has landed
  __syncthreads();
  C4_TSTAMP(2);
  for (int layer = 1; layer <= n_layers; layer++) {
    const bool is_second = (layer & 1) == 0;         // second conv of a block: T -> X, += residual
    if (G::kStageW) {
      // the other stage was last read at the start of the previous layer, a barrier ago: refill it
      // now for the NEXT layer (lands under this layer's MFMAs), then copy this layer's fragments out
      stage_layer_weights(layer + 1);
      fetch_staged_weights(layer);
    }
    auto next_weights = [&]() __attribute__((always_inline)) {
      if (!G::kStageW && layer < n_layers) load_layer_weights(layer + 1);
    };
    if (kFuseOut && layer == n_layers) {   // the final layer (always the second conv of a block) stores the tower's output itself
      tower_layer<C, NB, false, true, true, kTilesPerWave, MTW, kPrefetch>(T, X, wf, p.bias + (size_t)layer * C, tile_lo, m0, lane, next_weights,
                                                                          p.out, board0, p.n_boards);
#ifdef C4_PHASE_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (layer < 20) { C4_TSTAMP(2 + layer); tw_flush(2 + layer); }
#endif
      C4_TL_END(1, p.out);
      return;
    }
    if (is_second) tower_layer<C, NB, false, true, false, kTilesPerWave, MTW, kPrefetch>(T, X, wf, p.bias + (size_t)layer * C, tile_lo, m0, lane, next_weights);
    else tower_layer<C, NB, false, false, false, kTilesPerWave, MTW, kPrefetch>(X, T, wf, p.bias + (size_t)layer * C, tile_lo, m0, lane, next_weights);
    if (G::kStageW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef C4_PHASE_STAMPS
    if (layer < 20) C4_TSTAMP(2 + layer);
#endif
  }
  }

  if (ST) __syncthreads();   // (streamed path without residual blocks: every thread copies conv0's image out)
  // ---- X -> out[g][cell][C] (C = 64 with held weights, or no residual blocks: conv0's image is the output) ----
  for (int i = tid; i < NB * 42 * G::KG; i += NT) {
    const int kg = i % G::KG;
    const int bc = i / G::KG;
    const int b = bc / 42, cell = bc - b * 42;
    const uint32_t g = board0 + b;
    if (g < p.n_boards) {
      const uint4 v = X[kg * G::kPlane + b * kBS + cell_slot(cell / 7, cell % 7)];
      reinterpret_cast<uint4*>(p.out)[((size_t)g * 42 + cell) * G::KG + kg] = v;
    }
  }
  C4_TL_END(1, p.out);
}

template <int C, int NB, int NT, int MS, bool ST = false, int KD = 3, bool RG = false>
int launch_tower(const TowerParams& p, uint32_t n_boards, hipStream_t stream, int device) {
  // + the wavefront pairs' hand-over counters; RG: + the weight ring (4 stages) and the residual layers' biases
  const int kLds = Geo<C, NB>::kLdsBytes + (ST ? 64 : 0) + (RG ? 4 * RingFeed<C>::kStageSlots * 16 + 2 * (int)p.n_blocks * C * 4 : 0);
  auto k = c4_conv_tower_kernel<C, NB, NT, MS, ST, KD, RG>;
  hipError_t e = c4host::opt_in_lds((const void*)k, RG ? 160 * 1024 : kLds, device);   // (RG: the size depends on the number of layers; the opt-in is made once per kernel)
  if (e != hipSuccess) return c4host::fail(C4_ERR_HIP, std::string("c4_conv_tower_bf16: LDS opt-in (") + std::to_string(kLds) + " bytes) on device " + std::to_string(device) + ": " + hipGetErrorString(e));
  k<<<dim3((n_boards + NB - 1) / NB), dim3(NT), kLds, stream>>>(p.planes, p.w0, p.w, p.bias, p.out, p.n_boards, p.n_blocks);
  e = hipGetLastError();
  if (e != hipSuccess) return c4host::fail(C4_ERR_HIP, std::string("c4_conv_tower_bf16 launch: ") + hipGetErrorString(e));
  return C4_OK;
}

}  // namespace

#ifdef C4_PHASE_STAMPS
// diagnostic build only: mean time per phase (us) of the 32-channel tower's workgroups since the last reset and the span from the
// earliest entry to the latest exit (one launch)
extern "C" int c4_debug_tower_phases(double* phase_us, int n, double* span_us, unsigned long long* n_workgroups, int reset) {
  unsigned long long h[80];
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(c4_tower_clk), sizeof h) != hipSuccess) return C4_ERR_HIP;
  for (int i = 0; i < n && i < 76; i++) phase_us[i] = h[0] ? (double)h[1 + i] / (double)h[0] * 0.01 : 0.0;   // [61..68]: lifetime by wavefront number (64 channels)
  if (span_us) *span_us = h[79] > h[78] ? (double)(h[79] - h[78]) * 0.01 : 0.0;
  if (n_workgroups) *n_workgroups = h[0];
  if (reset) { unsigned long long z[80] = {0}; z[78] = ~0ull; if (hipMemcpyToSymbol(HIP_SYMBOL(c4_tower_clk), z, sizeof z) != hipSuccess) return C4_ERR_HIP; }
  return C4_OK;
}
#endif

C4_TL_SETTER(c4_debug_timeline_tower)

extern "C" {

// Layouts (prepared by c4a0_amd/nn.py::pack_tower_weights):
//   w0_dev   bf16 [3][C/16][64][8] : conv0, k-step s / lane k-group g <-> tap 4 s + g, element j <-> input channel j (2 real)
//   w_dev    bf16 [2*n_blocks][9][C/16][C/32][64][8] : lane l element j = W[co = 16 m + (l & 15)][ci = 32 kc + 8 (l >> 4) + j][tap]
//   bias_dev f32  [1 + 2*n_blocks][C]
//   out_dev  bf16 [n_boards][42][C]   (cell-major, channels last)
// Runs on the device `stream` belongs to (the current device for the null stream).
int c4_conv_tower_bf16(const void* planes_dev, const void* w0_dev, const void* w_dev, const float* bias_dev,
                       uint32_t n_boards, uint32_t channels, uint32_t n_blocks, void* out_dev, uint32_t config, void* stream) {
  if (!planes_dev || !w0_dev || !bias_dev || !out_dev || (n_blocks && !w_dev)) return c4host::fail(C4_ERR_BAD_ARG, "c4_conv_tower_bf16: null argument");
  if (channels != 32 && channels != 64) return c4host::fail(C4_ERR_BAD_ARG, "c4_conv_tower_bf16: channels must be 32 or 64");
  if (n_boards == 0) return C4_OK;
  const int device = c4host::stream_device((hipStream_t)stream);
  c4host::DeviceGuard guard(device);
  if (guard.error() != hipSuccess) return c4host::fail(C4_ERR_HIP, std::string("c4_conv_tower_bf16: hipSetDevice: ") + hipGetErrorString(guard.error()));
  TowerParams p{(const uint16_t*)planes_dev, (const bf16x8*)w0_dev, (const bf16x8*)w_dev, bias_dev, (uint16_t*)out_dev, n_boards, n_blocks};
  // config (what a workgroup owns, never a board's arithmetic): 0 = by size (below), 1 = 16 boards / 8 wavefronts,
  // 2 = 8 boards / 8 wavefronts, 3 = 16 boards / 12 wavefronts (27.3 vs 28.0 us alone at 2 048 boards, no difference
  // in the bench).  64 channels: 2 .. 6 below (1 = the default shape at any size).
  if (config > 6 || (channels == 32 && config > 5)) return c4host::fail(C4_ERR_BAD_ARG, "c4_conv_tower_bf16: config must be 0 (automatic) .. 5 (32 channels) / .. 6 (64 channels)");
  if (channels == 64 && config == 4 && Geo<64, 8>::kLdsBytes + 64 + 4 * RingFeed<64>::kStageSlots * 16 + 2 * (int)n_blocks * 64 * 4 > 160 * 1024)
    return c4host::fail(C4_ERR_BAD_ARG, "c4_conv_tower_bf16: config 4 (weights through an LDS ring) keeps the layers' biases in LDS behind the ring: at most 23 residual blocks");   // an explicitly asked-for shape is run or refused, never swapped for another
  if (channels == 32 && config == 1) return launch_tower<32, 16, 512, 1>(p, n_boards, (hipStream_t)stream, device);
  if (channels == 32 && config == 2) return launch_tower<32, 8, 512, 1>(p, n_boards, (hipStream_t)stream, device);
  if (channels == 32 && config == 3) return launch_tower<32, 16, 768, 1>(p, n_boards, (hipStream_t)stream, device);
  // round 6, for the narrow launches of a job's tail (a round there is latency, and the tower's nine dependent layers were a third of it):
  // fewer cell tiles per wavefront and layer -- 4: 4 boards on 12 wavefronts (one tile each), 5: 2 boards on 6 wavefronts (one tile each).
  // Alone under rocprofv3, 4-block tower (profiles/r06_tower_small.txt): 256 boards 16.8 us (8 boards per workgroup) -> 11.8 -> 9.8;
  // 512: 16.4 -> 12.4 -> 11.2; 1 024: 17.3 -> 13.2 (two boards per workgroup: 15.3); 2 048: 19.2 against 25.2 (four): the cuts below
  if (channels == 32 && config == 4) return launch_tower<32, 4, 768, 1>(p, n_boards, (hipStream_t)stream, device);
  if (channels == 32 && config == 5) return launch_tower<32, 2, 384, 1>(p, n_boards, (hipStream_t)stream, device);
  if (channels == 32 && n_boards <= 512) return launch_tower<32, 2, 384, 1>(p, n_boards, (hipStream_t)stream, device);
  if (channels == 32 && n_boards <= 1024) return launch_tower<32, 4, 768, 1>(p, n_boards, (hipStream_t)stream, device);
  if (channels == 32 && n_boards <= 8 * 160)
    // small launches (1 025 .. 1 280 boards; round 6: the still smaller ones above): 8 boards per workgroup, three tiles in flight per wave, so
    // that the launch spreads over twice as many CUs (2 048 boards alone: 31.6 -> 20.5 us).  NOT used for
    // the 2 048-board launches of two concurrent sessions: there the other session fills the rest of
    // the chip and what counts is CU-time per board, which is 30 % higher this way (measured: -1.7 %
    // games/s at BASELINE config 2); a caller whose launch has the chip to itself asks for config 2 up to
    // 2 048 boards (c4a0_amd/nn.py latency_mode: 27.3 -> 19.1 us).
    return launch_tower<32, 8, 512, 1>(p, n_boards, (hipStream_t)stream, device);
#ifdef C4_DIAG_VARIANTS   // diagnostic build only (build.py --diag): measured alternatives, not part of libc4a0_hip.so
  static const int st32 = [] { const char* e = getenv("C4_TOWER32_STREAM"); return e ? atoi(e) : 0; }();   // the streamed layer at 32 channels (slower there)
  if (channels == 32 && st32) return launch_tower<32, 16, 512, 1, true>(p, n_boards, (hipStream_t)stream, device);
  static const int held = [] { const char* e = getenv("C4_TOWER64_HELD"); return e ? atoi(e) : 0; }();   // round 2's 64-channel kernel (a layer's weights held in registers)
  if (channels == 64 && held) return launch_tower<64, 8, 512, 2>(p, n_boards, (hipStream_t)stream, device);
#endif
  if (channels == 32)
    return launch_tower<32, 16, 512, 1>(p, n_boards, (hipStream_t)stream, device);   // 8 waves: two per SIMD
  // round 6, the narrow launches of a 64-channel job's tail -- 5: 4 boards on 8 wavefronts, 6: 2 boards on 4 wavefronts (pairs split the
  // output channels as in the default).  Alone under rocprofv3, 8-block tower (profiles/r06_tower_small.txt): 256 boards 80.9 us (8 boards
  // per workgroup) -> 48.0 -> 36.2; 512: 83.2 -> 51.3 -> 39.4; 768: 86.3 -> 56.2 -> 56.7; 1 024: 87.9 -> 60.1 -> 61.5; 1 280: 92.5 against 100.6 / 92.1
  if (channels == 64 && (config == 5 || (config == 0 && n_boards > 512 && n_boards <= 1024))) return launch_tower<64, 4, 512, 2, true>(p, n_boards, (hipStream_t)stream, device);
  if (channels == 64 && (config == 6 || (config == 0 && n_boards <= 512))) return launch_tower<64, 2, 256, 2, true>(p, n_boards, (hipStream_t)stream, device);
  // 64 channels, round 5: FOUR wavefronts, one per SIMD, each with all 64 output channels of two whole boards (no hand-over between
  // layers at all) and a weight ring 6 / 3 k-steps deep in its 512 registers (config 2 / 3)
  if (channels == 64 && config == 2) return launch_tower<64, 8, 256, 1, true, 6>(p, n_boards, (hipStream_t)stream, device);
  if (channels == 64 && config == 3) return launch_tower<64, 8, 256, 1, true, 3>(p, n_boards, (hipStream_t)stream, device);
  // config 4 (round 5): the 8-wavefront kernel with the weights through an LDS ring, fetched once per workgroup four taps ahead
  // (as long as the layers' biases fit behind it: up to 23 residual blocks)
  if (channels == 64 && config == 4 && Geo<64, 8>::kLdsBytes + 64 + 4 * RingFeed<64>::kStageSlots * 16 + 2 * (int)n_blocks * 64 * 4 <= 160 * 1024)
    return launch_tower<64, 8, 512, 2, true, 2, true>(p, n_boards, (hipStream_t)stream, device);
  return launch_tower<64, 8, 512, 2, true>(p, n_boards, (hipStream_t)stream, device);   // 8 wavefronts: pairs split the output channels; weights streamed
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// Output layers of both heads in one launch (nn.py:84-85, 98-99): policy Linear(F -> 7) +
// LogSoftmax and value Linear(F -> 2) + Tanh, reading the two hidden activations once and
// writing straight into the tensors the step kernel is bound to.  F = 42 * C is 1344 or 2688:
// nine dot products per board, HBM-bound on the 2 x F bf16 activations per board.
// One wavefront per board; lanes stride the feature dimension with 16-byte loads.
// ------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// kBPW boards per wavefront share every weight chunk load (9 x 16 bytes per lane and iteration)
template <int kBPW>
__global__ __launch_bounds__(256) void c4_head_out_kernel(const uint4* __restrict__ hp, const uint4* __restrict__ hv,
                                                          const uint4* __restrict__ wp, const uint4* __restrict__ wv,
                                                          const float* __restrict__ bp, const float* __restrict__ bv,
                                                          uint32_t n_boards, uint32_t f8, uint32_t sp8, uint32_t sv8,
                                                          float* __restrict__ logprobs, float* __restrict__ q) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t g0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kBPW;
  if (g0 >= n_boards) return;
  float acc[kBPW][9];
#pragma unroll
  for (int b = 0; b < kBPW; b++)
#pragma unroll
    for (int o = 0; o < 9; o++) acc[b][o] = 0.f;
  for (uint32_t i = lane; i < f8; i += 64) {
    uint32_t xw[kBPW][2][4];
#pragma unroll
    for (int b = 0; b < kBPW; b++) {
      const uint32_t g = (g0 + b < n_boards) ? g0 + b : g0;   // tail boards recompute board g0 (never stored)
      const uint4 xp = hp[(size_t)g * sp8 + i], xv = hv[(size_t)g * sv8 + i];
      xw[b][0][0] = xp.x; xw[b][0][1] = xp.y; xw[b][0][2] = xp.z; xw[b][0][3] = xp.w;
      xw[b][1][0] = xv.x; xw[b][1][1] = xv.y; xw[b][1][2] = xv.z; xw[b][1][3] = xv.w;
    }
#pragma unroll
    for (int o = 0; o < 9; o++) {
      const uint4 w = (o < 7) ? wp[(size_t)o * f8 + i] : wv[(size_t)(o - 7) * f8 + i];
      const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int b = 0; b < kBPW; b++)
#pragma unroll
        for (int j = 0; j < 4; j++)   // v_dot2c_f32_bf16: two bf16 products accumulated in f32
          acc[b][o] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, xw[b][o < 7 ? 0 : 1][j]), __builtin_bit_cast(bf16x2, ww[j]), acc[b][o], false);
    }
  }
#pragma unroll
  for (int b = 0; b < kBPW; b++) {
    float v[9];
#pragma unroll
    for (int o = 0; o < 9; o++) v[o] = wave_sum(acc[b][o]) + (o < 7 ? bp[o] : bv[o - 7]);
    const uint32_t g = g0 + b;
    if (lane == 0 && g < n_boards) {
      float mx = v[0];
#pragma unroll
      for (int o = 1; o < 7; o++) mx = fmaxf(mx, v[o]);
      float s = 0.f;
#pragma unroll
      for (int o = 0; o < 7; o++) s += expf(v[o] - mx);
      const float lse = mx + logf(s);
#pragma unroll
      for (int o = 0; o < 7; o++) logprobs[(size_t)g * 7 + o] = v[o] - lse;
      q[(size_t)g * 2 + 0] = tanhf(v[7]);
      q[(size_t)g * 2 + 1] = tanhf(v[8]);
    }
  }
}


// MFMA form of the same computation (used when F is a multiple of 32 * 7 * 6, i.e. F = 42 * C): c4_head_out.hpp, shared with
// the fused output + step kernel of c4_session.hip.
using c4ho::kHeadSteps;
using c4ho::kHeadWaves;
template <int kRows>
__global__ __launch_bounds__(64 * kHeadWaves, 1) void c4_head_out_mfma_kernel(
    const uint4* __restrict__ hp, const uint4* __restrict__ hv, const uint4* __restrict__ wp, const uint4* __restrict__ wv,
    const float* __restrict__ bp, const float* __restrict__ bv, uint32_t n_boards, uint32_t f8, uint32_t sp8, uint32_t sv8,
    float* __restrict__ logprobs, float* __restrict__ q) {
  __shared__ c4ho::Shared sh;
  c4ho::head_out_block<kRows>(sh, hp, hv, wp, wv, bp, bv, n_boards, f8, sp8, sv8, logprobs, q, blockIdx.x);
}

}  // namespace

namespace {
// float32 -> bf16, round to nearest even (what torch's .to(bfloat16) does; NaN -> the quiet NaN 0x7FC0)
__device__ __forceinline__ uint32_t bf16_rne(float f) {
  const uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0u;
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// 8 values per thread: two 16-byte reads (over PCIe when the batch lives in pinned host memory), one 16-byte store;
// boards past the batch's own count (a captured launch covers a whole bucket of rows) become empty boards
__global__ __launch_bounds__(256) void c4_planes_from_f32_kernel(const c4_f32_batch* __restrict__ slot, const float* __restrict__ src,
                                                                 uint32_t n_valid, uint4* __restrict__ dst, uint32_t n8) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  if (slot) { src = slot->data; n_valid = slot->n_boards; }
  uint4 o = make_uint4(0, 0, 0, 0);
  if (8 * (size_t)i < 84 * (size_t)n_valid) {     // 84 values per board: an odd board count ends inside a thread's eight
    const float4* s = reinterpret_cast<const float4*>(src) + 2 * (size_t)i;
    float4 a = s[0], b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (8 * (size_t)i + 4 < 84 * (size_t)n_valid) b = s[1];
    o.x = bf16_rne(a.x) | (bf16_rne(a.y) << 16);
    o.y = bf16_rne(a.z) | (bf16_rne(a.w) << 16);
    o.z = bf16_rne(b.x) | (bf16_rne(b.y) << 16);
    o.w = bf16_rne(b.z) | (bf16_rne(b.w) << 16);
  }
  dst[i] = o;
}
}  // namespace

extern "C" int c4_planes_from_f32(const c4_f32_batch* batch_slot, const float* src, uint32_t n_boards, void* planes_dev, uint32_t n_rows_out,
                                  void* stream) {
  if ((!batch_slot && !src && n_boards) || !planes_dev) return c4host::fail(C4_ERR_BAD_ARG, "c4_planes_from_f32: null argument");
  if (((uintptr_t)src | (uintptr_t)planes_dev | (uintptr_t)batch_slot) & 15) return c4host::fail(C4_ERR_BAD_ARG, "c4_planes_from_f32: arrays must be 16-byte aligned");
  if (n_rows_out % 2) return c4host::fail(C4_ERR_BAD_ARG, "c4_planes_from_f32: n_rows_out must be even (84 values per board, 8 per thread)");
  if (!batch_slot && n_boards > n_rows_out) return c4host::fail(C4_ERR_BAD_ARG, "c4_planes_from_f32: n_boards > n_rows_out");
  if (n_rows_out == 0) return C4_OK;
  const int device = c4host::stream_device((hipStream_t)stream);
  c4host::DeviceGuard guard(device);
  if (guard.error() != hipSuccess) return c4host::fail(C4_ERR_HIP, std::string("c4_planes_from_f32: hipSetDevice: ") + hipGetErrorString(guard.error()));
  const uint32_t n8 = n_rows_out / 2 * 21;     // 84 n / 8
  c4_planes_from_f32_kernel<<<dim3((n8 + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(batch_slot, src, n_boards, (uint4*)planes_dev, n8);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return c4host::fail(C4_ERR_HIP, std::string("c4_planes_from_f32 launch: ") + hipGetErrorString(e));
  return C4_OK;
}

extern "C" int c4_head_out_bf16(const void* hidden_policy_dev, const void* hidden_value_dev, const void* w_policy_dev,
                                const void* w_value_dev, const float* b_policy_dev, const float* b_value_dev,
                                uint32_t n_boards, uint32_t features, uint32_t policy_row_stride, uint32_t value_row_stride,
                                float* logprobs_dev, float* q_dev, void* stream) {
  if (!hidden_policy_dev || !hidden_value_dev || !w_policy_dev || !w_value_dev || !b_policy_dev || !b_value_dev || !logprobs_dev || !q_dev)
    return c4host::fail(C4_ERR_BAD_ARG, "c4_head_out_bf16: null argument");
  if (features % 8 != 0 || policy_row_stride % 8 != 0 || value_row_stride % 8 != 0)
    return c4host::fail(C4_ERR_BAD_ARG, "c4_head_out_bf16: features and row strides must be multiples of 8 elements");
  if (n_boards == 0) return C4_OK;
  const int device = c4host::stream_device((hipStream_t)stream);
  c4host::DeviceGuard guard(device);
  if (guard.error() != hipSuccess) return c4host::fail(C4_ERR_HIP, std::string("c4_head_out_bf16: hipSetDevice: ") + hipGetErrorString(guard.error()));
  if ((features / 8) % (4 * kHeadSteps * kHeadWaves) == 0) {
    // (8 boards per workgroup -- twice the workgroups for launches that leave half the CUs without one -- was
    // measured too: 6.1 vs 6.0 us at 2 048 boards, no gain; the template parameter stays for the next experiment)
    c4_head_out_mfma_kernel<16><<<dim3((n_boards + 15) / 16), dim3(64 * kHeadWaves), 0, (hipStream_t)stream>>>(
          (const uint4*)hidden_policy_dev, (const uint4*)hidden_value_dev, (const uint4*)w_policy_dev, (const uint4*)w_value_dev,
          b_policy_dev, b_value_dev, n_boards, features / 8, policy_row_stride / 8, value_row_stride / 8, logprobs_dev, q_dev);
  } else {
    constexpr int kBoardsPerWave = 2;   // other feature counts: dot-product form (measured best of 1 / 2 / 4 boards per wavefront)
    c4_head_out_kernel<kBoardsPerWave><<<dim3((n_boards + 4 * kBoardsPerWave - 1) / (4 * kBoardsPerWave)), dim3(256), 0, (hipStream_t)stream>>>(
        (const uint4*)hidden_policy_dev, (const uint4*)hidden_value_dev, (const uint4*)w_policy_dev, (const uint4*)w_value_dev,
        b_policy_dev, b_value_dev, n_boards, features / 8, policy_row_stride / 8, value_row_stride / 8, logprobs_dev, q_dev);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return c4host::fail(C4_ERR_HIP, std::string("c4_head_out_bf16 launch: ") + hipGetErrorString(e));
  return C4_OK;
}
