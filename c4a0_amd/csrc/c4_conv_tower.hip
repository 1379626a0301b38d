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
//     padded to 8 columns (one shared zero column between rows) plus zero rows above and below,
//     so a 3x3 tap is a constant slot offset 8*dr + dc and needs no bounds test.
//   * GEMM orientation: D[co, cell] = sum_k W[co, k] * X[k, cell] with v_mfma_f32_16x16x32_bf16:
//     A = weights (held in registers for the whole layer), B = activations, one k-step = one tap
//     x 32 input channels = one ds_read_b128 per lane.  A tile is 16 consecutive padded cells
//     (two board rows): 16 distinct 16-byte slots per channel group -> bank-conflict free.
//   * D leaves the MFMA as 4 consecutive output channels per lane for one cell: bias, ReLU and
//     the residual add happen in registers and go back to LDS as one 8-byte store.
//   * 4 waves per workgroup (one per SIMD), each owning 1/4 of the tiles of every layer;
//     two tiles are in flight per wave so consecutive MFMAs never wait on their accumulator.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/c4a0_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPCS = 72;      // padded slots per board per channel group (66 used)
constexpr int kTilesPerBoard = 3;

// slot of board cell (row r, column c): rows 1..6 of an 8-wide padded image, +1 so that the
// (-1,-1) tap of the first tile lane stays inside the board's own region
__device__ __forceinline__ int cell_slot(int r, int c) { return (r + 1) * 8 + (c + 1) + 1; }

template <int C, int NB>
struct Geo {
  static constexpr int KG = C / 8;                 // channel groups of 8 (16-byte slots)
  static constexpr int MT = C / 16;                // output-channel tiles of 16
  static constexpr int KC = C / 32;                // k-steps per tap
  static constexpr int kPlane = NB * kPCS;         // slots per channel-group plane
  static constexpr int kBufSlots = KG * kPlane;    // slots per activation buffer
  static constexpr int kLdsBytes = 2 * kBufSlots * 16;
  static constexpr int kTiles = NB * kTilesPerBoard;
  static_assert((kPlane * 16) % 256 == 0, "channel-group planes must keep the bank phase");
  static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
};

struct TowerParams {
  const uint16_t* planes;   // [G][2][42] bf16
  const bf16x8* w0;         // conv0 fragments  [3 steps][MT][64 lanes]
  const bf16x8* w;          // block conv fragments [2*n_blocks][9 taps][MT][KC][64 lanes]
  const float* bias;        // [1 + 2*n_blocks][C]
  uint16_t* out;            // [G][42][C] bf16
  uint32_t n_boards;
  uint32_t n_blocks;
};

template <int C, int NB>
__global__ __launch_bounds__(256) void c4_conv_tower_kernel(TowerParams p) {
  using G = Geo<C, NB>;
  extern __shared__ __attribute__((aligned(256))) uint8_t lds_raw[];
  uint4* X = reinterpret_cast<uint4*>(lds_raw);    // block input / residual stream
  uint4* T = X + G::kBufSlots;                     // intermediate (and the conv0 input image)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 15;   // cell within the tile / MFMA column
  const int lg = lane >> 4;   // MFMA k-group (0..3)
  const uint32_t board0 = blockIdx.x * NB;

  // ---- zero both images (halo cells stay zero for the whole kernel) ----
  for (int i = tid; i < 2 * G::kBufSlots; i += 256) X[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // ---- stage the input planes: channel group 0 of T holds {plane0, plane1, 0 x6} per cell ----
  for (int i = tid; i < NB * 42; i += 256) {
    const int b = i / 42, cell = i - b * 42;
    const uint32_t g = board0 + b;
    if (g < p.n_boards) {
      const uint32_t v0 = p.planes[(size_t)g * 84 + cell];
      const uint32_t v1 = p.planes[(size_t)g * 84 + 42 + cell];
      T[b * kPCS + cell_slot(cell / 7, cell % 7)] = make_uint4(v0 | (v1 << 16), 0, 0, 0);
    }
  }
  __syncthreads();

  const int n_layers = 1 + 2 * (int)p.n_blocks;
  // tiles of this wave: contiguous range, processed two at a time
  constexpr int kTilesPerWave = (G::kTiles + 3) / 4;
  const int tile_lo = wave * kTilesPerWave;
  const int tile_hi = (tile_lo + kTilesPerWave < G::kTiles) ? tile_lo + kTilesPerWave : G::kTiles;

  for (int layer = 0; layer < n_layers; layer++) {
    const bool is_conv0 = layer == 0;
    const bool is_second = !is_conv0 && ((layer & 1) == 0);   // layers 2,4,..: second conv of a block
    // source / destination images: conv0: T(input) -> X ; first conv: X -> T ; second conv: T -> X (+= residual)
    const uint4* src = (is_conv0 || is_second) ? T : X;
    uint4* dst = (is_conv0 || is_second) ? X : T;

    // ---- this layer's weights and bias into registers ----
    bf16x8 wf[9][G::MT][G::KC];
    if (is_conv0) {
#pragma unroll
      for (int s = 0; s < 3; s++)
#pragma unroll
        for (int m = 0; m < G::MT; m++) wf[s][m][0] = p.w0[(s * G::MT + m) * 64 + lane];
    } else {
      const bf16x8* wl = p.w + (size_t)(layer - 1) * 9 * G::MT * G::KC * 64;
#pragma unroll
      for (int t = 0; t < 9; t++)
#pragma unroll
        for (int m = 0; m < G::MT; m++)
#pragma unroll
          for (int kc = 0; kc < G::KC; kc++) wf[t][m][kc] = wl[((t * G::MT + m) * G::KC + kc) * 64 + lane];
    }
    f32x4 bias4[G::MT];
#pragma unroll
    for (int m = 0; m < G::MT; m++) {
      const float* bp = p.bias + (size_t)layer * C + 16 * m + 4 * lg;
      bias4[m] = f32x4{bp[0], bp[1], bp[2], bp[3]};
    }

    for (int tile = tile_lo; tile < tile_hi; tile += 2) {
      const bool two = tile + 1 < tile_hi;
      int bidx[2], slot[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int tl = (u == 1 && !two) ? tile : tile + u;
        bidx[u] = tl / kTilesPerBoard;
        slot[u] = 9 + 16 * (tl - bidx[u] * kTilesPerBoard) + li;   // cell_slot of the tile's first cell is 9 + 16 j
      }
      f32x4 acc[2][G::MT];
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int m = 0; m < G::MT; m++) acc[u][m] = f32x4{0.f, 0.f, 0.f, 0.f};

      if (is_conv0) {
        // k-step s, k-group lg <-> tap 4 s + lg (taps >= 9 have zero weights); 8 "channels" per tap, 2 real
#pragma unroll
        for (int s = 0; s < 3; s++) {
          int tap = 4 * s + lg;
          tap = tap < 9 ? tap : 4;
          const int d = 8 * (tap / 3 - 1) + (tap % 3 - 1);
#pragma unroll
          for (int u = 0; u < 2; u++) {
            const uint4 raw = src[bidx[u] * kPCS + slot[u] + d];
            const bf16x8 bf = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
            for (int m = 0; m < G::MT; m++)
              acc[u][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][m][0], bf, acc[u][m], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < 9; t++) {
          const int d = 8 * (t / 3 - 1) + (t % 3 - 1);
#pragma unroll
          for (int kc = 0; kc < G::KC; kc++) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
              const uint4 raw = src[((4 * kc + lg) * NB + bidx[u]) * kPCS + slot[u] + d];
              const bf16x8 bf = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
              for (int m = 0; m < G::MT; m++)
                acc[u][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][m][kc], bf, acc[u][m], 0, 0, 0);
            }
          }
        }
      }

      // ---- epilogue: lane holds output channels 16 m + 4 lg + {0..3} of cell `slot` ----
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (u == 1 && !two) break;
        const bool valid = ((slot[u] - 1) & 7) != 0;   // padded column 0 is halo
        if (!valid) continue;
#pragma unroll
        for (int m = 0; m < G::MT; m++) {
          f32x4 v = acc[u][m] + bias4[m];
          // 8-byte half of the 16-byte slot of channel group 2 m + lg/2
          uint2* dp = reinterpret_cast<uint2*>(&dst[((2 * m + (lg >> 1)) * NB + bidx[u]) * kPCS + slot[u]]) + (lg & 1);
          if (is_second) {
            const bf16x4 old = __builtin_bit_cast(bf16x4, *dp);
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = (float)old[r] + (v[r] > 0.f ? v[r] : 0.f);
          }
          const bf16x4 o = __builtin_convertvector(v, bf16x4);
          *dp = __builtin_bit_cast(uint2, o);
        }
      }
    }
    __syncthreads();
  }

  // ---- X -> out[g][cell][C] ----
  for (int i = tid; i < NB * 42 * G::KG; i += 256) {
    const int kg = i % G::KG;
    const int bc = i / G::KG;
    const int b = bc / 42, cell = bc - b * 42;
    const uint32_t g = board0 + b;
    if (g < p.n_boards) {
      const uint4 v = X[(kg * NB + b) * kPCS + cell_slot(cell / 7, cell % 7)];
      reinterpret_cast<uint4*>(p.out)[((size_t)g * 42 + cell) * G::KG + kg] = v;
    }
  }
}

thread_local std::string g_tower_error;

}  // namespace

extern "C" {

// Layouts (prepared by c4a0_amd/nn.py::pack_tower_weights):
//   w0_dev   bf16 [3][C/16][64][8] : conv0, k-step s / lane k-group g <-> tap 4 s + g, element j <-> input channel j (2 real)
//   w_dev    bf16 [2*n_blocks][9][C/16][C/32][64][8] : lane l element j = W[co = 16 m + (l & 15)][ci = 32 kc + 8 (l >> 4) + j][tap]
//   bias_dev f32  [1 + 2*n_blocks][C]
//   out_dev  bf16 [n_boards][42][C]   (cell-major, channels last)
int c4_conv_tower_bf16(const void* planes_dev, const void* w0_dev, const void* w_dev, const float* bias_dev,
                       uint32_t n_boards, uint32_t channels, uint32_t n_blocks, void* out_dev, void* stream) {
  if (!planes_dev || !w0_dev || !bias_dev || !out_dev || (n_blocks && !w_dev)) return C4_ERR_BAD_ARG;
  if (n_boards == 0) return C4_OK;
  TowerParams p{(const uint16_t*)planes_dev, (const bf16x8*)w0_dev, (const bf16x8*)w_dev, bias_dev, (uint16_t*)out_dev, n_boards, n_blocks};
  hipError_t e = hipSuccess;
  if (channels == 32) {
    constexpr int NB = 16;
    constexpr int kLds = Geo<32, NB>::kLdsBytes;
    auto k = c4_conv_tower_kernel<32, NB>;
    e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e == hipSuccess) {
      k<<<dim3((n_boards + NB - 1) / NB), dim3(256), kLds, (hipStream_t)stream>>>(p);
      e = hipGetLastError();
    }
  } else if (channels == 64) {
    constexpr int NB = 8;
    constexpr int kLds = Geo<64, NB>::kLdsBytes;
    auto k = c4_conv_tower_kernel<64, NB>;
    e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e == hipSuccess) {
      k<<<dim3((n_boards + NB - 1) / NB), dim3(256), kLds, (hipStream_t)stream>>>(p);
      e = hipGetLastError();
    }
  } else {
    return C4_ERR_BAD_ARG;
  }
  return e == hipSuccess ? C4_OK : C4_ERR_HIP;
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

__global__ __launch_bounds__(256) void c4_head_out_kernel(const uint4* __restrict__ hp, const uint4* __restrict__ hv,
                                                          const uint4* __restrict__ wp, const uint4* __restrict__ wv,
                                                          const float* __restrict__ bp, const float* __restrict__ bv,
                                                          uint32_t n_boards, uint32_t f8, float* __restrict__ logprobs,
                                                          float* __restrict__ q) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= n_boards) return;
  float accp[7] = {0, 0, 0, 0, 0, 0, 0}, accv[2] = {0, 0};
  for (uint32_t i = lane; i < f8; i += 64) {
    const bf16x8 xp = __builtin_bit_cast(bf16x8, hp[(size_t)g * f8 + i]);
    const bf16x8 xv = __builtin_bit_cast(bf16x8, hv[(size_t)g * f8 + i]);
#pragma unroll
    for (int o = 0; o < 7; o++) {
      const bf16x8 w = __builtin_bit_cast(bf16x8, wp[(size_t)o * f8 + i]);
#pragma unroll
      for (int j = 0; j < 8; j++) accp[o] += (float)xp[j] * (float)w[j];
    }
#pragma unroll
    for (int o = 0; o < 2; o++) {
      const bf16x8 w = __builtin_bit_cast(bf16x8, wv[(size_t)o * f8 + i]);
#pragma unroll
      for (int j = 0; j < 8; j++) accv[o] += (float)xv[j] * (float)w[j];
    }
  }
#pragma unroll
  for (int o = 0; o < 7; o++) accp[o] = wave_sum(accp[o]) + bp[o];
#pragma unroll
  for (int o = 0; o < 2; o++) accv[o] = wave_sum(accv[o]) + bv[o];
  if (lane == 0) {
    float mx = accp[0];
#pragma unroll
    for (int o = 1; o < 7; o++) mx = fmaxf(mx, accp[o]);
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < 7; o++) s += expf(accp[o] - mx);
    const float lse = mx + logf(s);
#pragma unroll
    for (int o = 0; o < 7; o++) logprobs[(size_t)g * 7 + o] = accp[o] - lse;
    q[(size_t)g * 2 + 0] = tanhf(accv[0]);
    q[(size_t)g * 2 + 1] = tanhf(accv[1]);
  }
}

}  // namespace

extern "C" int c4_head_out_bf16(const void* hidden_policy_dev, const void* hidden_value_dev, const void* w_policy_dev,
                                const void* w_value_dev, const float* b_policy_dev, const float* b_value_dev,
                                uint32_t n_boards, uint32_t features, float* logprobs_dev, float* q_dev, void* stream) {
  if (!hidden_policy_dev || !hidden_value_dev || !w_policy_dev || !w_value_dev || !b_policy_dev || !b_value_dev || !logprobs_dev || !q_dev)
    return C4_ERR_BAD_ARG;
  if (features % 8 != 0) return C4_ERR_BAD_ARG;
  if (n_boards == 0) return C4_OK;
  c4_head_out_kernel<<<dim3((n_boards + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(
      (const uint4*)hidden_policy_dev, (const uint4*)hidden_value_dev, (const uint4*)w_policy_dev, (const uint4*)w_value_dev,
      b_policy_dev, b_value_dev, n_boards, features / 8, logprobs_dev, q_dev);
  return hipGetLastError() == hipSuccess ? C4_OK : C4_ERR_HIP;
}
