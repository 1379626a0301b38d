// c4_head_out.hpp -- the output layers of both heads for the 16 boards of one workgroup (reference src/c4a0/nn.py:84-85, 98-99:
// policy Linear(F -> 7) + LogSoftmax, value Linear(F -> 2) + Tanh), shared by the stand-alone kernel (c4_conv_tower.hip,
// c4_head_out_bf16) and by the fused output + step kernel (c4_session.hip, c4_session_step_head_out): ONE piece of code, so
// that both paths give a board the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace c4ho {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// a workgroup owns 16 boards; D[board][output] = sum_k X[board][k] * W[output][k] with
// v_mfma_f32_16x16x32_bf16 (A = 16 boards x 32 features straight from global memory, B = the 7 or 2
// weight rows, zero-padded to 16 columns).  The feature dimension is split over 6 wavefronts,
// every wavefront requests all of its operands before the first MFMA (one memory round trip), and
// the partial tiles meet in LDS.  No cross-lane reduction per output, 16-byte loads only.
constexpr int kHeadWaves = 6, kHeadSteps = 7;

struct Shared {
  f32x4 part[kHeadWaves][2][64];
  float tile[2][16][17];
  float res[16][12];          // per board: 7 log-probabilities, q_penalty, q_no_penalty (what the fused step wavefronts read)
};

// kRows = boards per workgroup: 16 (every row of the MFMA tile a board) or 8 (rows 8..15 repeat rows 0..7).
// Called by all 64 * kHeadWaves threads of workgroup `block`.  On return the boards' outputs are in global memory and in
// sh.res (visible to the workgroup after the caller's next barrier).
template <int kRows>
__device__ __forceinline__ void head_out_block(Shared& sh, const uint4* __restrict__ hp, const uint4* __restrict__ hv, const uint4* __restrict__ wp,
                                               const uint4* __restrict__ wv, const float* __restrict__ bp, const float* __restrict__ bv, uint32_t n_boards,
                                               uint32_t f8, uint32_t sp8, uint32_t sv8, float* __restrict__ logprobs, float* __restrict__ q, uint32_t block) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t row = lane & 15, kq = lane >> 4;
  const uint32_t g0 = block * kRows;
  const uint32_t gr = g0 + (row & (kRows - 1));
  const uint32_t g = gr < n_boards ? gr : n_boards - 1;                   // tail rows recompute the last board (never stored)
  const uint4* xp = hp + (size_t)g * sp8 + kq;
  const uint4* xv = hv + (size_t)g * sv8 + kq;
  const uint4* wpl = wp + (size_t)(row < 7 ? row : 0) * f8 + kq;           // B column = output `row`
  const uint4* wvl = wv + (size_t)(row < 2 ? row : 0) * f8 + kq;
  f32x4 accp = {0.f, 0.f, 0.f, 0.f}, accv = {0.f, 0.f, 0.f, 0.f};
  const uint32_t n_iter = f8 / (4 * kHeadSteps * kHeadWaves);
  for (uint32_t it = 0; it < n_iter; it++) {
    const uint32_t s0 = (it * kHeadWaves + wave) * kHeadSteps;           // first k-step (of 32 features) of this wavefront
    uint4 a_p[kHeadSteps], a_v[kHeadSteps], b_p[kHeadSteps], b_v[kHeadSteps];
#pragma unroll
    for (int s = 0; s < kHeadSteps; s++) {
      a_p[s] = xp[4 * (s0 + s)];
      a_v[s] = xv[4 * (s0 + s)];
      b_p[s] = wpl[4 * (s0 + s)];      // every lane loads (lanes beyond the 7 / 2 outputs re-read row 0) and is masked AFTERWARDS:
      b_v[s] = wvl[4 * (s0 + s)];      // a "load or zero" select makes hipcc branch around each load and drain vmcnt per element
    }
    // (round 3: with the select in the loop above the 28 requests of a wavefront went out two at a time, each pair
    // waited for -- twelve serial memory round trips, 8.2 of the kernel's 8.6 us.  Now one round trip.)
    __builtin_amdgcn_sched_barrier(0);   // ... and the scheduler must not re-interleave loads and MFMAs to save registers
    const uint32_t mp = row < 7 ? 0xFFFFFFFFu : 0u, mv = row < 2 ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int s = 0; s < kHeadSteps; s++) {
      b_p[s].x &= mp; b_p[s].y &= mp; b_p[s].z &= mp; b_p[s].w &= mp;
      b_v[s].x &= mv; b_v[s].y &= mv; b_v[s].z &= mv; b_v[s].w &= mv;
    }
#pragma unroll
    for (int s = 0; s < kHeadSteps; s++) {
      accp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_p[s]), __builtin_bit_cast(bf16x8, b_p[s]), accp, 0, 0, 0);
      accv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_v[s]), __builtin_bit_cast(bf16x8, b_v[s]), accv, 0, 0, 0);
    }
  }
  sh.part[wave][0][lane] = accp;
  sh.part[wave][1][lane] = accv;
  __syncthreads();
  if (wave < 2) {
    // wavefront 0 finishes the policy tile, wavefront 1 the value tile: lane holds output column
    // `row` of boards 4 kq .. 4 kq + 3
    f32x4 sum = sh.part[0][wave][lane];
#pragma unroll
    for (int w = 1; w < kHeadWaves; w++) sum += sh.part[w][wave][lane];
    const float bias = wave == 0 ? (row < 7 ? bp[row] : 0.f) : (row < 2 ? bv[row] : 0.f);
#pragma unroll
    for (int r = 0; r < 4; r++) sh.tile[wave][4 * kq + r][row] = sum[r] + bias;
  }
  __syncthreads();
  if (threadIdx.x < kRows) {
    const uint32_t b = threadIdx.x, gb = g0 + b;
    float v[9];
#pragma unroll
    for (int o = 0; o < 7; o++) v[o] = sh.tile[0][b][o];
    v[7] = sh.tile[1][b][0];
    v[8] = sh.tile[1][b][1];
    float mx = v[0];
#pragma unroll
    for (int o = 1; o < 7; o++) mx = fmaxf(mx, v[o]);
    float sm = 0.f;
#pragma unroll
    for (int o = 0; o < 7; o++) sm += expf(v[o] - mx);
    const float lse = mx + logf(sm);
    const float q0 = tanhf(v[7]), q1 = tanhf(v[8]);
#pragma unroll
    for (int o = 0; o < 7; o++) sh.res[b][o] = v[o] - lse;
    sh.res[b][7] = q0;
    sh.res[b][8] = q1;
    if (gb < n_boards) {
#pragma unroll
      for (int o = 0; o < 7; o++) logprobs[(size_t)gb * 7 + o] = v[o] - lse;
      q[(size_t)gb * 2 + 0] = q0;
      q[(size_t)gb * 2 + 1] = q1;
    }
  }
}

}  // namespace c4ho
