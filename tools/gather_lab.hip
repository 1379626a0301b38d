// gather_lab.hip -- what the memory system gives the tree walk's access pattern (a tool, not product).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_lab tools/gather_lab.hip && /tmp/gather_lab
//
// The step kernel's select/backup touch ONE 128-byte line per tree level, read by the 8 lanes of a
// game's lane group (16 bytes each), at an address that depends on the previous level.  Two
// questions, each as a function of the table size (L2 / Infinity Cache / HBM resident) and of the
// number of wavefronts per SIMD:
//   independent  -- random 128-byte lines, `ILP` independent lines in flight per lane group:
//                   the random-line THROUGHPUT ceiling (what a level-synchronous walk could reach)
//   chase        -- each lane group follows a chain of random lines (next index read from the
//                   line itself): the per-level LATENCY a select pays, idle and under load
// Prints one JSON object per configuration.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// every line's first dword = index of the next line of its chain (a random permutation step)
__global__ void k_fill(uint4* table, uint32_t n_lines, uint32_t seed) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_lines * 8u) return;
  const uint32_t line = i >> 3, sub = i & 7;
  const uint32_t nxt = (uint32_t)(((uint64_t)mix(line * 2654435761u + seed) * n_lines) >> 32);
  table[i] = make_uint4(sub == 0 ? nxt : line, line, sub, 0x1234);
}

template <int ILP>
__global__ __launch_bounds__(64) void k_independent(const uint4* __restrict__ table, uint32_t n_lines, uint32_t iters, uint32_t seed,
                                                    uint32_t* sink) {
  const uint32_t lane = threadIdx.x, sub = lane & 7, grp = (blockIdx.x * 64 + lane) >> 3;
  uint32_t acc = 0;
  uint32_t h = mix(grp * 0x9E3779B9u + seed);
  for (uint32_t it = 0; it < iters; it++) {
    uint4 v[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) {
      h = mix(h + 0x68bc21ebu * (k + 1));
      const uint32_t line = (uint32_t)(((uint64_t)h * n_lines) >> 32);
      v[k] = table[(size_t)line * 8 + sub];
    }
#pragma unroll
    for (int k = 0; k < ILP; k++) acc += v[k].x ^ v[k].w;
  }
  if (acc == 0xdeadbeef) sink[0] = acc;
}

__global__ __launch_bounds__(64) void k_chase(const uint4* __restrict__ table, uint32_t n_lines, uint32_t depth, uint32_t seed, uint32_t* sink) {
  const uint32_t lane = threadIdx.x, sub = lane & 7, grp = (blockIdx.x * 64 + lane) >> 3;
  uint32_t line = (uint32_t)(((uint64_t)mix(grp * 0x9E3779B9u + seed) * n_lines) >> 32);
  uint32_t acc = 0;
  for (uint32_t d = 0; d < depth; d++) {
    const uint4 v = table[(size_t)line * 8 + sub];
    acc += v.y;
    line = (uint32_t)__shfl((int)v.x, (int)(lane & ~7u), 64);   // the next line comes out of this one (lane 0 of the group)
  }
  if (acc == 0xdeadbeef) sink[0] = acc;
}

// the backup's pattern: every LANE updates one 16-byte entry of its own random line (read 12 bytes,
// add, write back): 64 lines per wavefront instruction, a partial-line write each
__global__ __launch_bounds__(64) void k_rmw16(uint4* table, uint32_t n_lines, uint32_t iters, uint32_t seed) {
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  uint32_t h = mix(t * 0x9E3779B9u + seed);
  for (uint32_t it = 0; it < iters; it++) {
    h = mix(h + 0x68bc21ebu);
    const uint32_t line = (uint32_t)(((uint64_t)h * n_lines) >> 32);
    uint4* e = table + (size_t)line * 8 + (h & 7u) % 7u;
    uint4 v = *e;
    v.y += 1; v.z += it; v.w ^= h;       // x (the chain index of the other tests) stays
    *e = v;
  }
}

// the expand's pattern: a lane group writes one whole random 128-byte line
__global__ __launch_bounds__(64) void k_write128(uint4* table, uint32_t n_lines, uint32_t iters, uint32_t seed) {
  const uint32_t lane = threadIdx.x, sub = lane & 7, grp = (blockIdx.x * 64 + lane) >> 3;
  uint32_t h = mix(grp * 0x9E3779B9u + seed);
  for (uint32_t it = 0; it < iters; it++) {
    h = mix(h + 0x68bc21ebu);
    const uint32_t line = (uint32_t)(((uint64_t)h * n_lines) >> 32);
    table[(size_t)line * 8 + sub] = make_uint4(line, it, sub, h);
  }
}

static float time_ms(hipEvent_t a, hipEvent_t b) {
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && atoi(argv[1]) == 1;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  uint32_t* sink;
  CHECK(hipMalloc(&sink, 4));
  const size_t sizes_mb[] = {16, 53, 200, 650, 2300, 18000};
  for (size_t mb : sizes_mb) {
    if (quick && mb > 2300) continue;
    const uint32_t n_lines = (uint32_t)(mb * 1000000ull / 128);
    uint4* table;
    CHECK(hipMalloc(&table, (size_t)n_lines * 128));
    k_fill<<<(n_lines * 8u + 255) / 256, 256>>>(table, n_lines, 7);
    CHECK(hipDeviceSynchronize());
    // independent lines: waves per SIMD 1, 4, 8 (grid = waves_per_simd * 1024 wavefronts of 8 lane groups)
    for (int wps : {1, 4, 8}) {
      for (int ilp : {1, 4}) {
        const uint32_t grid = 1024u * wps, iters = 2000 / ilp;
        auto launch = [&](uint32_t seed) {
          if (ilp == 1) k_independent<1><<<grid, 64>>>(table, n_lines, iters, seed, sink);
          else k_independent<4><<<grid, 64>>>(table, n_lines, iters, seed, sink);
        };
        launch(1);
        CHECK(hipEventRecord(e0));
        launch(2);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        const double ms = time_ms(e0, e1);
        const double lines = (double)grid * 8 * iters * ilp;
        printf("{\"test\": \"independent\", \"table_MB\": %zu, \"waves_per_simd\": %d, \"lines_in_flight_per_group\": %d, \"ms\": %.4f, \"TBps\": %.3f, \"G_lines_per_s\": %.3f}\n",
               mb, wps, ilp, ms, lines * 128 / ms / 1e9, lines / ms / 1e6);
      }
    }
    // dependent chain: 256 wavefronts (one per CU: the C2 launch shape, idle chip), then 1, 3, 8 wavefronts per SIMD
    for (int waves : {256, 1024, 3072, 8192}) {
      const uint32_t depth = 64;
      k_chase<<<waves, 64>>>(table, n_lines, depth, 1, sink);
      CHECK(hipEventRecord(e0));
      k_chase<<<waves, 64>>>(table, n_lines, depth, 2, sink);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      const double ms = time_ms(e0, e1);
      printf("{\"test\": \"chase\", \"table_MB\": %zu, \"wavefronts\": %d, \"depth\": %u, \"ms\": %.4f, \"ns_per_level\": %.1f, \"TBps\": %.3f}\n",
             mb, waves, depth, ms, ms * 1e6 / depth, (double)waves * 8 * depth * 128 / ms / 1e9);
    }
    // writes last (they destroy the chains): 16-byte read-modify-writes and whole-line writes
    for (int wps : {1, 4, 8}) {
      const uint32_t grid = 1024u * wps, iters = 500;
      k_rmw16<<<grid, 64>>>(table, n_lines, iters, 1);
      CHECK(hipEventRecord(e0));
      k_rmw16<<<grid, 64>>>(table, n_lines, iters, 2);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      double ms = time_ms(e0, e1);
      double ops = (double)grid * 64 * iters;
      printf("{\"test\": \"rmw16\", \"table_MB\": %zu, \"waves_per_simd\": %d, \"ms\": %.4f, \"G_updates_per_s\": %.3f, \"algorithmic_TBps\": %.3f}\n",
             mb, wps, ms, ops / ms / 1e6, ops * 32 / ms / 1e9);
      k_write128<<<grid, 64>>>(table, n_lines, iters, 3);
      CHECK(hipEventRecord(e0));
      k_write128<<<grid, 64>>>(table, n_lines, iters, 4);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      ms = time_ms(e0, e1);
      ops = (double)grid * 8 * iters;
      printf("{\"test\": \"write128\", \"table_MB\": %zu, \"waves_per_simd\": %d, \"ms\": %.4f, \"G_lines_per_s\": %.3f, \"TBps\": %.3f}\n",
             mb, wps, ms, ops / ms / 1e6, ops * 128 / ms / 1e9);
    }
    CHECK(hipFree(table));
    fflush(stdout);
  }
  return 0;
}
