// How fast can a wave stream a row-major f32 matrix straight into the MFMA
// A-operand layout (lane = (row i = l & 15, k-slot l >> 4), 16 bytes per lane:
// 16 rows x 64 contiguous bytes per load instruction) compared with the
// coalesced shape the LDS-DMA stages use (4 rows x 256 bytes)?
//   hipcc --offload-arch=gfx950 -O3 -o row_frag_stream row_frag_stream.hip
// Each wave owns 64 rows and walks all columns; U loads are issued before the
// first is consumed (the compiler waits for all of them: a batch, not a ring --
// several waves per SIMD cover the drain).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

constexpr int64_t N = 200000, LD = 8008;  // config 4: P = 8001 -> ld 8008

// MODE 0: 4 rows x 256 B per instruction; 1: 16 rows x 64 B; 2: 16 rows x 2 x 64 B
// (two instructions back to back take the two halves of 128 contiguous bytes)
template <int MODE, int U>
__global__ void stream_kernel(const float* __restrict__ X, float* out,
                              int64_t n_group) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int64_t n_wave = (int64_t)gridDim.x * (blockDim.x / 64);
  const float4* X4 = reinterpret_cast<const float4*>(X);
  const int64_t ldq = LD / 4;
  float acc = 0.f;
  for (int64_t g = gw; g < n_group; g += n_wave) {
    const int64_t row0 = g * 64;
    if (MODE == 0) {
      // a "step" = 64 rows x 64 columns = 16 instructions
      const int r_l = lane >> 4, q_l = lane & 15;
      for (int64_t q0 = 0; q0 + 16 <= ldq; q0 += 16) {
        for (int d0 = 0; d0 < 16; d0 += U) {
          float4 x[U];
#pragma unroll
          for (int u = 0; u < U; ++u)
            x[u] = X4[(row0 + 4 * (d0 + u) + r_l) * ldq + q0 + q_l];
#pragma unroll
          for (int u = 0; u < U; ++u) acc += (x[u].x + x[u].y) + (x[u].z + x[u].w);
        }
      }
    } else {
      const int i = lane & 15, k = lane >> 4;
      constexpr int QS = MODE == 1 ? 4 : 8;   // quads per row per step
      constexpr int PER = 4 * (QS / 4);       // instructions per step
      constexpr int STEPS = U / PER > 0 ? U / PER : 1;
      for (int64_t q0 = 0; q0 + QS * STEPS <= ldq; q0 += QS * STEPS) {
        float4 x[STEPS][PER];
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int h = 0; h < QS / 4; ++h)
              x[s][rt * (QS / 4) + h] =
                  X4[(row0 + 16 * rt + i) * ldq + q0 + QS * s +
                     (MODE == 1 ? k : 2 * k + h)];
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
#pragma unroll
          for (int u = 0; u < PER; ++u)
            acc += (x[s][u].x + x[s][u].y) + (x[s][u].z + x[s][u].w);
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int MODE, int U>
void run(const float* X, float* out, int waves_per_cu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int64_t n_group = N / 64;
  const int threads = 64 * waves_per_cu;
  stream_kernel<MODE, U><<<256, threads>>>(X, out, n_group);
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r)
    stream_kernel<MODE, U><<<256, threads>>>(X, out, n_group);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double bytes = (double)n_group * 64 * LD * 4;
  printf("mode %d (%s) U=%2d waves/CU=%2d: %.3f ms, %.0f GB/s\n", MODE,
         MODE == 0 ? "4 rows x 256 B" : MODE == 1 ? "16 rows x 64 B"
                                                  : "16 rows x 2 x 64 B",
         U, waves_per_cu, ms, bytes / (ms * 1e-3) / 1e9);
}

int main() {
  float* X;
  float* out;
  const size_t bytes = (size_t)N * LD * 4;
  hipMalloc(&X, bytes);
  hipMalloc(&out, 64);
  hipMemset(X, 0, bytes);
  for (int w : {4, 8, 16}) {
    run<0, 8>(X, out, w);
    run<0, 16>(X, out, w);
    run<1, 8>(X, out, w);
    run<1, 16>(X, out, w);
    run<1, 32>(X, out, w);
    run<2, 8>(X, out, w);
    run<2, 16>(X, out, w);
    run<2, 32>(X, out, w);
  }
  return 0;
}
