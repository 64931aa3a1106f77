// LDS throughput of the two ways to apply a stored entry of a tile:
//   gather  : acc += xs[id]          (ds_read_b64, what tiled_spmv_kernel does)
//   scatter : g[id] += w             (ds_add_f64 without return: what a Tdot
//                                     over the X layout would need)
// Random 8-byte slots inside a 100 KB region, 16 waves per CU, one workgroup
// per CU; ids come from registers (no memory traffic), so the loop is LDS-bound.
// Reports wave-instructions per microsecond per CU and the id-stream rate this
// would sustain chip-wide at 2 bytes per entry.
// Build: hipcc -O3 --offload-arch=gfx950 -o lds_atomic lds_atomic.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int W = 12544;

template <int MODE>
__global__ __launch_bounds__(1024) void k(int iters, unsigned seed, double* out) {
  extern __shared__ double lds[];
  for (int j = threadIdx.x; j < W; j += 1024) lds[j] = 1.0;
  __syncthreads();
  unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  double acc = 0.;
  const double w = 1e-3 * (threadIdx.x & 7);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s = s * 1664525u + 1013904223u;
      const unsigned id = (s >> 8) % (unsigned)W;
      if (MODE == 0) {
        acc += lds[id];
      } else if (MODE == 1) {
        __hip_atomic_fetch_add(&lds[id], w, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        lds[id] += w;  // racy read-modify-write (ds_read + ds_write), bound only
      }
    }
  }
  __syncthreads();
  if (MODE != 0) acc = lds[threadIdx.x];
  if (acc == 123.456) out[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
static int run(const char* name, double* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>),
                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int iters = 2000;
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), W * 8 + 64, 0, iters, 12345u, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double wave_instr = 16.0 * iters * 8;  // per CU
  const double entries = 256.0 * 1024 * iters * 8;
  printf("%-22s %.3f ms  %.1f wave-instr/us/CU  = %.0f G entries/s chip-wide = %.1f TB/s of 2-byte ids\n",
         name, ms, wave_instr / (ms * 1e3), entries / ms / 1e6, 2 * entries / ms / 1e9);
  return 0;
}

int main() {
  double* out;
  CK(hipMalloc(&out, 64));
  run<0>("gather  ds_read_b64", out);
  run<1>("scatter ds_add_f64", out);
  run<2>("racy read+write", out);
  return 0;
}
