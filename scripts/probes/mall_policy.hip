// Can one stream be kept resident in the 256 MB Infinity Cache (MALL) while a
// second, larger-than-the-rest stream passes through between its uses?
//
// The CG loop alternates X (215 MB of ids) and X^T (222 MB): together they
// exceed the MALL, so with plain allocation each evicts the other and every
// launch streams from HBM.  This probe reads buffer A (nt loads, like the
// kernel) and buffer B alternately and reports A's read rate for every
// combination of B's allocation kind and B's load cache-policy bits.  If some
// combination keeps A at the resident rate (~10 TB/s instead of ~6.5), X can
// live in the MALL.
// Build: hipcc -O3 --offload-arch=gfx950 -o mall_policy mall_policy.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));

#define LOADER(NAME, MOD)                                                      \
  __device__ __forceinline__ v4u NAME(const v4u* p) {                          \
    v4u r;                                                                     \
    asm volatile("global_load_dwordx4 %0, %1, off" MOD "\n s_waitcnt vmcnt(0)" \
                 : "=v"(r)                                                     \
                 : "v"(p)                                                      \
                 : "memory");                                                  \
    return r;                                                                  \
  }
// (the wait inside would serialise; the kernel below uses the unwaited form)
#undef LOADER

template <int MOD>
__device__ __forceinline__ void ld4(v4u& a, v4u& b, v4u& c, v4u& d,
                                    const v4u* p, int64_t st) {
  const v4u* p1 = p + st;
  const v4u* p2 = p + 2 * st;
  const v4u* p3 = p + 3 * st;
#define L4(M)                                                                  \
  asm volatile("global_load_dwordx4 %0, %4, off" M "\n"                        \
               "global_load_dwordx4 %1, %5, off" M "\n"                        \
               "global_load_dwordx4 %2, %6, off" M "\n"                        \
               "global_load_dwordx4 %3, %7, off" M "\n"                        \
               "s_waitcnt vmcnt(0)"                                            \
               : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)                        \
               : "v"(p), "v"(p1), "v"(p2), "v"(p3)                             \
               : "memory")
  if (MOD == 0) L4("");
  else if (MOD == 1) L4(" nt");
  else if (MOD == 2) L4(" sc0");
  else if (MOD == 3) L4(" sc1");
  else if (MOD == 4) L4(" sc0 sc1");
  else if (MOD == 5) L4(" sc0 sc1 nt");
  else if (MOD == 6) L4(" sc1 nt");
  else L4(" sc0 nt");
#undef L4
}

template <int MOD>
__global__ __launch_bounds__(1024) void rd(const v4u* __restrict__ src,
                                           int64_t n16, unsigned* sink) {
  v4u acc = {0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    v4u a, b, c, d;
    ld4<MOD>(a, b, c, d, src + i, stride);
    acc ^= a ^ b ^ c ^ d;
  }
  const unsigned f = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (f == 0x9E3779B9u) sink[0] = f;
}

#define CK(x)                                                     \
  do {                                                            \
    hipError_t e = (x);                                           \
    if (e != hipSuccess) {                                        \
      printf("%s: %s\n", #x, hipGetErrorString(e));               \
      return 1;                                                   \
    }                                                             \
  } while (0)

static void launch(int mod, const void* p, int64_t n16, void* sink, int wgs) {
  switch (mod) {
#define C(M) case M: hipLaunchKernelGGL(rd<M>, dim3(wgs), dim3(1024), 0, 0, (const v4u*)p, n16, (unsigned*)sink); break;
    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7)
#undef C
  }
}

int main(int argc, char** argv) {
  const int64_t bytes_a = argc > 1 ? atoll(argv[1]) : 215000000;
  const int64_t bytes_b = argc > 2 ? atoll(argv[2]) : 222000000;
  const int wgs = argc > 3 ? atoi(argv[3]) : 2048;
  const int64_t na = bytes_a / 16, nb = bytes_b / 16;
  void *a, *sink, *b[3] = {nullptr, nullptr, nullptr};
  CK(hipMalloc(&a, na * 16));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 1, na * 16));
  const char* kind[3] = {"hipMalloc", "uncached", "finegrained"};
  CK(hipMalloc(&b[0], nb * 16));
  if (hipExtMallocWithFlags(&b[1], nb * 16, hipDeviceMallocUncached) != hipSuccess) b[1] = nullptr;
  if (hipExtMallocWithFlags(&b[2], nb * 16, hipDeviceMallocFinegrained) != hipSuccess) b[2] = nullptr;
  for (int k = 0; k < 3; ++k)
    if (b[k]) CK(hipMemset(b[k], 2, nb * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* mods[8] = {"(none)", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc1 nt", "sc0 nt"};
  const int reps = 30;
  // baseline: A alone, re-read (resident) with nt
  {
    float tot = 0.f;
    for (int it = -3; it < reps; ++it) {
      CK(hipEventRecord(e0, 0));
      launch(1, a, na, sink, wgs);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 0) tot += ms;
    }
    printf("A alone (nt), %d wgs: %.1f us  %.0f GB/s\n", wgs, 1e3 * tot / reps, na * 16.0 * reps / tot / 1e6);
  }
  for (int k = 0; k < 3; ++k) {
    if (!b[k]) {
      printf("B %s: allocation not available\n", kind[k]);
      continue;
    }
    for (int mod = 0; mod < 8; ++mod) {
      float ta = 0.f, tb = 0.f;
      for (int it = -3; it < reps; ++it) {
        float ms;
        CK(hipEventRecord(e0, 0));
        launch(1, a, na, sink, wgs);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 0) ta += ms;
        CK(hipEventRecord(e0, 0));
        launch(mod, b[k], nb, sink, wgs);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 0) tb += ms;
      }
      printf("B %-11s loads %-10s: A %.1f us %.0f GB/s | B %.1f us %.0f GB/s\n",
             kind[k], mods[mod], 1e3 * ta / reps, na * 16.0 * reps / ta / 1e6,
             1e3 * tb / reps, nb * 16.0 * reps / tb / 1e6);
    }
  }
  return 0;
}
