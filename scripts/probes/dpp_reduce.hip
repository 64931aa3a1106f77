#include <hip/hip_runtime.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ inline double row_ror_add(double v, int) { return v; }
template <int CTRL>
__device__ inline double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__global__ void k(const double* in, double* out) {
  double a = in[threadIdx.x], b = in[64 + threadIdx.x];
  // transpose step
  unsigned alo = __double2loint(a), ahi = __double2hiint(a);
  unsigned blo = __double2loint(b), bhi = __double2hiint(b);
  v2u s0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
  v2u s1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
  double x = __hiloint2double(s1.x, s0.x), y = __hiloint2double(s1.y, s0.y);
  double m = x + y;
  m += dpp_mov<0x128>(m);
  m += dpp_mov<0x124>(m);
  m += dpp_mov<0x122>(m);
  m += dpp_mov<0x121>(m);
  out[threadIdx.x] = m;
  out[64 + threadIdx.x] = x;
  out[128 + threadIdx.x] = y;
}
int main() {
  double h[128], o[192]; for (int i = 0; i < 128; ++i) h[i] = i < 64 ? i : 1000 + (i - 64);
  double *d, *e; hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
  hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i += 8) printf("lane %2d: m %.0f x %.0f y %.0f\n", i, o[i], o[64 + i], o[128 + i]);
  return 0;
}
