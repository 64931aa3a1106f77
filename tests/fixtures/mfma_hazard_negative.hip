// NEGATIVE fixture of tests/test_kernel_resources.py: the two mistakes the asm
// MFMA kernels of csrc/dense_batch.hip must never contain.  Never run.
#include <hip/hip_runtime.h>
typedef double d4 __attribute__((ext_vector_type(4)));
// the documented first version: the conversion directly in front of its MFMA
__global__ void cvt_in_front_of_its_mfma(const float* x, const double* b, d4* out) {
  d4 acc = {0., 0., 0., 0.};
  double t;
  asm volatile("v_cvt_f64_f32 %1, %2\n\tv_mfma_f64_16x16x4_f64 %0, %1, %3, %0"
               : "+v"(acc), "=&v"(t) : "v"(x[threadIdx.x]), "v"(b[threadIdx.x]));
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
  out[threadIdx.x] = acc;
}
// an accumulator read 4 wait states after the MFMA that writes it
__global__ void result_read_too_early(const double* x, const double* b, d4* out) {
  d4 acc = {0., 0., 0., 0.};
  asm volatile("s_nop 3\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\ts_nop 3"
               : "+v"(acc) : "v"(x[threadIdx.x]), "v"(b[threadIdx.x]));
  out[threadIdx.x] = acc;
}
// an asm-issued load whose destination is copied before any wait (what a
// compiler that re-allocates around an in-flight ring register would emit)
__global__ void ring_register_touched_in_flight(const unsigned* p, unsigned* out) {
  unsigned x, off = 4u * threadIdx.x;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(x) : "v"(off), "s"(p) : "memory");
  out[threadIdx.x] = x + 1u;
}
// the round-4 bug: a value kept live across the slot's re-issue makes the
// register allocator copy the tied wait operand IN FRONT of the wait -- a
// v_mov from a register whose load is still in flight (here spelled out)
__global__ void ring_register_copied_before_its_wait(const unsigned* p, unsigned* out) {
  unsigned x, y, off = 4u * threadIdx.x;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(x) : "v"(off), "s"(p) : "memory");
  asm volatile("v_mov_b32 %0, %1\n\ts_waitcnt vmcnt(0)" : "=v"(y) : "v"(x));
  out[threadIdx.x] = y;
}
