// libbbx_hostrng.so -- host-side, reference-stream samplers for the Gibbs
// steps adjacent to the CG draw (exact-seed parity mode of the driver).
//
// The reference keeps the Polya-Gamma and tilted-stable draws on the host even
// on its GPU (CuPy) path and feeds them from NumPy PCG64 bit generators
// (random/random.py:17-22, random/normal/normal.pyx, random/uniform/uniform.pyx,
// linked against libnpyrandom, setup.py:12-13,34-46).  This library does the
// same: it receives the address of a NumPy `bitgen_t`
// (`PCG64(seed).ctypes.bit_generator`) and consumes it in exactly the
// reference's order through the shared sampler templates of samplers.hpp.
// Plain C++ (g++), no HIP: it loads on a CPU-only box.
#include <stdint.h>

#include "../../include/bbx.h"
#include "samplers.hpp"

extern "C" {
// numpy/random/bitgen.h
typedef struct bitgen {
  void* state;
  uint64_t (*next_uint64)(void* st);
  uint32_t (*next_uint32)(void* st);
  double (*next_double)(void* st);
  uint64_t (*next_raw)(void* st);
} bitgen_t;
// numpy/random/distributions.h (libnpyrandom.a): 256-strip ziggurat
double random_standard_normal(bitgen_t* bitgen_state);
}

namespace {
struct NumpyStream {
  bitgen_t* bg;
  inline double uniform() { return bg->next_double(bg->state); }
  inline double normal() { return random_standard_normal(bg); }
};
}  // namespace

extern "C" {

int bbx_host_polya_gamma(void* bitgen, int64_t n, const int32_t* shape,
                         const double* tilt, double* out) {
  if (!bitgen || !shape || !tilt || !out || n < 0) return BBX_ERR_INVALID;
  NumpyStream g{static_cast<bitgen_t*>(bitgen)};
  for (int64_t i = 0; i < n; ++i)
    out[i] = bbx::PolyaGamma::draw(g, shape[i], tilt[i]);
  return BBX_OK;
}

int bbx_host_tilted_stable(void* bitgen, int64_t n, const double* char_exp,
                           const double* tilt, double* out) {
  if (!bitgen || !char_exp || !tilt || !out || n < 0) return BBX_ERR_INVALID;
  NumpyStream g{static_cast<bitgen_t*>(bitgen)};
  for (int64_t i = 0; i < n; ++i) {
    if (!(char_exp[i] < 1.) || !(tilt[i] > 0.)) return BBX_ERR_INVALID;
    out[i] = bbx::TiltedStable::draw(g, char_exp[i], tilt[i]);
  }
  return BBX_OK;
}

int bbx_host_pg_right_mass(int64_t n, const double* z, double* log_form,
                           double* direct) {
  if (!z || !log_form || !direct || n < 0) return BBX_ERR_INVALID;
  for (int64_t i = 0; i < n; ++i) {
    const double rate = 0.5 * z[i] * z[i] + 0.125 * bbx::kPi * bbx::kPi;
    log_form[i] = bbx::PolyaGamma::right_mass(z[i], rate);
    direct[i] = bbx::PolyaGamma::right_mass_direct(z[i], rate);
  }
  return BBX_OK;
}

namespace {
struct OneUniform {   // hands the series test the uniform it is asked about
  double u;
  inline double uniform() { return u; }
  inline double normal() { return 0.; }
};
}  // namespace

int bbx_host_pg_series_accept(int64_t n, const double* x, const double* u,
                              int32_t* sequential, int32_t* direct) {
  if (!x || !u || !sequential || !direct || n < 0) return BBX_ERR_INVALID;
  for (int64_t i = 0; i < n; ++i) {
    if (!(x[i] > 0.) || !(u[i] > 0. && u[i] < 1.)) return BBX_ERR_INVALID;
    OneUniform a{u[i]}, b{u[i]};
    sequential[i] = bbx::PolyaGamma::series_accept(a, x[i]) ? 1 : 0;
    direct[i] = bbx::PolyaGamma::series_accept_direct(b, x[i]) ? 1 : 0;
  }
  return BBX_OK;
}

}  // extern "C"
