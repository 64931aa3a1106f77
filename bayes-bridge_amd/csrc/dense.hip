// Dense design operator (placeholder until the GEMV kernels land).
#include "common.hpp"
namespace bbx {
int launch_dot_dense(bbx_design*, const double*, const double*, double*) {
  return fail(BBX_ERR_STATE, "dense operator not built yet");
}
int launch_tdot_dense(bbx_design*, const double*, const double*,
                      const TdotEpilogue&, double*) {
  return fail(BBX_ERR_STATE, "dense operator not built yet");
}
}  // namespace bbx
extern "C" {
int bbx_design_create_dense(int64_t, int64_t, const void*, int, int,
                            const double*, int, int, bbx_design** out) {
  if (out) *out = nullptr;
  return bbx::fail(BBX_ERR_STATE, "dense operator not built yet");
}
int bbx_design_create_dense_dev(int64_t, int64_t, const void*, int, int,
                                const double*, int, int, bbx_design** out) {
  if (out) *out = nullptr;
  return bbx::fail(BBX_ERR_STATE, "dense operator not built yet");
}
}
