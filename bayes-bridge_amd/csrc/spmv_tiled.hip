// LDS-tiled sparse operator (placeholder until the tiled kernels land).
#include "common.hpp"
namespace bbx {
int build_tiled(bbx_design*) {
  return fail(BBX_ERR_STATE, "tiled format not built yet");
}
int launch_dot_tiled(bbx_design*, const double*, const double*, double*) {
  return fail(BBX_ERR_STATE, "tiled format not built yet");
}
int launch_tdot_tiled(bbx_design*, const double*, const double*,
                      const TdotEpilogue&, double*) {
  return fail(BBX_ERR_STATE, "tiled format not built yet");
}
}  // namespace bbx
