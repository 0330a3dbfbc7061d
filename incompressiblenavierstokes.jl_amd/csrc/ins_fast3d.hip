// Tiled 3-D fast paths (gfx950).  Placeholder dispatch until the LDS plane-ring kernel lands.
#include "ins_internal.h"

bool ins_fast3d_supported(const ins_grid* G) {
  (void)G;
  return false;
}

int ins_k_momentum_fast3d(const ins_grid* G, double visc, const double* u, double* F, hipStream_t s) {
  (void)G; (void)visc; (void)u; (void)F; (void)s;
  ins_set_error("fast3d momentum not built");
  return INS_ERR_UNSUPPORTED;
}
