// pcg_persist_diag.hip -- the stamped diagnostic build of the persistent mid-level solve: the same source as pcg_persist.hip with
// MID_DIAG defined (shader-clock stamps at the seams of an iteration) under names of its own, so that the production kernel's
// text is not touched by the instrumentation.  Selected by octane_vof_tune(plan, "persist_diag", 1); read by octane_vof_mid_stamps.
#define MID_DIAG 1
#define k_pcg_solve_mid k_pcg_solve_mid_diag
#define launch_pcg_solve_mid launch_pcg_solve_mid_diag
#define pcg_mid_configure pcg_mid_configure_diag
#include "pcg_persist.hip"
