/*
 * octane_xchg_rccl.h -- the collective transport of the one-band-per-process row-band solve on RCCL (liboctane_xchg_rccl.so,
 * octane_amd/csrc/exchange_rccl.cpp; optional, built by `make -C octane_amd/csrc -f Makefile.xchg` where RCCL is installed).
 * It fills an octane_vof_exchange (include/octane_vof.h) for octane_vof_mp_set_exchange: an all-gather of the bands' partial sums and the
 * batches of edge-row / flow-band transfers as grouped ncclSend / ncclRecv on device buffers.  north_star: "RCCL over xGMI only for halo
 * exchange / result gather".  No reference counterpart (the reference is single-GPU, ref src/oct_variational_optical_flow.cu:1251-1265).
 */
#ifndef OCTANE_XCHG_RCCL_H
#define OCTANE_XCHG_RCCL_H

#include "octane_vof.h"

#ifdef __cplusplus
extern "C" {
#endif

#define OCTANE_RCCL_ID_BYTES 128
/* Rank 0: OCTANE_RCCL_ID_BYTES bytes that the host program hands to every rank (the all-gather that carries the IPC handles will do). */
int octane_rccl_unique_id(void *out128);
/* Every rank (collective): a communicator of `world` ranks on `device`; *ex is ready for octane_vof_mp_set_exchange.  Returns 0. */
int octane_rccl_exchange_create(octane_vof_exchange *ex, const void *id128, int rank, int world, int device);
/* After the plan that used it is destroyed. */
void octane_rccl_exchange_destroy(octane_vof_exchange *ex);

#ifdef __cplusplus
}
#endif
#endif
