// exchange_rccl.cpp -- the collective transport of the one-band-per-process row-band solve carried by RCCL directly, for host
// programs without Python (liboctane_xchg_rccl.so, `make -C octane_amd/csrc -f Makefile.xchg`; NOT part of liboctane_vof.so, which has
// no collective library on its link line).  north_star: "RCCL over xGMI only for halo exchange / result gather".
//
// It implements the two callbacks of include/octane_vof.h (octane_vof_exchange) on one RCCL communicator and one stream:
//   all_gather  a grouped ncclSend of this rank's block to every other rank + an ncclRecv from every other rank into recv[c]
//               (the blocks land in a mirror whose layout ncclAllGather's contiguous output does not have);
//   sendrecv    the batch of edge-row / flow-band transfers as ONE ncclGroupStart ... ncclGroupEnd (RCCL matches the k-th send of a
//               rank to a peer with the peer's k-th receive from it, which is the contract the library states);
// both followed by a stream synchronisation: the library's calls are host-synchronous.
// The host program creates the id on rank 0 (octane_rccl_unique_id), distributes its 128 bytes -- the same all-gather that carries the
// IPC handles -- and every rank calls octane_rccl_exchange_create.  The Python binding of the same callbacks on torch.distributed
// (backend nccl = this library underneath) is octane_amd/exchange.py, which is what the tests run (gloo, host-staged: RCCL refuses
// two ranks on one GPU, profiles/r2_rccl_same_device.txt, and the test pool has one GPU per box).  THIS file has therefore only ever
// run with one rank (tests/test_gpu_tiled_mp.py::test_rccl_exchange_library_single_rank); on first contact with real peers the
// library's self-check (octane_vof_mp_selfcheck) compares what it moves with the plain plan before any frame is solved with it.
// No reference counterpart (the reference is single-GPU, ref src/oct_variational_optical_flow.cu:1251-1265).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>

#include "../../include/octane_xchg_rccl.h"

namespace {
struct RcclExchange {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, world = 1, device = 0;
};

int fail(const char *what, ncclResult_t r)
{
    fprintf(stderr, "octane exchange_rccl: %s: %s\n", what, ncclGetErrorString(r));
    return 1;
}

int cb_all_gather(void *user, const void *send, void *const *recv, size_t bytes)
{
    RcclExchange *x = static_cast<RcclExchange *>(user);
    if (hipSetDevice(x->device) != hipSuccess) return 1;
    ncclResult_t r = ncclGroupStart();
    for (int c = 0; c < x->world && r == ncclSuccess; c++) {
        if (c == x->rank) continue;
        r = ncclSend(send, bytes, ncclChar, c, x->comm, x->stream);
        if (r == ncclSuccess) r = ncclRecv(recv[c], bytes, ncclChar, c, x->comm, x->stream);
    }
    const ncclResult_t e = ncclGroupEnd();
    if (r != ncclSuccess || e != ncclSuccess) return fail("all_gather", r != ncclSuccess ? r : e);
    return hipStreamSynchronize(x->stream) == hipSuccess ? 0 : 1;
}

int cb_sendrecv(void *user, int n, const octane_vof_xfer *ops)
{
    RcclExchange *x = static_cast<RcclExchange *>(user);
    if (hipSetDevice(x->device) != hipSuccess) return 1;
    ncclResult_t r = ncclGroupStart();
    for (int i = 0; i < n && r == ncclSuccess; i++)
        r = ops[i].send ? ncclSend(ops[i].buf, ops[i].bytes, ncclChar, ops[i].peer, x->comm, x->stream)
                        : ncclRecv(ops[i].buf, ops[i].bytes, ncclChar, ops[i].peer, x->comm, x->stream);
    const ncclResult_t e = ncclGroupEnd();
    if (r != ncclSuccess || e != ncclSuccess) return fail("sendrecv", r != ncclSuccess ? r : e);
    return hipStreamSynchronize(x->stream) == hipSuccess ? 0 : 1;
}
}  // namespace

extern "C" {

static_assert(OCTANE_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "OCTANE_RCCL_ID_BYTES must be RCCL's unique-id size");

// rank 0: 128 bytes for every rank's octane_rccl_exchange_create
int octane_rccl_unique_id(void *out128)
{
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
    std::memcpy(out128, &id, sizeof id);
    return 0;
}

// Collective over the ranks.  Fills *ex for octane_vof_mp_set_exchange; keep it until octane_rccl_exchange_destroy(ex).
int octane_rccl_exchange_create(octane_vof_exchange *ex, const void *id128, int rank, int world, int device)
{
    if (!ex || !id128 || rank < 0 || rank >= world) return 1;
    RcclExchange *x = new RcclExchange();
    x->rank = rank; x->world = world; x->device = device;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking) != hipSuccess) { delete x; return 1; }
    const ncclResult_t r = ncclCommInitRank(&x->comm, world, id, rank);
    if (r != ncclSuccess) { (void)hipStreamDestroy(x->stream); delete x; return fail("ncclCommInitRank", r); }
    std::memset(ex, 0, sizeof *ex);
    ex->user = x; ex->all_gather = cb_all_gather; ex->sendrecv = cb_sendrecv;
    std::snprintf(ex->name, sizeof ex->name, "RCCL %d.%d (native, device buffers)", NCCL_MAJOR, NCCL_MINOR);
    return 0;
}

void octane_rccl_exchange_destroy(octane_vof_exchange *ex)
{
    if (!ex || !ex->user) return;
    RcclExchange *x = static_cast<RcclExchange *>(ex->user);
    (void)hipSetDevice(x->device);
    if (x->comm) (void)ncclCommDestroy(x->comm);
    if (x->stream) (void)hipStreamDestroy(x->stream);
    delete x;
    ex->user = nullptr;
}

}  // extern "C"
