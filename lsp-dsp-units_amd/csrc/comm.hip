// The one exchange step of the hot path: the cross-channel per-bin sum of the spectral path (BASELINE config 5, the
// MultiSpectralProcessor-style callback, util/MultiSpectralProcessor.h:41) when the channels are sharded over the
// GPUs of a node.  Every rank reduces its own channels on the device (mi_analyzer_bank_reduce_bins) and the partial
// sums are all-reduced with RCCL over xGMI, from this library's C++ host side -- one ncclAllReduce of
// frames x (2^(rank-1)+1) floats on the caller's stream, no torch involved.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first communicator call): single-GPU users of the library
// do not need it, and a process that already carries a copy (PyTorch's torch.distributed) shares that copy.
#include "mi_common.h"

#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace
{
    struct rccl_api
    {
        void *handle = nullptr;
        decltype(&ncclGetUniqueId)   get_unique_id = nullptr;
        decltype(&ncclCommInitRank)  comm_init_rank = nullptr;
        decltype(&ncclCommDestroy)   comm_destroy = nullptr;
        decltype(&ncclAllReduce)     all_reduce = nullptr;
        decltype(&ncclGetErrorString) error_string = nullptr;
        decltype(&ncclCommCount)     comm_count = nullptr;
        decltype(&ncclCommUserRank)  comm_user_rank = nullptr;
    };

    rccl_api *rccl()
    {
        static rccl_api api;
        static std::once_flag once;
        std::call_once(once, []
        {
            for (const char *name : { "librccl.so.1", "librccl.so" })
                if ((api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL)) != nullptr)
                    break;
            if (api.handle == nullptr)
                return;
            api.get_unique_id  = reinterpret_cast<decltype(api.get_unique_id)>(dlsym(api.handle, "ncclGetUniqueId"));
            api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(dlsym(api.handle, "ncclCommInitRank"));
            api.comm_destroy   = reinterpret_cast<decltype(api.comm_destroy)>(dlsym(api.handle, "ncclCommDestroy"));
            api.all_reduce     = reinterpret_cast<decltype(api.all_reduce)>(dlsym(api.handle, "ncclAllReduce"));
            api.error_string   = reinterpret_cast<decltype(api.error_string)>(dlsym(api.handle, "ncclGetErrorString"));
            api.comm_count     = reinterpret_cast<decltype(api.comm_count)>(dlsym(api.handle, "ncclCommCount"));
            api.comm_user_rank = reinterpret_cast<decltype(api.comm_user_rank)>(dlsym(api.handle, "ncclCommUserRank"));
        });
        const bool ok = api.handle && api.get_unique_id && api.comm_init_rank && api.comm_destroy && api.all_reduce &&
                        api.error_string && api.comm_count && api.comm_user_rank;
        return ok ? &api : nullptr;
    }

    int nccl_fail(const rccl_api *a, const char *what, ncclResult_t r)
    {
        return mi::fail(MI_EHIP, "%s failed: %s", what, a->error_string(r));
    }
} // namespace

struct mi_dspu_comm
{
    ncclComm_t  comm = nullptr;
    bool        owned = false;
    int         nranks = 1, rank = 0;
    // the side stream of mi_analyzer_bank_allreduce_bins_begin: a collective there runs beside whatever the caller's stream
    // goes on with; `ready` = the caller's stream has produced the slot's partial sums, `done` = the slot's collective is through
    hipStream_t side = nullptr;
    hipEvent_t  ready[MI_DSPU_COMM_SLOTS] = {}, done[MI_DSPU_COMM_SLOTS] = {};
    bool        pending[MI_DSPU_COMM_SLOTS] = {};
};

namespace
{
    int comm_side(mi_dspu_comm *c)                          // made on first use (a communicator that never overlaps has none)
    {
        if (c->side != nullptr)
            return MI_OK;
        // The LEAST urgent priority: the analysis launch of the next batch holds every CU (one persistent workgroup each, all of a
        // CU's LDS and registers) for its ~80 us; a collective kernel that got a CU first would hold up that CU's workgroup -- a
        // quarter of the launch's work -- by its own duration, the serial cost over again.  Least urgent, it takes the CUs the
        // analysis leaves (the reduction behind it does not fill them) and ends inside the two batches the caller's slots give it.
        int least = 0, most = 0;
        MI_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &most));
        // (all of it or none: a stream without its events would be taken for a complete set by the next call)
        hipStream_t side = nullptr;
        hipEvent_t ev[2 * MI_DSPU_COMM_SLOTS] = {};
        hipError_t e = hipStreamCreateWithPriority(&side, hipStreamNonBlocking, least);
        for (int k = 0; k < 2 * MI_DSPU_COMM_SLOTS && e == hipSuccess; ++k)
            e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
        if (e != hipSuccess)
        {
            for (hipEvent_t v : ev)
                if (v != nullptr)
                    (void)hipEventDestroy(v);
            if (side != nullptr)
                (void)hipStreamDestroy(side);
            MI_HIP_CHECK(e);
        }
        for (int k = 0; k < MI_DSPU_COMM_SLOTS; ++k)
        {
            c->ready[k] = ev[2 * k];
            c->done[k] = ev[2 * k + 1];
        }
        c->side = side;
        return MI_OK;
    }
}

extern "C" {

int mi_dspu_comm_unique_id(void *id128)
{
    MI_REQUIRE(id128 != nullptr, MI_EINVAL, "mi_dspu_comm_unique_id: NULL buffer");
    static_assert(sizeof(ncclUniqueId) == MI_DSPU_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    rccl_api *a = rccl();
    MI_REQUIRE(a != nullptr, MI_ENODEV, "RCCL (librccl.so.1) could not be loaded: %s", dlerror());
    ncclUniqueId id;
    const ncclResult_t r = a->get_unique_id(&id);
    if (r != ncclSuccess)
        return nccl_fail(a, "ncclGetUniqueId", r);
    memcpy(id128, &id, sizeof(id));
    return MI_OK;
}

int mi_dspu_comm_create(mi_dspu_comm_t **comm, const void *id128, int nranks, int rank)
{
    MI_REQUIRE(comm != nullptr, MI_EINVAL, "mi_dspu_comm_create: NULL result pointer");
    *comm = nullptr;
    MI_REQUIRE(id128 != nullptr && nranks >= 1 && rank >= 0 && rank < nranks, MI_EINVAL,
               "mi_dspu_comm_create: bad id / rank %d of %d", rank, nranks);
    MI_REQUIRE(mi_dspu_device_count() > 0, MI_ENODEV, "no HIP device available (there is no CPU fallback)");
    rccl_api *a = rccl();
    MI_REQUIRE(a != nullptr, MI_ENODEV, "RCCL (librccl.so.1) could not be loaded: %s", dlerror());
    mi_dspu_comm *c = new (std::nothrow) mi_dspu_comm();
    MI_REQUIRE(c != nullptr, MI_ENOMEM, "mi_dspu_comm_create: out of host memory");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    const ncclResult_t r = a->comm_init_rank(&c->comm, nranks, id, rank);     // one process per GPU: the current device
    if (r != ncclSuccess)
    {
        delete c;
        return nccl_fail(a, "ncclCommInitRank", r);
    }
    c->owned = true;
    c->nranks = nranks;
    c->rank = rank;
    *comm = c;
    return MI_OK;
}

int mi_dspu_comm_adopt(mi_dspu_comm_t **comm, void *nccl_comm)
{
    MI_REQUIRE(comm != nullptr, MI_EINVAL, "mi_dspu_comm_adopt: NULL result pointer");
    *comm = nullptr;
    MI_REQUIRE(nccl_comm != nullptr, MI_EINVAL, "mi_dspu_comm_adopt: NULL ncclComm_t");
    rccl_api *a = rccl();
    MI_REQUIRE(a != nullptr, MI_ENODEV, "RCCL (librccl.so.1) could not be loaded: %s", dlerror());
    mi_dspu_comm *c = new (std::nothrow) mi_dspu_comm();
    MI_REQUIRE(c != nullptr, MI_ENOMEM, "mi_dspu_comm_adopt: out of host memory");
    c->comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    c->owned = false;
    ncclResult_t r = a->comm_count(c->comm, &c->nranks);
    if (r == ncclSuccess)
        r = a->comm_user_rank(c->comm, &c->rank);
    if (r != ncclSuccess)
    {
        delete c;
        return nccl_fail(a, "ncclCommCount", r);
    }
    *comm = c;
    return MI_OK;
}

int mi_dspu_comm_destroy(mi_dspu_comm_t *c)
{
    if (c == nullptr)
        return MI_OK;
    rccl_api *a = rccl();
    if (c->side != nullptr)
    {
        (void)hipStreamSynchronize(c->side);
        for (int k = 0; k < MI_DSPU_COMM_SLOTS; ++k)
        {
            if (c->ready[k] != nullptr) (void)hipEventDestroy(c->ready[k]);
            if (c->done[k] != nullptr) (void)hipEventDestroy(c->done[k]);
        }
        (void)hipStreamDestroy(c->side);
    }
    if (c->owned && c->comm != nullptr && a != nullptr)
        (void)a->comm_destroy(c->comm);
    delete c;
    return MI_OK;
}

int mi_dspu_comm_info(const mi_dspu_comm_t *c, int *nranks, int *rank)
{
    MI_REQUIRE(c != nullptr, MI_ESTATE, "mi_dspu_comm_info: NULL communicator");
    if (nranks) *nranks = c->nranks;
    if (rank)   *rank = c->rank;
    return MI_OK;
}

int mi_analyzer_bank_allreduce_bins(mi_analyzer_bank_t *bank, float *bins, size_t frames, mi_dspu_comm_t *c, void *stream)
{
    MI_REQUIRE(bank != nullptr, MI_ESTATE, "mi_analyzer_bank_allreduce_bins: NULL bank");
    MI_REQUIRE(c != nullptr, MI_ESTATE, "mi_analyzer_bank_allreduce_bins: NULL communicator");
    if (frames == 0)
        return MI_OK;
    MI_REQUIRE(bins != nullptr, MI_EINVAL, "mi_analyzer_bank_allreduce_bins: NULL buffer");
    uint32_t nb = 0;
    const int q = mi_analyzer_bank_info(bank, nullptr, &nb, nullptr, nullptr);
    if (q != MI_OK)
        return q;
    rccl_api *a = rccl();
    MI_REQUIRE(a != nullptr, MI_ENODEV, "RCCL (librccl.so.1) could not be loaded");
    // in place, float32 sum, on the caller's stream: ordered behind the reduce_bins launches that filled `bins`
    const ncclResult_t r = a->all_reduce(bins, bins, frames * size_t(nb), ncclFloat, ncclSum, c->comm, mi::as_stream(stream));
    if (r != ncclSuccess)
        return nccl_fail(a, "ncclAllReduce", r);
    return MI_OK;
}

// The same exchange step BESIDE the caller's stream (round 6): the batch's collective is a small message whose latency -- a few
// tens of microseconds over xGMI -- otherwise stands between two batches on the one stream.  begin(): the side stream waits for
// what `stream` has enqueued so far (the batch's reduction: an event), sums `partial` over the ranks into `total` (the same buffer:
// in place) and marks the slot done; the caller's stream goes straight on with the next batch, which writes ANOTHER buffer.
// wait(): `stream` waits (on the device: the host does not block) for the slot's collective -- in front of whatever consumes
// `total`, and in front of the reduction that overwrites the slot's buffers two batches later.  Slots are the caller's double
// (or deeper) buffering; a slot that is begun again while pending is waited for on the side stream's own order.
int mi_analyzer_bank_allreduce_bins_begin(mi_analyzer_bank_t *bank, const float *partial, float *total, size_t frames,
                                          mi_dspu_comm_t *c, int slot, void *stream)
{
    MI_REQUIRE(bank != nullptr, MI_ESTATE, "mi_analyzer_bank_allreduce_bins_begin: NULL bank");
    MI_REQUIRE(c != nullptr, MI_ESTATE, "mi_analyzer_bank_allreduce_bins_begin: NULL communicator");
    MI_REQUIRE(slot >= 0 && slot < MI_DSPU_COMM_SLOTS, MI_EINVAL, "mi_analyzer_bank_allreduce_bins_begin: slot %d outside 0..%d", slot, MI_DSPU_COMM_SLOTS - 1);
    if (frames == 0)
        return MI_OK;
    MI_REQUIRE(partial != nullptr && total != nullptr, MI_EINVAL, "mi_analyzer_bank_allreduce_bins_begin: NULL buffer");
    uint32_t nb = 0;
    const int q = mi_analyzer_bank_info(bank, nullptr, &nb, nullptr, nullptr);
    if (q != MI_OK)
        return q;
    rccl_api *a = rccl();
    MI_REQUIRE(a != nullptr, MI_ENODEV, "RCCL (librccl.so.1) could not be loaded");
    const int rs = comm_side(c);
    if (rs != MI_OK)
        return rs;
    hipStream_t st = mi::as_stream(stream);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    MI_REQUIRE(!(st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone), MI_ESTATE,
               "mi_analyzer_bank_allreduce_bins_begin: the stream is being captured (the collective's side stream is not part of a graph)");
    MI_HIP_CHECK(hipEventRecord(c->ready[slot], st));
    MI_HIP_CHECK(hipStreamWaitEvent(c->side, c->ready[slot], 0));
    const ncclResult_t r = a->all_reduce(partial, total, frames * size_t(nb), ncclFloat, ncclSum, c->comm, c->side);
    if (r != ncclSuccess)
        return nccl_fail(a, "ncclAllReduce", r);
    MI_HIP_CHECK(hipEventRecord(c->done[slot], c->side));
    c->pending[slot] = true;
    return MI_OK;
}

int mi_dspu_comm_wait(mi_dspu_comm_t *c, int slot, void *stream)
{
    MI_REQUIRE(c != nullptr, MI_ESTATE, "mi_dspu_comm_wait: NULL communicator");
    MI_REQUIRE(slot >= 0 && slot < MI_DSPU_COMM_SLOTS, MI_EINVAL, "mi_dspu_comm_wait: slot %d outside 0..%d", slot, MI_DSPU_COMM_SLOTS - 1);
    if (!c->pending[slot])                                  // nothing begun in this slot (or already waited for): nothing to wait for
        return MI_OK;
    MI_HIP_CHECK(hipStreamWaitEvent(mi::as_stream(stream), c->done[slot], 0));
    c->pending[slot] = false;
    return MI_OK;
}

} // extern "C"
