// Device/stream/memory plumbing of the C-ABI (include/mi_dspu.h).
#include "mi_common.h"
#include <cstdlib>
#include <cstring>
#include <vector>
#include <mutex>
#include <map>

namespace mi
{
    char *error_buffer()
    {
        static thread_local char buf[512] = "";
        return buf;
    }

    int fail(int code, const char *fmt, ...)
    {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(error_buffer(), 512, fmt, ap);
        va_end(ap);
        return code;
    }
    static thread_local hipEvent_t tl_start = nullptr, tl_stop = nullptr;

    bool compat_bits()
    {
        const char *v = getenv("MI_DSPU_COMPAT_BITS");
        return v != nullptr && v[0] != '\0' && v[0] != '0';
    }

    bool test_path(const char *name)
    {
        const char *v = getenv("MI_DSPU_TEST_PATH");
        if (v == nullptr || name == nullptr)
            return false;
        const size_t n = strlen(name);
        for (const char *p = v; (p = strstr(p, name)) != nullptr; p += n)
            if ((p == v || p[-1] == ',') && (p[n] == '\0' || p[n] == ','))
                return true;
        return false;
    }

    static thread_local const char *tl_last_launch = "";
    void note_launch(const char *kernel) { tl_last_launch = kernel; }

    void take_profile_events(hipEvent_t *start, hipEvent_t *stop)
    {
        *start = tl_start;
        *stop  = tl_stop;
        tl_start = nullptr;
        tl_stop  = nullptr;
    }
} // namespace mi

namespace
{
    struct capture_note { const void *bank; const char *what; mi::position_fn fn; uint64_t sig; uint64_t epoch; };
    std::mutex g_capture_lock;
    std::map<hipStream_t, std::vector<capture_note>> g_captures;       // streams being captured through mi_dspu_graph_begin_capture
    // Banks whose device buffers have been re-made since they were created (the convolver's ring, grown by its first batch
    // of frames): a graph captured before that has the old addresses and sizes baked into its launches.  Every captured
    // bank's epoch is kept with the executable graph and compared again at every launch.
    std::map<const void *, uint64_t> g_epochs;
    uint64_t epoch_of(const void *bank)                                // (g_capture_lock held)
    {
        auto it = g_epochs.find(bank);
        return it == g_epochs.end() ? 0 : it->second;
    }
    struct graph_handle
    {
        hipGraphExec_t exec = nullptr;
        std::vector<capture_note> banks;
    };
}

namespace mi
{
    int capture_touch(hipStream_t st, const void *bank, const char *what, position_fn fn)
    {
        if (st == nullptr)
            return MI_OK;
        hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &status) != hipSuccess || status != hipStreamCaptureStatusActive)
            return MI_OK;
        std::lock_guard<std::mutex> lock(g_capture_lock);
        auto it = g_captures.find(st);
        if (it == g_captures.end())
            return fail(MI_ESTATE, "the %s bank keeps ring positions on the host: capture it through mi_dspu_graph_begin_capture / "
                        "end_capture, which check that the captured calls bring it back to its starting positions", what);
        for (const capture_note &n : it->second)
            if (n.bank == bank)
                return MI_OK;
        it->second.push_back(capture_note{ bank, what, fn, fn(bank), epoch_of(bank) });
        return MI_OK;
    }

    void bank_epoch_bump(const void *bank)
    {
        std::lock_guard<std::mutex> lock(g_capture_lock);
        ++g_epochs[bank];
    }

    void bank_epoch_forget(const void *bank)
    {
        std::lock_guard<std::mutex> lock(g_capture_lock);
        g_epochs.erase(bank);
    }
}

extern "C" {

int mi_dspu_profile_next_launch(void *start_event, void *stop_event)
{
    mi::tl_start = reinterpret_cast<hipEvent_t>(start_event);
    mi::tl_stop  = reinterpret_cast<hipEvent_t>(stop_event);
    return MI_OK;
}


const char *mi_dspu_last_launch(void) { return mi::tl_last_launch; }

const char *mi_dspu_source_sha(const char *file)
{
    // build/src_sha.h: { "biquad.hip", "0123..." }, ... written by the Makefile from the sources of this build
    static const struct { const char *file, *sha; } table[] = {
#if __has_include("src_sha.h")
#include "src_sha.h"
#endif
        { nullptr, nullptr } };
    for (int i = 0; file != nullptr && table[i].file != nullptr; ++i)
        if (strcmp(table[i].file, file) == 0)
            return table[i].sha;
    return nullptr;
}

int mi_dspu_abi_version(void) { return MI_DSPU_ABI_VERSION; }

const char *mi_dspu_last_error(void) { return mi::error_buffer(); }

int mi_dspu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
    {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int mi_dspu_set_device(int device)
{
    MI_HIP_CHECK(hipSetDevice(device));
    return MI_OK;
}

int mi_dspu_malloc(void **dev_ptr, size_t bytes)
{
    MI_REQUIRE(dev_ptr != nullptr, MI_EINVAL, "mi_dspu_malloc: NULL result pointer");
    *dev_ptr = nullptr;
    if (bytes == 0)
        return MI_OK;
    hipError_t e = hipMalloc(dev_ptr, bytes);
    if (e == hipErrorOutOfMemory)
        return mi::fail(MI_ENOMEM, "hipMalloc(%zu) out of memory", bytes);
    MI_HIP_CHECK(e);
    return MI_OK;
}

int mi_dspu_free(void *dev_ptr)
{
    if (dev_ptr != nullptr)
        MI_HIP_CHECK(hipFree(dev_ptr));
    return MI_OK;
}

int mi_dspu_memset(void *dev_ptr, int value, size_t bytes, void *stream)
{
    if (bytes > 0)
        MI_HIP_CHECK(hipMemsetAsync(dev_ptr, value, bytes, mi::as_stream(stream)));
    return MI_OK;
}

int mi_dspu_copy_h2d(void *dev_dst, const void *host_src, size_t bytes, void *stream)
{
    if (bytes > 0)
        MI_HIP_CHECK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, mi::as_stream(stream)));
    return MI_OK;
}

int mi_dspu_copy_d2h(void *host_dst, const void *dev_src, size_t bytes, void *stream)
{
    if (bytes > 0)
        MI_HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, mi::as_stream(stream)));
    return MI_OK;
}

int mi_dspu_copy_d2d(void *dev_dst, const void *dev_src, size_t bytes, void *stream)
{
    if (bytes > 0)
        MI_HIP_CHECK(hipMemcpyAsync(dev_dst, dev_src, bytes, hipMemcpyDeviceToDevice, mi::as_stream(stream)));
    return MI_OK;
}

int mi_dspu_stream_create(void **stream)
{
    MI_REQUIRE(stream != nullptr, MI_EINVAL, "mi_dspu_stream_create: NULL result pointer");
    hipStream_t s;
    MI_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return MI_OK;
}

int mi_dspu_stream_destroy(void *stream)
{
    if (stream != nullptr)
        MI_HIP_CHECK(hipStreamDestroy(mi::as_stream(stream)));
    return MI_OK;
}

int mi_dspu_stream_synchronize(void *stream)
{
    MI_HIP_CHECK(hipStreamSynchronize(mi::as_stream(stream)));
    return MI_OK;
}

int mi_dspu_event_create(void **event)
{
    MI_REQUIRE(event != nullptr, MI_EINVAL, "mi_dspu_event_create: NULL result pointer");
    hipEvent_t e;
    MI_HIP_CHECK(hipEventCreate(&e));
    *event = e;
    return MI_OK;
}

int mi_dspu_event_destroy(void *event)
{
    if (event != nullptr)
        MI_HIP_CHECK(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return MI_OK;
}

int mi_dspu_event_record(void *event, void *stream)
{
    MI_HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(event), mi::as_stream(stream)));
    return MI_OK;
}

int mi_dspu_event_synchronize(void *event)
{
    MI_HIP_CHECK(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return MI_OK;
}

int mi_dspu_event_elapsed_ms(float *ms, void *start, void *stop)
{
    MI_REQUIRE(ms != nullptr, MI_EINVAL, "mi_dspu_event_elapsed_ms: NULL result pointer");
    MI_HIP_CHECK(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return MI_OK;
}

int mi_dspu_graph_begin_capture(void *stream)
{
    MI_REQUIRE(stream != nullptr, MI_EINVAL, "mi_dspu_graph_begin_capture: the NULL stream cannot be captured");
    MI_HIP_CHECK(hipStreamBeginCapture(mi::as_stream(stream), hipStreamCaptureModeThreadLocal));
    std::lock_guard<std::mutex> lock(g_capture_lock);
    g_captures[mi::as_stream(stream)].clear();
    return MI_OK;
}

int mi_dspu_graph_end_capture(void *stream, void **graph_exec)
{
    MI_REQUIRE(graph_exec != nullptr, MI_EINVAL, "mi_dspu_graph_end_capture: NULL result pointer");
    *graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    std::vector<capture_note> notes;
    {
        std::lock_guard<std::mutex> lock(g_capture_lock);
        auto it = g_captures.find(mi::as_stream(stream));
        if (it != g_captures.end())
        {
            notes.swap(it->second);
            g_captures.erase(it);
        }
    }
    MI_HIP_CHECK(hipStreamEndCapture(mi::as_stream(stream), &graph));
    // A bank that keeps ring positions on the host passes them to its kernels by value: the captured launches are right
    // for every replay only if the sequence brings each such bank back to the positions it started from.
    for (const capture_note &n : notes)
        if (n.fn(n.bank) != n.sig)
        {
            // The banks' host-side positions have already moved on over the captured calls although no kernel ran.  Only a
            // signature of the starting positions was kept, so they cannot be rolled back: instead the captured work is
            // EXECUTED once here -- device rings and host positions agree again, exactly as if the calls had been made
            // eagerly -- and the graph is dropped.  The caller goes on with eager calls (bench.py does).
            hipGraphExec_t once = nullptr;
            hipError_t e = hipGraphInstantiate(&once, graph, nullptr, nullptr, 0);
            if (e == hipSuccess) e = hipGraphLaunch(once, mi::as_stream(stream));
            if (e == hipSuccess) e = hipStreamSynchronize(mi::as_stream(stream));
            if (once != nullptr) (void)hipGraphExecDestroy(once);
            (void)hipGraphDestroy(graph);
            if (e != hipSuccess)
                return mi::fail(MI_EHIP, "mi_dspu_graph_end_capture: the %s bank does not return to its starting positions over the "
                                "captured calls, and running them once instead failed (%s): reset every bank the capture touched",
                                n.what, hipGetErrorString(e));
            return mi::fail(MI_ESTATE, "mi_dspu_graph_end_capture: the %s bank does not return to its starting positions over the "
                            "captured calls -- capture a whole number of its position periods (DESIGN.md 3.7).  The captured calls "
                            "have been executed ONCE (device state and host positions agree); no graph was made", n.what);
        }
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    MI_HIP_CHECK(e);
    // the executable graph's device-side image goes up now instead of inside the first launch (best effort: a runtime that
    // cannot upload simply does it at that launch)
    if (hipGraphUpload(exec, mi::as_stream(stream)) != hipSuccess)
        (void)hipGetLastError();
    graph_handle *h = new graph_handle;
    h->exec = exec;
    h->banks.swap(notes);
    *graph_exec = h;
    return MI_OK;
}

int mi_dspu_graph_launch(void *graph_exec, void *stream)
{
    MI_REQUIRE(graph_exec != nullptr, MI_EINVAL, "mi_dspu_graph_launch: NULL graph");
    graph_handle *h = static_cast<graph_handle *>(graph_exec);
    {
        std::lock_guard<std::mutex> lock(g_capture_lock);
        for (const capture_note &n : h->banks)
            if (epoch_of(n.bank) != n.epoch)
                return mi::fail(MI_ESTATE, "mi_dspu_graph_launch: the %s bank has re-made its device buffers since this graph was "
                                "captured (a first batch of frames grows the convolver's ring): the graph's launches hold the old "
                                "ones -- capture again", n.what);
    }
    MI_HIP_CHECK(hipGraphLaunch(h->exec, mi::as_stream(stream)));
    return MI_OK;
}

int mi_dspu_graph_destroy(void *graph_exec)
{
    if (graph_exec != nullptr)
    {
        graph_handle *h = static_cast<graph_handle *>(graph_exec);
        const hipError_t e = hipGraphExecDestroy(h->exec);
        delete h;
        MI_HIP_CHECK(e);
    }
    return MI_OK;
}

} // extern "C"
