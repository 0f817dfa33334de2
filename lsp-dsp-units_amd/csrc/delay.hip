// Delay-line and ring-buffer banks for gfx950:
//   mi_delay_bank -- `channels` x lsp::dspu::Delay      (reference: src/main/util/Delay.cpp:51-582)
//   mi_ring_bank  -- `channels` x lsp::dspu::RingBuffer (reference: src/main/util/RingBuffer.cpp:48-209)
// Pure data movement: every kernel is a gather/scatter whose index arithmetic is the reference's own
// (unsigned 32-bit head/tail/size, same modulo expressions), so results are bit-exact by construction.
// Delays are per channel.  The channels of a bank advance together (one head, passed by value) unless the caller uses the
// *_rows calls, which move a subset only: every channel then carries an offset to the common head (util/Delay.h:35 -- each
// reference object has its own nHead), kept on the device and touched only by calls that move a subset.
#include "mi_common.h"

#include <cstdint>
#include <vector>

namespace
{
    constexpr uint32_t DELAY_GAP = 0x200;       // Delay.cpp:26

    // gain modes shared by the process variants (Delay.cpp:104-397)
    enum { G_NONE = 0, G_SCALAR = 1, G_VECTOR = 2 };

    // Which line a row of the call's buffers belongs to, and where that line's write position is: rows == NULL: row r is
    // channel r; off == NULL: every line sits at the common head.
    struct row_map { const uint32_t *rows; const uint32_t *off; };
    __device__ __forceinline__ uint32_t line_of(const row_map &m, uint32_t r) { return m.rows ? m.rows[r] : r; }
    __device__ __forceinline__ uint32_t head_of(const row_map &m, uint32_t ch, uint32_t head, uint32_t size)
    {
        return m.off ? (head + m.off[ch]) % size : head;
    }

    __device__ __forceinline__ float apply_gain(float v, int mode, float k, const float *gv, size_t i)
    {
        return (mode == G_SCALAR) ? v * k : (mode == G_VECTOR) ? v * gv[i] : v;
    }

    // ring[(head + i) % size] = src[i] for the last min(count, size) samples (Delay::append, Delay.cpp:76-102)
    __global__ __launch_bounds__(256)
    void ring_append_kernel(float *ring, uint32_t size, uint32_t head0, const float *src, size_t src_stride, size_t count,
                            const row_map rm)
    {
        const uint32_t r = blockIdx.y, ch = line_of(rm, r), head = head_of(rm, ch, head0, size);
        const size_t first = (count > size) ? count - size : 0;         // older samples would be overwritten anyway
        for (size_t i = first + size_t(blockIdx.x) * 256 + threadIdx.x; i < count; i += size_t(gridDim.x) * 256)
            ring[size_t(ch) * size + (head + i) % size] = src[size_t(r) * src_stride + i];
    }

    // the same with four samples per lane: count, the write position and first are multiples of four (16-byte cells)
    __global__ __launch_bounds__(256)
    void ring_append4_kernel(float *ring, uint32_t size, uint32_t head0, const float *src, size_t src_stride, size_t count,
                             const row_map rm)
    {
        const uint32_t r = blockIdx.y, ch = line_of(rm, r), head = head_of(rm, ch, head0, size);
        const size_t first = (count > size) ? count - size : 0;
        for (size_t i = first + 4 * (size_t(blockIdx.x) * 256 + threadIdx.x); i < count; i += 4 * size_t(gridDim.x) * 256)
            *reinterpret_cast<float4 *>(ring + size_t(ch) * size + (head + i) % size) =
                *reinterpret_cast<const float4 *>(src + size_t(r) * src_stride + i);
    }

    // dst[i] (+)= gain * ring[(tail_c + i) % size] -- the "shift data from buffer" half of Delay::process
    __global__ __launch_bounds__(256)
    void ring_read_kernel(float *dst, size_t dst_stride, const float *ring, uint32_t size, uint32_t head0,
                          const uint32_t *__restrict__ delay, size_t count, int add, int gmode, float k,
                          const float *gv, size_t gv_stride, const row_map rm)
    {
        const uint32_t r = blockIdx.y, ch = line_of(rm, r), head = head_of(rm, ch, head0, size);
        const uint32_t tail = (head + size - delay[ch]) % size;         // Delay.cpp:101
        for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < count; i += size_t(gridDim.x) * 256)
        {
            const float v = apply_gain(ring[size_t(ch) * size + (tail + i) % size], gmode, k, gv + size_t(r) * gv_stride, i);
            float *d = dst + size_t(r) * dst_stride + i;
            *d = add ? *d + v : v;
        }
    }

    // Not-aliased fast path of Delay::process: dst[i] (+)= gain * (i >= d ? src[i - d] : ring[(tail + i) % size]).
    // Same values as the push/pull pieces of the reference (a cell read `d` behind the write position holds
    // src[i - d] as soon as i >= d), without cutting the block into pieces.
    __global__ __launch_bounds__(256)
    void delay_direct_kernel(float *dst, size_t dst_stride, const float *src, size_t src_stride, const float *ring,
                             uint32_t size, uint32_t head0, const uint32_t *__restrict__ delay, size_t count,
                             int add, int gmode, float k, const float *gv, size_t gv_stride, const row_map rm)
    {
        const uint32_t r = blockIdx.y, ch = line_of(rm, r), head = head_of(rm, ch, head0, size);
        const uint32_t d = delay[ch];
        const uint32_t tail = (head + size - d) % size;
        const float *x = src + size_t(r) * src_stride;
        for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < count; i += size_t(gridDim.x) * 256)
        {
            const float raw = (i >= d) ? x[i - d] : ring[size_t(ch) * size + (tail + i) % size];
            const float v = apply_gain(raw, gmode, k, gv + size_t(r) * gv_stride, i);
            float *o = dst + size_t(r) * dst_stride + i;
            *o = add ? *o + v : v;
        }
    }

    // Blocks no longer than the shortest delay of the bank (and than size - longest delay): no cell read in this call
    // is written in this call, so "push then pull" collapses into one pass, in place or not:
    //   v = ring[(tail + i) % size];  ring[(head + i) % size] = src[i];  dst[i] (+)= gain * v
    __global__ __launch_bounds__(256)
    void delay_exchange_kernel(float *dst, size_t dst_stride, const float *src, size_t src_stride, float *ring,
                               uint32_t size, uint32_t head0, const uint32_t *__restrict__ delay, size_t count,
                               int add, int gmode, float k, const float *gv, size_t gv_stride, const row_map rm)
    {
        const uint32_t row = blockIdx.y, ch = line_of(rm, row), head = head_of(rm, ch, head0, size);
        const uint32_t tail = (head + size - delay[ch]) % size;
        float *r = ring + size_t(ch) * size;
        for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < count; i += size_t(gridDim.x) * 256)
        {
            const float x = src[size_t(row) * src_stride + i];
            const float v = apply_gain(r[(tail + i) % size], gmode, k, gv + size_t(row) * gv_stride, i);
            r[(head + i) % size] = x;
            float *o = dst + size_t(row) * dst_stride + i;
            *o = add ? *o + v : v;
        }
    }

    // The two steady-state kernels above with FOUR consecutive samples per lane (16-byte accesses): when the block, the write
    // position and every delay of the call are multiples of four (the line's length always is) no 16-byte cell straddles
    // the end of the line or the border between "still in the line" and "already in this block".  Same cells, same values.
    __device__ __forceinline__ float4 apply_gain4(float4 v, int mode, float k, const float *gv, size_t i)
    {
        if (mode == G_SCALAR)
            return make_float4(v.x * k, v.y * k, v.z * k, v.w * k);
        if (mode == G_VECTOR)
        {
            const float4 g = *reinterpret_cast<const float4 *>(gv + i);
            return make_float4(v.x * g.x, v.y * g.y, v.z * g.z, v.w * g.w);
        }
        return v;
    }

    __global__ __launch_bounds__(256)
    void delay_exchange4_kernel(float *dst, size_t dst_stride, const float *src, size_t src_stride, float *ring,
                                uint32_t size, uint32_t head0, const uint32_t *__restrict__ delay, size_t count,
                                int add, int gmode, float k, const float *gv, size_t gv_stride, const row_map rm)
    {
        const uint32_t row = blockIdx.y, ch = line_of(rm, row), head = head_of(rm, ch, head0, size);
        const uint32_t tail = (head + size - delay[ch]) % size;
        float *r = ring + size_t(ch) * size;
        for (size_t i = 4 * (size_t(blockIdx.x) * 256 + threadIdx.x); i < count; i += 4 * size_t(gridDim.x) * 256)
        {
            const float4 x = *reinterpret_cast<const float4 *>(src + size_t(row) * src_stride + i);
            const float4 v = apply_gain4(*reinterpret_cast<const float4 *>(r + (tail + i) % size), gmode, k, gv + size_t(row) * gv_stride, i);
            *reinterpret_cast<float4 *>(r + (head + i) % size) = x;
            float4 *o = reinterpret_cast<float4 *>(dst + size_t(row) * dst_stride + i);
            if (add)
            {
                const float4 p = *o;
                *o = make_float4(p.x + v.x, p.y + v.y, p.z + v.z, p.w + v.w);
            }
            else
                *o = v;
        }
    }

    __global__ __launch_bounds__(256)
    void delay_direct4_kernel(float *dst, size_t dst_stride, const float *src, size_t src_stride, const float *ring,
                              uint32_t size, uint32_t head0, const uint32_t *__restrict__ delay, size_t count,
                              int add, int gmode, float k, const float *gv, size_t gv_stride, const row_map rm)
    {
        const uint32_t r = blockIdx.y, ch = line_of(rm, r), head = head_of(rm, ch, head0, size);
        const uint32_t d = delay[ch];
        const uint32_t tail = (head + size - d) % size;
        const float *x = src + size_t(r) * src_stride;
        for (size_t i = 4 * (size_t(blockIdx.x) * 256 + threadIdx.x); i < count; i += 4 * size_t(gridDim.x) * 256)
        {
            const float4 raw = (i >= d) ? *reinterpret_cast<const float4 *>(x + (i - d))
                                        : *reinterpret_cast<const float4 *>(ring + size_t(ch) * size + (tail + i) % size);
            const float4 v = apply_gain4(raw, gmode, k, gv + size_t(r) * gv_stride, i);
            float4 *o = reinterpret_cast<float4 *>(dst + size_t(r) * dst_stride + i);
            if (add)
            {
                const float4 p = *o;
                *o = make_float4(p.x + v.x, p.y + v.y, p.z + v.z, p.w + v.w);
            }
            else
                *o = v;
        }
    }

    // Delay::process_ramping (Delay.cpp:399-546): the read position slides from the old delay to the new one.
    // Reproduces the reference's chunked write-then-read order in closed form: the sample read at output offset o
    // is the newest input written to that ring cell by the end of o's chunk, else the cell's old content.
    __global__ __launch_bounds__(256)
    void delay_ramp_kernel(float *dst, size_t dst_stride, const float *src, size_t src_stride, const float *ring,
                           uint32_t size, uint32_t head0, const uint32_t *__restrict__ old_delay,
                           const uint32_t *__restrict__ new_delay, size_t count, int gmode, float k,
                           const float *gv, size_t gv_stride, const row_map rm)
    {
        const uint32_t row = blockIdx.y, ch = line_of(rm, row);         // row of the call's buffers, line of the bank
        const uint32_t head = head_of(rm, ch, head0, size);
        const uint32_t od = old_delay[ch], nd = new_delay[ch];
        const uint32_t old_tail = (head + size - od) % size;
        const float *x = src + size_t(row) * src_stride;
        const float *rb = ring + size_t(ch) * size;
        if (od == nd)                                                   // Delay.cpp:402-406: plain process
        {
            for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < count; i += size_t(gridDim.x) * 256)
            {
                const float v = (i >= od) ? x[i - od] : rb[(old_tail + i) % size];
                dst[size_t(row) * dst_stride + i] = apply_gain(v, gmode, k, gv + size_t(row) * gv_stride, i);
            }
            return;
        }
        const size_t gap = size - ((nd > od) ? nd : od);               // free_gap
        const float delta = 1.0f + float(int64_t(od) - int64_t(nd)) / float(count);
        for (size_t o = size_t(blockIdx.x) * 256 + threadIdx.x; o < count; o += size_t(gridDim.x) * 256)
        {
            const size_t chunk_end = ((o / gap + 1) * gap < count) ? (o / gap + 1) * gap : count;
            const size_t tail = (size_t(old_tail) + size_t(int64_t(delta * float(o)))) % size;    // Delay.cpp:434
            const size_t i0 = (tail + size - head) % size;              // input index that lands on this cell
            float v;
            if (i0 < chunk_end)
                v = x[i0 + ((chunk_end - 1 - i0) / size) * size];       // newest write so far
            else
                v = rb[tail];
            dst[size_t(row) * dst_stride + o] = apply_gain(v, gmode, k, gv + size_t(row) * gv_stride, o);
        }
    }

    // RingBuffer::get(dst, offset, count) (RingBuffer.cpp:147-183) for every channel
    __global__ __launch_bounds__(256)
    void ring_get_kernel(float *dst, size_t dst_stride, const float *ring, uint32_t cap, uint32_t head,
                         size_t offset, size_t count)
    {
        const uint32_t ch = blockIdx.y;
        // leading zeros while the requested position is older than the buffer
        size_t lead = 0;
        size_t off = offset;
        if (off >= cap)
        {
            lead = (count < off - cap + 1) ? count : off - cap + 1;
            off -= lead;
        }
        const bool empty = (off >= cap);
        const size_t tail = empty ? 0 : (size_t(head) + cap - off - 1) % cap;
        const size_t rest = count - lead;
        const size_t to_read = empty ? 0 : ((rest < off + 1) ? rest : off + 1);
        for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < count; i += size_t(gridDim.x) * 256)
        {
            float v = 0.0f;
            if (i >= lead && i - lead < to_read)
                v = ring[size_t(ch) * cap + (tail + (i - lead)) % cap];
            dst[size_t(ch) * dst_stride + i] = v;
        }
    }

    __global__ __launch_bounds__(256)
    void fill_kernel(float *p, size_t n, float v)
    {
        for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256)
            p[i] = v;
    }

    inline dim3 grid_for(size_t count, uint32_t channels)
    {
        size_t gx = (count + 255) / 256;
        if (gx > 64) gx = 64;
        if (gx == 0) gx = 1;
        return dim3(uint32_t(gx), channels);
    }
} // namespace

struct mi_delay_bank
{
    uint32_t    channels = 0, size = 0, head = 0;
    std::vector<uint32_t> delay;
    float      *d_ring = nullptr, *d_scratch = nullptr;
    size_t      scratch_floats = 0;
    uint32_t   *d_delay = nullptr, *d_delay_new = nullptr;
    bool        delay_dirty = true;
    // lines with positions of their own (the *_rows calls): offset of every line's write position to `head`; all zero --
    // and the kernels told nothing about it -- until a call moves a subset
    // The offsets live on the device and are moved there (offsets_update_kernel); `off` is the host's mirror of them, kept
    // by the same arithmetic, never uploaded.  The row list of a call is the caller's memory: it goes through one of a few
    // pinned staging slots (a slot is taken again only after the copy out of it has run), so a *_rows call blocks on nothing.
    std::vector<uint32_t> off;
    uint32_t   *d_off = nullptr, *d_rows = nullptr;
    uint32_t    rows_cap = 0;
    bool        off_any = false;
    static constexpr int ROW_SLOTS = 4;
    uint32_t   *h_rows = nullptr;                   // pinned: [ROW_SLOTS][rows_cap]
    hipEvent_t  row_ev[ROW_SLOTS] = { nullptr, nullptr, nullptr, nullptr };
    bool        row_busy[ROW_SLOTS] = { false, false, false, false };
    int         row_slot = 0;
};

// the lines a call moves: all of them (rows == nullptr, n == channels) or the listed ones
struct delay_rows
{
    const uint32_t *host = nullptr;     // channel of every row of the call's buffers
    uint32_t        n = 0;
    row_map         rm = { nullptr, nullptr };
};

namespace
{
    int sync_delays(mi_delay_bank *b, hipStream_t st)
    {
        if (!b->delay_dirty)
            return MI_OK;
        MI_HIP_CHECK(hipMemcpyAsync(b->d_delay, b->delay.data(), b->channels * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        MI_HIP_CHECK(hipStreamSynchronize(st));
        b->delay_dirty = false;
        return MI_OK;
    }

    // off[line] = value (set) or (off[line] + value) % size, for the listed lines (rows == nullptr: lines 0 .. n - 1)
    __global__ void offsets_update_kernel(uint32_t *off, const uint32_t *rows, uint32_t n, uint32_t size, uint32_t value, int set)
    {
        const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
        if (k >= n)
            return;
        const uint32_t c = (rows != nullptr) ? rows[k] : k;
        off[c] = set ? value : (off[c] + value) % size;
    }

    int update_offsets(mi_delay_bank *b, const delay_rows &dr, uint32_t value, int set, hipStream_t st)
    {
        hipLaunchKernelGGL(offsets_update_kernel, dim3((dr.n + 255) / 256), dim3(256), 0, st, b->d_off, dr.rm.rows, dr.n, b->size, value, set);
        MI_HIP_CHECK(hipGetLastError());
        return MI_OK;
    }

    bool capturing(hipStream_t st)
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        return st != nullptr && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    }

    // The lines of a call.  hrows == nullptr: the whole bank.  The kernels hear about offsets only when there are any.
    int make_rows(mi_delay_bank *b, const uint32_t *hrows, uint32_t n_rows, delay_rows *dr, hipStream_t st)
    {
        dr->host = hrows;
        dr->n = (hrows != nullptr) ? n_rows : b->channels;
        if (hrows != nullptr)
        {
            // the row list is copied at call time: a replay of a captured call would move whatever the staging slot holds then
            MI_REQUIRE(!capturing(st), MI_ESTATE, "a call on a subset of the lines (*_rows) cannot be captured into a graph: its row "
                       "list is read from the caller's memory when the call is made");
            if (b->off.empty())
                b->off.assign(b->channels, 0);
            if (n_rows > b->rows_cap)                       // (rare: the staging grows; what is in flight has to land first)
            {
                MI_HIP_CHECK(hipStreamSynchronize(st));
                for (int k = 0; k < mi_delay_bank::ROW_SLOTS; ++k)
                    b->row_busy[k] = false;
                (void)hipFree(b->d_rows);
                (void)hipHostFree(b->h_rows);
                b->d_rows = nullptr;
                b->h_rows = nullptr;
                b->rows_cap = 0;
                const uint32_t cap = (n_rows > 64) ? n_rows : 64;
                MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_rows), cap * sizeof(uint32_t)));
                MI_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&b->h_rows), size_t(mi_delay_bank::ROW_SLOTS) * cap * sizeof(uint32_t),
                                           hipHostMallocDefault));
                b->rows_cap = cap;
            }
            const int slot = b->row_slot;
            b->row_slot = (slot + 1) % mi_delay_bank::ROW_SLOTS;
            if (b->row_ev[slot] == nullptr)
                MI_HIP_CHECK(hipEventCreateWithFlags(&b->row_ev[slot], hipEventDisableTiming));
            if (b->row_busy[slot])
                MI_HIP_CHECK(hipEventSynchronize(b->row_ev[slot]));        // ROW_SLOTS calls ago: long done
            uint32_t *stage = b->h_rows + size_t(slot) * b->rows_cap;
            std::copy(hrows, hrows + n_rows, stage);
            MI_HIP_CHECK(hipMemcpyAsync(b->d_rows, stage, n_rows * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            MI_HIP_CHECK(hipEventRecord(b->row_ev[slot], st));
            b->row_busy[slot] = true;
            if (b->d_off == nullptr)                        // from here on the lines no longer move as one
            {
                MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_off), b->channels * sizeof(uint32_t)));
                MI_HIP_CHECK(hipMemsetAsync(b->d_off, 0, b->channels * sizeof(uint32_t), st));
            }
            b->off_any = true;
        }
        dr->rm.rows = (hrows != nullptr) ? b->d_rows : nullptr;
        dr->rm.off = b->off_any ? b->d_off : nullptr;
        return MI_OK;
    }

    // what a call did to the positions: a whole-bank call has moved `head`; a call on a subset leaves `head` where it was
    // and moves the offsets of its lines by the same amount, on the device (behind the call's kernels) and in the mirror
    int settle_rows(mi_delay_bank *b, const delay_rows &dr, uint32_t head_before, hipStream_t st)
    {
        if (dr.host == nullptr)
            return MI_OK;
        const uint32_t delta = (b->head + b->size - head_before) % b->size;
        b->head = head_before;
        if (delta == 0)
            return MI_OK;
        for (uint32_t r = 0; r < dr.n; ++r)
            b->off[dr.host[r]] = (b->off[dr.host[r]] + delta) % b->size;
        return update_offsets(b, dr, delta, 0, st);
    }

    int append(mi_delay_bank *b, const delay_rows &dr, const float *src, size_t stride, size_t count, hipStream_t st)
    {
        bool quads = (count % 4 == 0) && (b->head % 4 == 0) && (stride % 4 == 0) && (reinterpret_cast<uintptr_t>(src) % 16 == 0);
        for (uint32_t k = 0; quads && b->off_any && k < dr.n; ++k)
            quads = b->off[dr.host ? dr.host[k] : k] % 4 == 0;
        if (quads)
            hipLaunchKernelGGL(ring_append4_kernel, grid_for(count / 4, dr.n), dim3(256), 0, st,
                               b->d_ring, b->size, b->head, src, stride, count, dr.rm);
        else
        hipLaunchKernelGGL(ring_append_kernel, grid_for(count, dr.n), dim3(256), 0, st,
                           b->d_ring, b->size, b->head, src, stride, count, dr.rm);
        MI_HIP_CHECK(hipGetLastError());
        b->head = uint32_t((size_t(b->head) + count) % b->size);
        return MI_OK;
    }

    // Delay::append(): a whole buffer or more keeps the last nSize samples from cell 0 on and restarts the write position
    // there (Delay.cpp:95-99).  The absolute position matters: process_ramping's read index wraps modulo 2^64 before
    // it is reduced modulo nSize (Delay.cpp:434), which depends on where the tail sits when the delay grows quickly.
    int append_block(mi_delay_bank *b, delay_rows &dr, const float *src, size_t stride, size_t count, hipStream_t st)
    {
        if (count < b->size)
            return append(b, dr, src, stride, count, st);
        // every line of the call restarts at cell 0
        int r = MI_OK;
        if (dr.host == nullptr)
        {
            b->head = 0;
            if (b->off_any)
            {
                std::fill(b->off.begin(), b->off.end(), 0u);
                MI_HIP_CHECK(hipMemsetAsync(b->d_off, 0, b->channels * sizeof(uint32_t), st));
            }
        }
        else
        {
            const uint32_t v = (b->size - b->head % b->size) % b->size;
            for (uint32_t k = 0; k < dr.n; ++k)
                b->off[dr.host[k]] = v;
            r = update_offsets(b, dr, v, 1, st);
            if (r != MI_OK)
                return r;
        }
        const uint32_t h = b->head;
        r = append(b, dr, src + (count - b->size), stride, b->size, st);
        b->head = h;                                                    // a whole lap: the position is where it was
        return r;
    }

    // a private copy of the caller's input when dst aliases src
    int stage_input(mi_delay_bank *b, uint32_t rows, const float **src, size_t *stride, size_t count, hipStream_t st)
    {
        const size_t need = size_t(rows) * count;
        if (need > b->scratch_floats)
        {
            (void)hipFree(b->d_scratch);
            b->d_scratch = nullptr;
            b->scratch_floats = 0;
            MI_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&b->d_scratch), need * sizeof(float)));
            b->scratch_floats = need;
        }
        MI_HIP_CHECK(hipMemcpy2DAsync(b->d_scratch, count * sizeof(float), *src, *stride * sizeof(float),
                                      count * sizeof(float), rows, hipMemcpyDeviceToDevice, st));
        *src = b->d_scratch;
        *stride = count;
        return MI_OK;
    }
} // namespace

namespace mi
{
    int delay_bank_view(mi_delay_bank_t *b, delay_view *v)
    {
        MI_REQUIRE(b != nullptr && v != nullptr, MI_ESTATE, "delay_bank_view: NULL bank");
        v->ring = b->d_ring;
        v->size = b->size;
        v->head = b->head;
        v->delay = b->delay.empty() ? 0 : b->delay[0];
        for (uint32_t d : b->delay)
            if (d != v->delay)
                v->delay = UINT32_MAX;
        if (b->off_any)                                     // lines at positions of their own: no common view
            v->delay = UINT32_MAX;
        return MI_OK;
    }

    void delay_bank_advance(mi_delay_bank_t *b, size_t samples)
    {
        b->head = uint32_t((size_t(b->head) + samples) % b->size);
    }

    // what the launches of a call take by value from the host: the head and the delays
    uint64_t delay_bank_positions(const void *bank)
    {
        const mi_delay_bank *b = static_cast<const mi_delay_bank *>(bank);
        uint64_t h = b->head;
        for (uint32_t d : b->delay)
            h = position_mix(h, d);
        if (b->off_any)
            for (uint32_t o : b->off)
                h = position_mix(h, o);
        return h;
    }
} // namespace mi

extern "C" {

int mi_delay_bank_create(mi_delay_bank_t **bank, uint32_t channels, size_t max_size)
{
    MI_REQUIRE(bank != nullptr, MI_EINVAL, "mi_delay_bank_create: NULL result pointer");
    *bank = nullptr;
    MI_REQUIRE(channels > 0, MI_EINVAL, "mi_delay_bank_create: channels must be > 0");
    MI_REQUIRE(mi_dspu_device_count() > 0, MI_ENODEV, "no HIP device available (there is no CPU fallback)");
    mi_delay_bank *b = new (std::nothrow) mi_delay_bank();
    MI_REQUIRE(b != nullptr, MI_ENOMEM, "mi_delay_bank_create: out of host memory");
    b->channels = channels;
    // Delay.cpp:53: size = align_size(max_size + DELAY_GAP, DELAY_GAP)
    b->size = uint32_t(((max_size + DELAY_GAP + DELAY_GAP - 1) / DELAY_GAP) * DELAY_GAP);
    b->delay.assign(channels, 0);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&b->d_ring), size_t(channels) * b->size * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_delay), channels * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b->d_delay_new), channels * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(b->d_ring, 0, size_t(channels) * b->size * sizeof(float));
    if (e != hipSuccess)
    {
        mi_delay_bank_destroy(b);
        return mi::fail(e == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP, "mi_delay_bank_create: %s", hipGetErrorString(e));
    }
    *bank = b;
    return MI_OK;
}

int mi_delay_bank_destroy(mi_delay_bank_t *b)
{
    if (b == nullptr)
        return MI_OK;
    (void)hipFree(b->d_ring); (void)hipFree(b->d_scratch); (void)hipFree(b->d_delay); (void)hipFree(b->d_delay_new);
    (void)hipFree(b->d_off); (void)hipFree(b->d_rows);
    (void)hipHostFree(b->h_rows);
    for (int k = 0; k < mi_delay_bank::ROW_SLOTS; ++k)
        if (b->row_ev[k] != nullptr)
            (void)hipEventDestroy(b->row_ev[k]);
    delete b;
    return MI_OK;
}

int mi_delay_bank_set_delay(mi_delay_bank_t *b, uint32_t channel, size_t delay)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_delay_bank_set_delay: NULL bank");
    const uint32_t d = uint32_t(delay % b->size);                      // Delay.cpp:571
    if (channel == UINT32_MAX)
        std::fill(b->delay.begin(), b->delay.end(), d);
    else
    {
        MI_REQUIRE(channel < b->channels, MI_EINVAL, "mi_delay_bank_set_delay: channel %u out of range", channel);
        b->delay[channel] = d;
    }
    b->delay_dirty = true;
    return MI_OK;
}

int mi_delay_bank_get(const mi_delay_bank_t *b, uint32_t channel, uint32_t *delay, uint32_t *size, uint32_t *head, uint32_t *tail)
{
    MI_REQUIRE(b != nullptr && channel < b->channels, MI_EINVAL, "mi_delay_bank_get: bad argument");
    if (delay) *delay = b->delay[channel];
    if (size)  *size = b->size;
    const uint32_t h = b->off_any ? (b->head + b->off[channel]) % b->size : b->head;
    if (head)  *head = h;
    if (tail)  *tail = (h + b->size - b->delay[channel]) % b->size;
    return MI_OK;
}

int mi_delay_bank_clear(mi_delay_bank_t *b, void *stream)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_delay_bank_clear: NULL bank");
    MI_HIP_CHECK(hipMemsetAsync(b->d_ring, 0, size_t(b->channels) * b->size * sizeof(float), mi::as_stream(stream)));
    return MI_OK;
}

static int delay_append_impl(mi_delay_bank_t *b, const uint32_t *rows, uint32_t n_rows, const float *in, size_t count,
                             size_t in_stride, void *stream, const char *who)
{
    MI_REQUIRE(b != nullptr && (count == 0 || in != nullptr), MI_EINVAL, "%s: bad argument", who);
    if (count == 0 || (rows != nullptr && n_rows == 0))
        return MI_OK;
    hipStream_t st = mi::as_stream(stream);
    int r = mi::capture_touch(st, b, "delay", mi::delay_bank_positions);
    if (r != MI_OK)
        return r;
    delay_rows dr;
    r = make_rows(b, rows, n_rows, &dr, st);
    if (r != MI_OK)
        return r;
    const uint32_t head_before = b->head;
    r = append_block(b, dr, in, in_stride, count, st);
    const int rs = settle_rows(b, dr, head_before, st);
    return (r != MI_OK) ? r : rs;
}

static int delay_process_impl(mi_delay_bank_t *b, const uint32_t *rows, uint32_t n_rows, float *out, const float *in,
                              size_t count, size_t out_stride, size_t in_stride, int add, int gain_mode, float gain,
                              const float *gain_vec, size_t gain_stride, void *stream, const char *who)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "%s: NULL bank", who);
    if (count == 0 || (rows != nullptr && n_rows == 0))
        return MI_OK;
    MI_REQUIRE(out != nullptr && in != nullptr, MI_EINVAL, "%s: NULL buffer", who);
    MI_REQUIRE(gain_mode != G_VECTOR || gain_vec != nullptr, MI_EINVAL, "%s: NULL gain vector", who);
    hipStream_t st = mi::as_stream(stream);
    int r = mi::capture_touch(st, b, "delay", mi::delay_bank_positions);
    if (r != MI_OK)
        return r;
    r = sync_delays(b, st);
    if (r != MI_OK)
        return r;
    delay_rows dr;
    r = make_rows(b, rows, n_rows, &dr, st);
    if (r != MI_OK)
        return r;
    const uint32_t head_before = b->head;
    // The reference alternates "push to_do samples / pull to_do samples" in pieces of at most size - delay
    // (Delay.cpp:113-142) so that a pull never reads a cell a later push of the same call already overwrote.
    // Same order here, with the piece bounded by the largest delay among the lines of the call.
    uint32_t dmin = UINT32_MAX, dmax = 0;
    for (uint32_t k = 0; k < dr.n; ++k)
    {
        const uint32_t d = b->delay[dr.host ? dr.host[k] : k];
        dmin = (d < dmin) ? d : dmin;
        dmax = (d > dmax) ? d : dmax;
    }
    // four samples per lane when nothing of the call straddles a 16-byte cell
    bool quads = (count % 4 == 0) && (b->head % 4 == 0) && (out_stride % 4 == 0) && (in_stride % 4 == 0) &&
                 ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(in)) % 16 == 0) &&
                 (gain_mode != G_VECTOR || (gain_stride % 4 == 0 && reinterpret_cast<uintptr_t>(gain_vec) % 16 == 0));
    for (uint32_t k = 0; quads && k < dr.n; ++k)
    {
        const uint32_t c = dr.host ? dr.host[k] : k;
        quads = (b->delay[c] % 4 == 0) && (!b->off_any || b->off[c] % 4 == 0);
    }
    if (count <= dmin && count <= size_t(b->size - dmax))
    {
        if (quads)
            hipLaunchKernelGGL(delay_exchange4_kernel, grid_for(count / 4, dr.n), dim3(256), 0, st,
                               out, out_stride, in, in_stride, b->d_ring, b->size, b->head, b->d_delay, count, add, gain_mode,
                               gain, gain_vec, gain_stride, dr.rm);
        else
        hipLaunchKernelGGL(delay_exchange_kernel, grid_for(count, dr.n), dim3(256), 0, st,
                           out, out_stride, in, in_stride, b->d_ring, b->size, b->head, b->d_delay, count, add, gain_mode,
                           gain, gain_vec, gain_stride, dr.rm);
        MI_HIP_CHECK(hipGetLastError());
        b->head = uint32_t((size_t(b->head) + count) % b->size);
        return settle_rows(b, dr, head_before, st);
    }
    if (static_cast<const void *>(out) != static_cast<const void *>(in))
    {
        if (quads)
            hipLaunchKernelGGL(delay_direct4_kernel, grid_for(count / 4, dr.n), dim3(256), 0, st,
                               out, out_stride, in, in_stride, b->d_ring, b->size, b->head, b->d_delay, count, add, gain_mode,
                               gain, gain_vec, gain_stride, dr.rm);
        else
        hipLaunchKernelGGL(delay_direct_kernel, grid_for(count, dr.n), dim3(256), 0, st,
                           out, out_stride, in, in_stride, b->d_ring, b->size, b->head, b->d_delay, count, add, gain_mode,
                           gain, gain_vec, gain_stride, dr.rm);
        MI_HIP_CHECK(hipGetLastError());
        r = append(b, dr, in, in_stride, count, st);
        const int rs = settle_rows(b, dr, head_before, st);
        return (r != MI_OK) ? r : rs;
    }
    if (dmax == 0)
    {
        // In place without a delay the reference appends the block as a whole and then scales it (Delay.cpp:107-111,
        // 155-160, 204-209, 254-259, 303-308, 352-357): a block of at least the line's length restarts the line at cell 0
        // (:95-99), and the absolute position matters to a later process_ramping() (:434).
        r = append_block(b, dr, in, in_stride, count, st);
        if (r == MI_OK && (add || gain_mode != G_NONE))
        {
            hipLaunchKernelGGL(delay_direct_kernel, grid_for(count, dr.n), dim3(256), 0, st,
                               out, out_stride, in, in_stride, b->d_ring, b->size, b->head, b->d_delay, count, add, gain_mode,
                               gain, gain_vec, gain_stride, dr.rm);
            if (hipGetLastError() != hipSuccess)
                r = mi::fail(MI_EHIP, "%s: launch failed", who);
        }
        const int rs = settle_rows(b, dr, head_before, st);
        return (r != MI_OK) ? r : rs;
    }
    const size_t gap = b->size - dmax;
    size_t done = 0;
    while (done < count && r == MI_OK)
    {
        const size_t n = (count - done < gap) ? count - done : gap;
        const uint32_t head_piece = b->head;
        r = append(b, dr, in + done, in_stride, n, st);
        if (r != MI_OK)
            break;
        hipLaunchKernelGGL(ring_read_kernel, grid_for(n, dr.n), dim3(256), 0, st,
                           out + done, out_stride, b->d_ring, b->size, head_piece, b->d_delay, n, add, gain_mode, gain,
                           gain_vec ? gain_vec + done : nullptr, gain_stride, dr.rm);
        if (hipGetLastError() != hipSuccess)
            r = mi::fail(MI_EHIP, "%s: launch failed", who);
        done += n;
    }
    const int rs = settle_rows(b, dr, head_before, st);
    return (r != MI_OK) ? r : rs;
}

static int delay_check_rows(const mi_delay_bank_t *b, const uint32_t *rows, uint32_t n_rows, const char *who)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "%s: NULL bank", who);
    MI_REQUIRE(rows != nullptr || n_rows == 0, MI_EINVAL, "%s: NULL row list", who);
    MI_REQUIRE(n_rows <= b->channels, MI_EINVAL, "%s: more rows than the bank has lines", who);
    std::vector<uint8_t> seen(b->channels, 0);
    for (uint32_t r = 0; r < n_rows; ++r)
    {
        MI_REQUIRE(rows[r] < b->channels, MI_EINVAL, "%s: line %u out of range", who, rows[r]);
        MI_REQUIRE(!seen[rows[r]], MI_EINVAL, "%s: line %u listed twice", who, rows[r]);
        seen[rows[r]] = 1;
    }
    return MI_OK;
}

int mi_delay_bank_append(mi_delay_bank_t *b, const float *in, size_t count, size_t in_stride, void *stream)
{
    return delay_append_impl(b, nullptr, 0, in, count, in_stride, stream, "mi_delay_bank_append");
}

int mi_delay_bank_append_rows(mi_delay_bank_t *b, const uint32_t *rows, uint32_t n_rows, const float *in, size_t count,
                              size_t in_stride, void *stream)
{
    const int r = delay_check_rows(b, rows, n_rows, "mi_delay_bank_append_rows");
    if (r != MI_OK)
        return r;
    return delay_append_impl(b, rows, n_rows, in, count, in_stride, stream, "mi_delay_bank_append_rows");
}

int mi_delay_bank_process(mi_delay_bank_t *b, float *out, const float *in, size_t count, size_t out_stride,
                          size_t in_stride, int add, int gain_mode, float gain, const float *gain_vec,
                          size_t gain_stride, void *stream)
{
    return delay_process_impl(b, nullptr, 0, out, in, count, out_stride, in_stride, add, gain_mode, gain, gain_vec,
                              gain_stride, stream, "mi_delay_bank_process");
}

int mi_delay_bank_process_rows(mi_delay_bank_t *b, const uint32_t *rows, uint32_t n_rows, float *out, const float *in,
                               size_t count, size_t out_stride, size_t in_stride, int add, int gain_mode, float gain,
                               const float *gain_vec, size_t gain_stride, void *stream)
{
    const int r = delay_check_rows(b, rows, n_rows, "mi_delay_bank_process_rows");
    if (r != MI_OK)
        return r;
    return delay_process_impl(b, rows, n_rows, out, in, count, out_stride, in_stride, add, gain_mode, gain, gain_vec,
                              gain_stride, stream, "mi_delay_bank_process_rows");
}

static int delay_ramping_impl(mi_delay_bank_t *b, const uint32_t *rows, uint32_t n_rows, float *out, const float *in,
                              const uint32_t *new_delays, size_t count, size_t out_stride, size_t in_stride, int gain_mode, float gain,
                              const float *gain_vec, size_t gain_stride, void *stream, const char *who)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "%s: NULL bank", who);
    MI_REQUIRE(new_delays != nullptr, MI_EINVAL, "%s: NULL delays", who);
    if (count == 0 || (rows != nullptr && n_rows == 0))                 // Delay.cpp:407-408
        return MI_OK;
    MI_REQUIRE(out != nullptr && in != nullptr, MI_EINVAL, "%s: NULL buffer", who);
    hipStream_t st = mi::as_stream(stream);
    int r = mi::capture_touch(st, b, "delay", mi::delay_bank_positions);
    if (r != MI_OK)
        return r;
    r = sync_delays(b, st);
    if (r != MI_OK)
        return r;
    // the delays the lines of the call slide to; the other lines keep theirs (row r of new_delays belongs to line rows[r])
    const uint32_t n = (rows != nullptr) ? n_rows : b->channels;
    std::vector<uint32_t> nd(b->delay);
    for (uint32_t k = 0; k < n; ++k)
    {
        MI_REQUIRE(new_delays[k] < b->size, MI_EINVAL, "%s: delay %u does not fit the line", who, new_delays[k]);
        nd[rows ? rows[k] : k] = new_delays[k];
    }
    MI_HIP_CHECK(hipMemcpyAsync(b->d_delay_new, nd.data(), nd.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    MI_HIP_CHECK(hipStreamSynchronize(st));
    const float *src = in;
    size_t stride = in_stride;
    if (static_cast<const void *>(out) == static_cast<const void *>(in))
    {
        r = stage_input(b, n, &src, &stride, count, st);
        if (r != MI_OK)
            return r;
    }
    delay_rows dr;
    r = make_rows(b, rows, n_rows, &dr, st);
    if (r != MI_OK)
        return r;
    const uint32_t head_before = b->head;
    hipLaunchKernelGGL(delay_ramp_kernel, grid_for(count, dr.n), dim3(256), 0, st,
                       out, out_stride, src, stride, b->d_ring, b->size, b->head, b->d_delay, b->d_delay_new,
                       count, gain_mode, gain, gain_vec, gain_stride, dr.rm);
    MI_HIP_CHECK(hipGetLastError());
    r = append(b, dr, src, stride, count, st);
    const int rs = settle_rows(b, dr, head_before, st);
    if (r != MI_OK || rs != MI_OK)
        return (r != MI_OK) ? r : rs;
    b->delay = nd;                                                      // Delay.cpp:444-445
    b->delay_dirty = true;
    return MI_OK;
}

int mi_delay_bank_process_ramping(mi_delay_bank_t *b, float *out, const float *in, const uint32_t *new_delays,
                                  size_t count, size_t out_stride, size_t in_stride, int gain_mode, float gain,
                                  const float *gain_vec, size_t gain_stride, void *stream)
{
    return delay_ramping_impl(b, nullptr, 0, out, in, new_delays, count, out_stride, in_stride, gain_mode, gain, gain_vec, gain_stride,
                              stream, "mi_delay_bank_process_ramping");
}

int mi_delay_bank_process_ramping_rows(mi_delay_bank_t *b, const uint32_t *rows, uint32_t n_rows, float *out, const float *in,
                                       const uint32_t *new_delays, size_t count, size_t out_stride, size_t in_stride, int gain_mode,
                                       float gain, const float *gain_vec, size_t gain_stride, void *stream)
{
    const int r = delay_check_rows(b, rows, n_rows, "mi_delay_bank_process_ramping_rows");
    if (r != MI_OK)
        return r;
    return delay_ramping_impl(b, rows, n_rows, out, in, new_delays, count, out_stride, in_stride, gain_mode, gain, gain_vec, gain_stride,
                              stream, "mi_delay_bank_process_ramping_rows");
}

} // extern "C"

// =============================================================================================================
struct mi_ring_bank
{
    uint32_t    channels = 0, capacity = 0, head = 0;
    float      *d_ring = nullptr;
    bool        host_shared = false;        // storage is pinned host memory mapped into the device
};

static uint64_t ring_bank_positions(const void *bank) { return static_cast<const mi_ring_bank *>(bank)->head; }

extern "C" {

static int ring_bank_create(mi_ring_bank_t **bank, uint32_t channels, size_t size, float fill, float **host_view);

int mi_ring_bank_create(mi_ring_bank_t **bank, uint32_t channels, size_t size, float fill)
{
    return ring_bank_create(bank, channels, size, fill, nullptr);
}

int mi_ring_bank_create_shared(mi_ring_bank_t **bank, uint32_t channels, size_t size, float fill, float **host_view)
{
    MI_REQUIRE(host_view != nullptr, MI_EINVAL, "mi_ring_bank_create_shared: NULL view pointer");
    *host_view = nullptr;
    return ring_bank_create(bank, channels, size, fill, host_view);
}

static int ring_bank_create(mi_ring_bank_t **bank, uint32_t channels, size_t size, float fill, float **host_view)
{
    MI_REQUIRE(bank != nullptr, MI_EINVAL, "mi_ring_bank_create: NULL result pointer");
    *bank = nullptr;
    MI_REQUIRE(channels > 0 && size > 0, MI_EINVAL, "mi_ring_bank_create: channels and size must be > 0");
    MI_REQUIRE(mi_dspu_device_count() > 0, MI_ENODEV, "no HIP device available (there is no CPU fallback)");
    mi_ring_bank *b = new (std::nothrow) mi_ring_bank();
    MI_REQUIRE(b != nullptr, MI_ENOMEM, "mi_ring_bank_create: out of host memory");
    b->channels = channels;
    b->capacity = uint32_t(size);
    const size_t bytes = size_t(channels) * size * sizeof(float);
    b->host_shared = (host_view != nullptr);
    // shared: pinned host memory that the device addresses through the same pointer (coherent), so a host that reads or
    // writes the raw storage (RingBuffer::data(), util/RingBuffer.h:130) and the kernels see the same cells
    hipError_t e = b->host_shared ? hipHostMalloc(reinterpret_cast<void **>(&b->d_ring), bytes, hipHostMallocMapped)
                                  : hipMalloc(reinterpret_cast<void **>(&b->d_ring), bytes);
    if (e != hipSuccess)
    {
        delete b;
        return mi::fail(e == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP, "mi_ring_bank_create: %s", hipGetErrorString(e));
    }
    *bank = b;
    const int r = mi_ring_bank_fill(b, fill, nullptr);                  // RingBuffer::init fills (RingBuffer.cpp:48-63)
    if (r == MI_OK && host_view != nullptr)
    {
        MI_HIP_CHECK(hipStreamSynchronize(nullptr));
        *host_view = b->d_ring;
    }
    return r;
}

int mi_ring_bank_destroy(mi_ring_bank_t *b)
{
    if (b == nullptr)
        return MI_OK;
    if (b->host_shared)
        (void)hipHostFree(b->d_ring);
    else
        (void)hipFree(b->d_ring);
    delete b;
    return MI_OK;
}

int mi_ring_bank_fill(mi_ring_bank_t *b, float value, void *stream)    // clear() / fill(), RingBuffer.cpp:108-120
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_ring_bank_fill: NULL bank");
    {
        const int r = mi::capture_touch(mi::as_stream(stream), b, "ring buffer", ring_bank_positions);
        if (r != MI_OK)
            return r;
    }
    b->head = 0;
    const size_t n = size_t(b->channels) * b->capacity;
    hipLaunchKernelGGL(fill_kernel, dim3(uint32_t((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0,
                       mi::as_stream(stream), b->d_ring, n, value);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int mi_ring_bank_append(mi_ring_bank_t *b, const float *in, size_t count, size_t in_stride, size_t *appended, void *stream)
{
    MI_REQUIRE(b != nullptr && (count == 0 || in != nullptr), MI_EINVAL, "mi_ring_bank_append: bad argument");
    if (appended)
        *appended = (count > b->capacity) ? b->capacity : count;        // RingBuffer.cpp:78-83,105
    if (count == 0)
        return MI_OK;
    {
        const int r = mi::capture_touch(mi::as_stream(stream), b, "ring buffer", ring_bank_positions);
        if (r != MI_OK)
            return r;
    }
    const bool whole = count > b->capacity;                             // keeps the newest `capacity` from cell 0 on,
    if (whole)                                                          // and the head stays there (RingBuffer.cpp:78-83)
    {
        in += count - b->capacity;
        count = b->capacity;
        b->head = 0;
    }
    hipLaunchKernelGGL(ring_append_kernel, grid_for(count, b->channels), dim3(256), 0, mi::as_stream(stream),
                       b->d_ring, b->capacity, b->head, in, in_stride, count, row_map{ nullptr, nullptr });
    MI_HIP_CHECK(hipGetLastError());
    // RingBuffer.cpp:85-103: the head wraps only when the block runs PAST the end; a block that ends exactly at the end
    // leaves nHead == nCapacity (head_position() reports it; every position is taken modulo the capacity afterwards)
    const size_t end = size_t(b->head) + count;
    b->head = whole ? 0 : uint32_t((end > b->capacity) ? end - b->capacity : end);
    return MI_OK;
}

int mi_ring_bank_get(mi_ring_bank_t *b, float *out, size_t offset, size_t count, size_t out_stride, size_t *read, void *stream)
{
    MI_REQUIRE(b != nullptr && (count == 0 || out != nullptr), MI_EINVAL, "mi_ring_bank_get: bad argument");
    // return value of RingBuffer::get(dst, offset, count), RingBuffer.cpp:147-183
    size_t off = offset, cnt = count, got = 0;
    if (off >= b->capacity)
    {
        const size_t lead = (cnt < off - b->capacity + 1) ? cnt : off - b->capacity + 1;
        off -= lead;
        cnt -= lead;
    }
    if (off < b->capacity)
        got = (cnt < off + 1) ? cnt : off + 1;
    if (read)
        *read = got;
    if (count == 0)
        return MI_OK;
    {
        const int r = mi::capture_touch(mi::as_stream(stream), b, "ring buffer", ring_bank_positions);
        if (r != MI_OK)
            return r;
    }
    hipLaunchKernelGGL(ring_get_kernel, grid_for(count, b->channels), dim3(256), 0, mi::as_stream(stream),
                       out, out_stride, b->d_ring, b->capacity, b->head, offset, count);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int mi_ring_bank_info(const mi_ring_bank_t *b, size_t offset, uint32_t *capacity, uint32_t *head, uint32_t *tail_position)
{
    MI_REQUIRE(b != nullptr, MI_ESTATE, "mi_ring_bank_info: NULL bank");
    if (capacity) *capacity = b->capacity;
    if (head)     *head = b->head;
    if (tail_position)                                                  // RingBuffer.cpp:140-145
        *tail_position = (offset < b->capacity) ? uint32_t((size_t(b->head) + b->capacity - offset - 1) % b->capacity) : b->head;
    return MI_OK;
}

} // extern "C"
