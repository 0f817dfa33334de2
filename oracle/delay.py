"""
ORACLE (test infrastructure only).  Restatement of lsp::dspu::Delay (/root/reference/src/main/util/Delay.cpp:51-582)
and lsp::dspu::RingBuffer (src/main/util/RingBuffer.cpp:48-209) with the reference's own index arithmetic
(uint32 head/tail/size and the same chunked loops).  Pinned by the reference's exact known-answer test
src/test/utest/util/ringbuffer.cpp:30-192 (replayed in tests/test_oracle_delay.py); Delay has no direct reference
test and is pinned by the delay-line identity y[n] = x[n - d].
"""
import numpy as np

DELAY_GAP = 0x200
F = np.float32


class Delay:
    def __init__(self, max_size):
        self.size = ((max_size + DELAY_GAP + DELAY_GAP - 1) // DELAY_GAP) * DELAY_GAP
        self.buf = np.zeros(self.size, np.float32)
        self.head = self.tail = self.delay = 0

    def set_delay(self, delay):
        delay %= self.size
        self.delay = delay
        self.tail = (self.head + self.size - delay) % self.size

    def _push(self, src):
        n = len(src)
        end = self.head + n
        if end > self.size:
            cut = self.size - self.head
            self.buf[self.head:] = src[:cut]
            self.buf[:end - self.size] = src[cut:]
        else:
            self.buf[self.head:end] = src

    def append(self, src):
        src = np.asarray(src, np.float32)
        n = len(src)
        if n < self.size:
            self._push(src)
            self.head = (self.head + n) % self.size
        else:
            self.buf[:] = src[n - self.size:]
            self.head = 0
        self.tail = (self.head + self.size - self.delay) % self.size

    def process(self, src, gain=None, add_to=None, in_place=False):
        """Delay::process / process_add with optional scalar or vector gain (Delay.cpp:104-397).
        in_place: dst == src.  Without a delay the reference then appends the block as a whole and scales it
        (:107-111, 155-160, 204-209, 254-259, 303-308, 352-357), which restarts the line at cell 0 for a block of at least
        the line's length (:95-99)."""
        src = np.asarray(src, np.float32)
        if in_place and self.delay == 0:
            self.append(src)
            v = src if gain is None else (src * (gain if np.ndim(gain) else F(gain))).astype(np.float32)
            return v.copy() if add_to is None else (src + v).astype(np.float32)
        dst = np.empty_like(src) if add_to is None else np.array(add_to, np.float32, copy=True)
        gap = self.size - self.delay
        pos, count = 0, len(src)
        while count > 0:
            n = min(count, gap)
            self._push(src[pos:pos + n])
            self.head = (self.head + n) % self.size
            idx = (self.tail + np.arange(n)) % self.size
            v = self.buf[idx]
            if gain is not None:
                g = gain[pos:pos + n] if np.ndim(gain) else F(gain)
                v = (v * g).astype(np.float32)
            if add_to is None:
                dst[pos:pos + n] = v
            else:
                dst[pos:pos + n] = (dst[pos:pos + n] + v).astype(np.float32)
            self.tail = (self.tail + n) % self.size
            pos += n; count -= n
        return dst

    def process_ramping(self, src, delay, gain=None):
        """Delay.cpp:399-546."""
        src = np.asarray(src, np.float32)
        count = len(src)
        if delay == self.delay:
            return self.process(src, gain)
        if count == 0:
            return np.empty(0, np.float32)
        gap = self.size - max(delay, self.delay)
        delta = F(F(1.0) + F(F(self.delay - delay) / F(count)))
        old_tail = self.tail
        dst = np.empty_like(src)
        offset = 0
        while offset < count:
            n = min(count - offset, gap)
            self._push(src[offset:offset + n])
            for i in range(n):
                # size_t + ssize_t is an unsigned 64-bit sum (Delay.cpp:434): a negative step wraps modulo 2^64 first
                t = ((old_tail + int(F(delta * F(offset)))) % (1 << 64)) % self.size
                v = self.buf[t]
                if gain is not None:
                    v = F(v * (gain[offset] if np.ndim(gain) else F(gain)))
                dst[offset] = v
                offset += 1
            self.head = (self.head + n) % self.size
        self.tail = (self.head + self.size - delay) % self.size
        self.delay = delay
        return dst


class RingBuffer:
    def __init__(self, size, fill=0.0):
        self.cap = size
        self.head = 0
        self.data = np.full(size, fill, np.float32)

    def append(self, data):
        data = np.atleast_1d(np.asarray(data, np.float32))
        count = len(data)
        if count > self.cap:
            self.head = 0
            self.data[:] = data[count - self.cap:]
            return self.cap
        if self.head + count > self.cap:
            p1 = self.cap - self.head
            self.data[self.head:] = data[:p1]
            self.data[:count - p1] = data[p1:]
            self.head = count - p1
        else:
            self.data[self.head:self.head + count] = data
            self.head += count
        return count

    def append_one(self, v):
        self.data[self.head] = v
        self.head = (self.head + 1) % self.cap

    def get_one(self, offset):
        if offset >= self.cap:
            return F(0.0)
        return self.data[(self.head + self.cap - offset - 1) % self.cap]

    def tail_position(self, offset):
        return (self.head + self.cap - offset - 1) % self.cap if offset < self.cap else self.head

    def get(self, offset, count):
        """Returns (dst, to_read) like RingBuffer::get(dst, offset, count)."""
        dst = np.full(count, np.nan, np.float32)
        pos = 0
        if offset >= self.cap:
            lead = min(count, offset - self.cap + 1)
            dst[:lead] = 0.0
            offset -= lead
            if offset >= self.cap:
                return dst, 0
            count -= lead; pos = lead
        tail = (self.head + self.cap - offset - 1) % self.cap
        to_read = min(count, offset + 1)
        idx = (tail + np.arange(to_read)) % self.cap
        dst[pos:pos + to_read] = self.data[idx]
        if count > to_read:
            dst[pos + to_read:pos + count] = 0.0
        return dst, to_read
