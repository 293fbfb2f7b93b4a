// RingBuffer.h -- RingBuffer2D<T> as the reference's recorders see it
// (src/RingBuffer.h:210-621): push()/mark()/at()/size()/reserve()/isDirty() with the same
// index arithmetic, including its quirks (size(start) == capacity when start == head,
// Reservation::init storing end = 0).  Storage is one contiguous block (the reference's
// chunking only matters for its allocator); the chunk size still rounds the capacity.
#pragma once

#include <cassert>
#include <cstddef>
#include <vector>

namespace ro {

template <class T> class RingBuffer2D {
public:
    RingBuffer2D() {}
    RingBuffer2D(int width, int chunkSize) { setGeometry(width, chunkSize); }
    RingBuffer2D(int width, int chunkSize, int capacity) { resize(width, chunkSize, capacity); }
    ~RingBuffer2D() { release(); }
    RingBuffer2D(const RingBuffer2D &) = delete;
    RingBuffer2D &operator=(const RingBuffer2D &) = delete;

    // Where the rows live.  By default the heap; a backend whose rows arrive by DMA hands in an allocator of page-locked
    // memory (ro_pinned_alloc / ro_pinned_free of the C ABI) BEFORE resize().  alloc may fail (nullptr): the heap again.
    typedef void *(*AllocFn)(void *ctx, size_t bytes);
    typedef void (*FreeFn)(void *ctx, void *p);
    void setStorage(AllocFn alloc, FreeFn free_fn, void *ctx)
    {
        release();
        alloc_ = alloc;
        free_ = free_fn;
        ctx_ = ctx;
    }
    bool storageIsCustom() const { return custom_ != nullptr; }

    int  getWidth() const { return width_; }
    int  getCapacity() const { return capacity_; }
    int  getSize() const { return size_; }
    int  getChunkRows() const { return chunkRows_; }
    bool isEmpty() const { return size_ == 0; }
    bool isFull() const { return size_ >= capacity_ && capacity_ > 0; }          // :412-415

    void clear() { head_ = 0; size_ = 0; }

    // src/RingBuffer.h:428-457: chunkRows = ceil(chunkSize / rowBytes); capacity rounded up to chunks
    void resize(int width, int chunkSize, int capacity)
    {
        setGeometry(width, chunkSize);
        int chunks = capacity / chunkRows_;
        if (capacity % chunkRows_ > 0) chunks += 1;
        capacity_ = chunks * chunkRows_;
        release();
        const size_t n = (size_t)capacity_ * (size_t)width_;
        if (alloc_ && n > 0) custom_ = static_cast<T *>(alloc_(ctx_, n * sizeof(T)));
        if (custom_) {
            for (size_t i = 0; i < n; ++i) custom_[i] = T();
            base_ = custom_;
        } else {
            heap_.assign(n, T());
            base_ = heap_.data();
        }
        clear();
    }
    void resize(int capacity) { resize(width_, chunkSize_, capacity); }

    // :482-496: hands out the head row, advances, marks overlapped reservations dirty
    T *push()
    {
        T *row = &base_[(size_t)head_ * width_];
        head_ = (head_ + 1) % capacity_;
        if (!isFull()) size_++;
        for (auto &r : reservations_)
            if (r.alive && isInRange(head_, r.start, r.end)) r.dirty = true;
        return row;
    }
    // n pushes at once: the same bookkeeping as n calls of push() (a reservation is dirty as soon as ANY of the heads
    // the ring passes through falls inside it), the rows handed out as at most two runs of consecutive slots --
    // fill(first row of the run, its row count, rows handed out before it).  What a frontend's whole process() call
    // of samples costs per call instead of per sample.
    template <class F> void pushRun(int n, F fill)
    {
        int done = 0;
        while (done < n) {
            const int run = n - done < capacity_ - head_ ? n - done : capacity_ - head_;
            fill(&base_[(size_t)head_ * width_], run, done);
            markHeads((head_ + 1) % capacity_, run);
            head_ = (head_ + run) % capacity_;
            size_ = size_ + run < capacity_ ? size_ + run : capacity_;
            done += run;
        }
    }
    // advance the head over n rows somebody else has already written in place (a DMA into data()): push()'s bookkeeping
    void pushWritten(int n) { pushRun(n, [](T *, int, int) {}); }
    // ... and what has to happen BEFORE that somebody writes: the n slots ahead of the head are about to be overwritten,
    // so every reservation the next n pushes would mark dirty is dirty now (its reader must not trust those rows)
    void markAhead(int n)
    {
        if (n > capacity_) n = capacity_;
        if (n > 0) markHeads((head_ + 1) % capacity_, n);
    }
    T  *data() { return base_; }
    T  *at(int mark) { return &base_[(size_t)normalizeRowIndex(mark) * width_]; }    // :498-503
    int mark() const { return head_; }                                               // :505-509

    int normalizeRowIndex(int value) const                                           // :360-369
    {
        while (value < 0) value += capacity_;
        return value % capacity_;
    }
    int size(int start, int end) const                                               // :543-551
    {
        start = normalizeRowIndex(start);
        end = normalizeRowIndex(end);
        if (end > start) return end - start;
        return (capacity_ - start) + end;
    }
    int  size(int start) const { return size(start, head_); }                        // :555-560
    bool isInRange(int index, int start, int end) const                              // :562-573
    {
        index = normalizeRowIndex(index);
        start = normalizeRowIndex(start);
        end = normalizeRowIndex(end);
        if (end > start) return index >= start && index < end;
        return index >= start || index < end;
    }
    int reserve(int start, int end)                                                  // :583-601
    {
        start = normalizeRowIndex(start);
        (void)normalizeRowIndex(end);
        int handle;
        if (!freeReservations_.empty()) {
            handle = freeReservations_.back();
            freeReservations_.pop_back();
        } else {
            handle = (int)reservations_.size();
            reservations_.push_back(Reservation());
        }
        reservations_[handle] = Reservation{start, 0, true, false};                  // end = 0: :524-529
        return handle;
    }
    bool freeReservation(int handle)                                                 // :610-616
    {
        if (handle < 0 || handle >= (int)reservations_.size()) return false;
        reservations_[handle].alive = false;
        freeReservations_.push_back(handle);
        return true;
    }
    bool isDirty(int handle) const                                                   // :617-620
    {
        assert(handle >= 0 && handle < (int)reservations_.size());
        return reservations_[handle].dirty;
    }

private:
    struct Reservation { int start = 0, end = 0; bool alive = false, dirty = false; };

    // push()'s dirty rule for the `run` heads first, first + 1, ... (mod capacity) at once: a reservation is dirty when
    // one of them lies in [start, end) as isInRange reads it (end <= start wraps; end == start is the whole ring)
    void markHeads(int first, int run)
    {
        for (auto &r : reservations_) {
            if (!r.alive || r.dirty) continue;
            const int s0 = normalizeRowIndex(r.start), e0 = normalizeRowIndex(r.end);
            const int len = e0 > s0 ? e0 - s0 : capacity_ - s0 + e0;
            const int d = s0 >= first ? s0 - first : s0 + capacity_ - first;      // steps from `first` up to start
            if (d < run || d + len > capacity_) r.dirty = true;                  // the heads reach start, or begin inside
        }
    }

    void setGeometry(int width, int chunkSize)
    {
        width_ = width;
        chunkSize_ = chunkSize;
        const int rowSize = (int)sizeof(T) * width_;
        chunkRows_ = chunkSize_ / rowSize;
        if (chunkSize_ % rowSize != 0) chunkRows_++;
    }

    void release()
    {
        if (custom_ && free_) free_(ctx_, custom_);
        custom_ = nullptr;
        base_ = nullptr;
        heap_.clear();
        heap_.shrink_to_fit();
    }

    int width_ = 0, chunkSize_ = 0, chunkRows_ = 0, capacity_ = 0, head_ = 0, size_ = 0;
    T *base_ = nullptr;                  // = custom_ or heap_.data()
    T *custom_ = nullptr;
    std::vector<T> heap_;
    AllocFn alloc_ = nullptr;
    FreeFn free_ = nullptr;
    void *ctx_ = nullptr;
    std::vector<Reservation> reservations_;
    std::vector<int> freeReservations_;
};

}  // namespace ro
