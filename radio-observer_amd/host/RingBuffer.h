// RingBuffer.h -- RingBuffer2D<T> as the reference's recorders see it
// (src/RingBuffer.h:210-621): push()/mark()/at()/size()/reserve()/isDirty() with the same
// index arithmetic, including its quirks (size(start) == capacity when start == head,
// Reservation::init storing end = 0).  Storage is one contiguous block (the reference's
// chunking only matters for its allocator); the chunk size still rounds the capacity.
#pragma once

#include <cassert>
#include <vector>

namespace ro {

template <class T> class RingBuffer2D {
public:
    RingBuffer2D() {}
    RingBuffer2D(int width, int chunkSize) { setGeometry(width, chunkSize); }
    RingBuffer2D(int width, int chunkSize, int capacity) { resize(width, chunkSize, capacity); }

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
        data_.assign((size_t)capacity_ * (size_t)width_, T());
        clear();
    }
    void resize(int capacity) { resize(width_, chunkSize_, capacity); }

    // :482-496: hands out the head row, advances, marks overlapped reservations dirty
    T *push()
    {
        T *row = &data_[(size_t)head_ * width_];
        head_ = (head_ + 1) % capacity_;
        if (!isFull()) size_++;
        for (auto &r : reservations_)
            if (r.alive && isInRange(head_, r.start, r.end)) r.dirty = true;
        return row;
    }
    T  *at(int mark) { return &data_[(size_t)normalizeRowIndex(mark) * width_]; }    // :498-503
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

    void setGeometry(int width, int chunkSize)
    {
        width_ = width;
        chunkSize_ = chunkSize;
        const int rowSize = (int)sizeof(T) * width_;
        chunkRows_ = chunkSize_ / rowSize;
        if (chunkSize_ % rowSize != 0) chunkRows_++;
    }

    int width_ = 0, chunkSize_ = 0, chunkRows_ = 0, capacity_ = 0, head_ = 0, size_ = 0;
    std::vector<T> data_;
    std::vector<Reservation> reservations_;
    std::vector<int> freeReservations_;
};

}  // namespace ro
