// h5emit.h -- a minimal HDF5 *emitter* for the prediction file of `helen call_consensus`.
//
// libhdf5 spends ~100-120 us creating the group and three small datasets of one window
// (DataStore.py:123-133) whatever the caller does -- 8 k windows/s per process against 81 k windows/s of
// device throughput -- which is why round 1 sharded the writer over processes and files.  The layout of
// a prediction file is fixed and append-only, so this writes the bytes itself: the datasets' raw data
// and object headers as the windows arrive (one sequential stream through a large buffer), the group
// structures at close, when every name is known.  Only the oldest, checksum-free structures of the
// HDF5 File Format Specification (version 1.x: superblock version 0, version 1 object headers, "old
// style" groups = local heap + version 1 B-tree + symbol table nodes, contiguous / compact layouts) are
// produced; any libhdf5 >= 1.6 and h5py read them, and so does helen's stitch.
//
//   file     := superblock | objects ... | group structures ...
//   dataset  := raw data (8-aligned) + object header {dataspace v1, datatype v1 fixed-point,
//               fill value v2, layout v3 contiguous (compact for the scalars)}
//   group    := local heap (names) + symbol table nodes (8 sorted entries each) + B-tree nodes
//               (32 children each, key = heap offset of the largest name of the child to its left)
//               + object header {symbol table message}
#pragma once
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace h5emit {

constexpr uint64_t kUndef = ~0ull;
constexpr int kLeafK = 4;       // symbol table node: up to 2 * kLeafK symbols   (libhdf5's defaults,
constexpr int kInternalK = 16;  // B-tree node: up to 2 * kInternalK children     recorded in the superblock)
constexpr size_t kSnodBytes = 8 + 2 * kLeafK * 40;
constexpr size_t kTreeBytes = 24 + 2 * kInternalK * 8 + (2 * kInternalK + 1) * 8;

struct Child {
    std::string name;
    uint64_t header;   // address of the child's object header
};

// The file buffer: bytes appended at the end, WITHOUT the value-initialisation a std::vector insists on (a window's 14 KB
// are written exactly once, by whoever fills them) and with storage that stays put until it is handed to a flusher.
class Bytes {
   public:
    Bytes() = default;
    Bytes(const Bytes&) = delete;
    Bytes& operator=(const Bytes&) = delete;
    Bytes(Bytes&& o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
    Bytes& operator=(Bytes&& o) noexcept {
        if (this != &o) {
            free(p_);
            p_ = o.p_; n_ = o.n_; cap_ = o.cap_;
            o.p_ = nullptr; o.n_ = o.cap_ = 0;
        }
        return *this;
    }
    ~Bytes() { free(p_); }
    uint8_t* data() { return p_; }
    const uint8_t* data() const { return p_; }
    size_t size() const { return n_; }
    size_t capacity() const { return cap_; }
    bool empty() const { return n_ == 0; }
    void clear() { n_ = 0; }
    void reserve(size_t c) {
        if (c > cap_) {
            p_ = (uint8_t*)realloc(p_, c);
            if (!p_) abort();
            cap_ = c;
        }
    }
    uint8_t* extend(size_t n) {                 // n more bytes, uninitialised
        if (n_ + n > cap_) reserve(std::max(n_ + n, cap_ + cap_ / 2));
        uint8_t* q = p_ + n_;
        n_ += n;
        return q;
    }
    void append(const void* b, size_t n) { memcpy(extend(n), b, n); }
    void zeros(size_t n) { memset(extend(n), 0, n); }
    void push_back(uint8_t v) { *extend(1) = v; }

   private:
    uint8_t* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

class File {
   public:
    ~File() {
        stop_flushers();
        if (fd_ >= 0) ::close(fd_);
    }
    bool open(const char* path) {
        fd_ = ::open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
        if (fd_ < 0) return false;
        for (int t = 0; t < kFlushers; ++t) flushers_.emplace_back([this]() { flusher_loop(); });
        buf_.reserve(kFlush + (1 << 20));
        // room for the superblock (96 bytes), written last
        buf_.clear();
        buf_.zeros(kDataStart);
        base_ = 0;
        return true;
    }
    bool ok() const { return fd_ >= 0 && !failed_.load(); }

    // ---- datasets ----
    // integer dataset of `rank` dims with contiguous layout; returns the object header address
    // `fill(dst)` writes the `bytes` of the dataset straight into the file buffer (a conversion need not pass through
    // a buffer of its own); data == nullptr selects it
    template <class Fill>
    uint64_t dataset_filled(Fill&& fill, size_t bytes, int elem_size, bool is_signed, int rank, const uint64_t* dims) {
        align8();
        const uint64_t addr = tell();
        fill(buf_.extend(bytes));
        if (buf_.size() >= kFlush) flush();
        return dataset_header(addr, bytes, elem_size, is_signed, rank, dims);
    }
    uint64_t dataset(const void* data, size_t bytes, int elem_size, bool is_signed, int rank, const uint64_t* dims) {
        align8();
        const uint64_t addr = tell();
        last_data_ = addr;
        put(data, bytes);
        return dataset_header(addr, bytes, elem_size, is_signed, rank, dims);
    }
    uint64_t dataset_header(uint64_t addr, size_t bytes, int elem_size, bool is_signed, int rank, const uint64_t* dims) {
        align8();
        const uint64_t hdr = tell();
        const size_t msgs = (8 + 8 + 8 * rank) + (8 + 16) + (8 + 8) + (8 + 24);
        header_prefix(4, msgs);
        msg_dataspace(rank, dims);
        msg_datatype(elem_size, is_signed);
        msg_fill(/*alloc late*/ 2);
        msg_head(0x0008, 24);   // layout v3, contiguous
        put8(3); put8(1); put64(addr); put64(bytes); pad(6);
        return hdr;
    }
    // one-element array of a fixed-length string (how the image files store `contig`), compact layout
    uint64_t string1(const std::string& v) {
        align8();
        const uint64_t hdr = tell();
        const size_t len = v.size() ? v.size() : 1, data = (len + 7) & ~(size_t)7;
        const size_t lay = (4 + data + 7) & ~(size_t)7;
        const uint64_t dims[1] = {1};
        header_prefix(4, (8 + 16) + (8 + 8) + (8 + 8) + (8 + lay));
        msg_dataspace(1, dims);
        msg_head(0x0003, 8);    // datatype: class 3 (string), version 1, null-terminated ASCII, `len` bytes
        put8(0x13); put8(0); put8(0); put8(0); put32((uint32_t)len);
        msg_fill(1);
        msg_head(0x0008, (uint16_t)lay);
        put8(3); put8(0); put16((uint16_t)len);
        put(v.data(), v.size());
        pad(lay - 4 - v.size());
        return hdr;
    }
    // scalar int64 dataset with compact layout (the value lives in the object header)
    uint64_t scalar_i64(int64_t v) {
        align8();
        const uint64_t hdr = tell();
        const size_t msgs = (8 + 8) + (8 + 16) + (8 + 8) + (8 + 16);
        header_prefix(4, msgs);
        msg_dataspace(0, nullptr);
        msg_datatype(8, true);
        msg_fill(/*alloc early: required for compact*/ 1);
        msg_head(0x0008, 16);   // layout v3, compact: size u16 + data
        put8(3); put8(0); put16(8); put64((uint64_t)v); pad(4);
        return hdr;
    }

    // ---- stamps ----
    // A block of objects whose bytes depend on where it lands only through the absolute addresses inside it (a window's three
    // datasets and their group: fixed names, fixed shapes) is emitted ONCE, into memory, at two different addresses; what
    // differs between the two copies are the address fields.  Every later use copies the block and adds its address to those
    // fields (the writer's time per window went into some 200 small appends and a dozen allocations before; now one copy, a
    // handful of patches and the window's data).  `marks` = offsets the builder asked to remember (where data goes),
    // `result` = offset of the object header the block is referred to by.
    struct Stamp {
        std::vector<uint8_t> bytes;
        std::vector<uint32_t> patches;
        std::vector<uint64_t> marks;      // (offset, bytes) pairs of the data areas
        std::vector<uint64_t> literal;    // (offset, bytes) pairs of everything else
        uint64_t result = 0;
    };
    // emit(File&, std::vector<uint64_t>& marks) -> address of the block's object header; the block must start 8-aligned
    template <class Emit>
    static Stamp make_stamp(Emit&& emit) {
        Stamp s[2];
        const uint64_t at[2] = {kDataStart, kDataStart + (1ull << 20)};
        for (int i = 0; i < 2; ++i) {
            File f;
            f.memory_ = true;
            f.base_ = at[i];
            s[i].result = emit(f, s[i].marks) - at[i];
            for (auto& m : s[i].marks) m -= at[i];
            s[i].bytes.assign(f.buf_.data(), f.buf_.data() + f.buf_.size());
        }
        const size_t n = s[0].bytes.size();
        for (size_t p = 0; p < n;) {
            if (s[0].bytes[p] == s[1].bytes[p]) { ++p; continue; }
            // the lowest byte that a difference of 1 << 20 changes is byte 2 of the little-endian field
            if (p < 2 || p + 6 > n) { s[0].bytes.clear(); return std::move(s[0]); }                   // (not a stampable block)
            const size_t o = p - 2;
            uint64_t v0, v1;
            memcpy(&v0, s[0].bytes.data() + o, 8);
            memcpy(&v1, s[1].bytes.data() + o, 8);
            if (v1 - v0 != (1ull << 20)) { s[0].bytes.clear(); return std::move(s[0]); }
            v0 -= at[0];
            memcpy(s[0].bytes.data() + o, &v0, 8);
            s[0].patches.push_back((uint32_t)o);
            p = o + 8;
        }
        uint64_t at_byte = 0;
        for (size_t k = 0; k + 1 < s[0].marks.size(); k += 2) {
            if (s[0].marks[k] > at_byte) { s[0].literal.push_back(at_byte); s[0].literal.push_back(s[0].marks[k] - at_byte); }
            at_byte = s[0].marks[k] + s[0].marks[k + 1];
        }
        if (n > at_byte) { s[0].literal.push_back(at_byte); s[0].literal.push_back(n - at_byte); }
        return std::move(s[0]);
    }
    // Appends the block (8-aligned); -> its bytes in the file buffer, for the caller to fill the marked data areas, and the
    // address of its object header.  stamped() afterwards hands a full buffer on.
    uint8_t* stamp(const Stamp& st, uint64_t* header) {
        uint64_t addr;
        uint8_t* p = reserve_block(st, &addr, header);
        fill_block(st, p, addr);
        return p;
    }
    void stamped() {
        if (buf_.size() >= kFlush) flush();
    }
    // The two halves of stamp(), for a caller that fills its blocks LATER and on other threads: room for the block now (the
    // addresses are final), its bytes whenever -- before the buffer is handed on, which before_flush announces.  The
    // buffer's storage does not move in between (its reserve covers everything appended between two flush checks).
    uint8_t* reserve_block(const Stamp& st, uint64_t* addr, uint64_t* header) {
        align8();
        *addr = tell();
        *header = *addr + st.result;
        return buf_.extend(st.bytes.size());
    }
    // (everything outside the marked data areas: `literal` = [offset, length) pairs; thread-safe: touches only the block)
    static void fill_block(const Stamp& st, uint8_t* p, uint64_t addr) {
        for (size_t k = 0; k + 1 < st.literal.size(); k += 2) memcpy(p + st.literal[k], st.bytes.data() + st.literal[k], st.literal[k + 1]);
        for (uint32_t o : st.patches) {
            uint64_t v;
            memcpy(&v, p + o, 8);
            v += addr;
            memcpy(p + o, &v, 8);
        }
    }
    void set_before_flush(std::function<void()> f) { before_flush_ = std::move(f); }
    bool full() const { return buf_.size() >= kFlush; }
    uint64_t last_data_address() const { return last_data_; }   // where the newest dataset()'s raw data begins

    // ---- groups ----
    // old-style group over `kids` (any order; sorted here by name, bytewise like strcmp); returns the object
    // header address, and the B-tree / heap addresses for the root entry of the superblock
    uint64_t group(std::vector<Child>& kids, uint64_t* btree_out = nullptr, uint64_t* heap_out = nullptr) {
        const auto by_name = [](const Child& a, const Child& b) { return a.name < b.name; };
        if (!std::is_sorted(kids.begin(), kids.end(), by_name)) std::sort(kids.begin(), kids.end(), by_name);
        return group_sorted(kids.size(), [&](size_t i) -> const std::string& { return kids[i].name; },
                            [&](size_t i) { return kids[i].header; }, btree_out, heap_out);
    }
    // the same over `n` members ALREADY in name order, named and addressed through the two accessors (a contig's 300,000
    // regions are written from the writer's own table without copying a name)
    template <typename NameAt, typename HeaderAt>
    uint64_t group_sorted(size_t n_kids, NameAt&& name_at, HeaderAt&& header_at, uint64_t* btree_out = nullptr,
                          uint64_t* heap_out = nullptr) {
        // local heap: "" at offset 0, then the names, each NUL-terminated and padded to 8
        std::vector<uint64_t> off(n_kids);
        size_t heap_bytes = 8;
        for (size_t i = 0; i < n_kids; ++i) {
            off[i] = heap_bytes;
            heap_bytes += (name_at(i).size() + 1 + 7) & ~(size_t)7;
        }
        std::vector<uint8_t> heap(heap_bytes, 0);
        for (size_t i = 0; i < n_kids; ++i) memcpy(heap.data() + off[i], name_at(i).data(), name_at(i).size());
        align8();
        const uint64_t heap_addr = tell();
        put("HEAP", 4); put8(0); pad(3);
        put64(heap.size());        // data segment size
        put64(1);                  // free list head: H5HL_FREE_NULL (no free block)
        put64(heap_addr + 32);     // data segment address: right behind this prefix
        put(heap.data(), heap.size());
        // symbol table nodes
        struct Node { uint64_t addr, max_key; };
        std::vector<Node> level;
        for (size_t i = 0; i < n_kids; i += 2 * kLeafK) {
            const size_t n = std::min<size_t>(2 * kLeafK, n_kids - i);
            const uint64_t a = tell();
            uint8_t node[kSnodBytes];
            memset(node, 0, sizeof(node));
            memcpy(node, "SNOD", 4);
            node[4] = 1;
            const uint16_t n16 = (uint16_t)n;
            memcpy(node + 6, &n16, 2);
            for (size_t k = 0; k < n; ++k) {
                memcpy(node + 8 + 40 * k, &off[i + k], 8);
                const uint64_t h = header_at(i + k);
                memcpy(node + 8 + 40 * k + 8, &h, 8);
            }
            put(node, sizeof(node));
            level.push_back({a, off[i + n - 1]});
        }
        // B-tree levels, bottom up; an empty group is one empty leaf-level node
        int depth = 0;
        uint64_t root;
        for (;;) {
            std::vector<Node> up;
            const size_t per = 2 * kInternalK;
            const size_t nodes = std::max<size_t>(1, (level.size() + per - 1) / per);
            const uint64_t first = tell();
            for (size_t j = 0; j < nodes; ++j) {
                const size_t lo = j * per, n = level.empty() ? 0 : std::min(per, level.size() - lo);
                const uint64_t a = first + j * kTreeBytes;
                put("TREE", 4); put8(0); put8((uint8_t)depth); put16((uint16_t)n);
                put64(j ? a - kTreeBytes : kUndef);
                put64(j + 1 < nodes ? a + kTreeBytes : kUndef);
                put64(lo ? level[lo - 1].max_key : 0);   // left key: the largest name of everything to the left ("" at the far left)
                for (size_t k = 0; k < n; ++k) {
                    put64(level[lo + k].addr);
                    put64(level[lo + k].max_key);
                }
                pad((per - n) * 16);
                up.push_back({a, n ? level[lo + n - 1].max_key : 0});
            }
            level.swap(up);
            ++depth;
            if (level.size() == 1) {
                root = level[0].addr;
                break;
            }
        }
        const uint64_t hdr = tell();
        header_prefix(1, 8 + 16);
        msg_head(0x0011, 16);   // symbol table message
        put64(root); put64(heap_addr);
        if (btree_out) *btree_out = root;
        if (heap_out) *heap_out = heap_addr;
        return hdr;
    }

    // superblock with `root` (from group()) as the root group, flush, close
    bool finish(uint64_t root_header, uint64_t root_btree, uint64_t root_heap) {
        align8();
        const uint64_t eof = tell();
        flush();
        stop_flushers();             // every piece is in the file (or failed_ says otherwise)
        uint8_t sb[96];
        size_t o = 0;
        auto w8 = [&](uint8_t v) { sb[o++] = v; };
        auto w16 = [&](uint16_t v) { memcpy(sb + o, &v, 2); o += 2; };
        auto w32 = [&](uint32_t v) { memcpy(sb + o, &v, 4); o += 4; };
        auto w64 = [&](uint64_t v) { memcpy(sb + o, &v, 8); o += 8; };
        const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
        memcpy(sb, sig, 8); o = 8;
        w8(0); w8(0); w8(0); w8(0); w8(0); w8(8); w8(8); w8(0);
        w16(kLeafK); w16(kInternalK); w32(0);
        w64(0); w64(kUndef); w64(eof); w64(kUndef);
        w64(0); w64(root_header); w32(1); w32(0); w64(root_btree); w64(root_heap);
        bool good = !failed_.load() && ::pwrite(fd_, sb, sizeof(sb), 0) == (ssize_t)sizeof(sb);
        good = (::close(fd_) == 0) && good;
        fd_ = -1;
        return good;
    }

   private:
    static constexpr size_t kFlush = 8 << 20;
    static constexpr size_t kDataStart = 2048;
    // A full buffer goes to the file BEHIND the caller: the kernel's copy into the page cache (or a RAM-backed file's
    // pages) is as long as the formatting of the next buffer, and two pieces of a file can be copied at once.  Pieces
    // are written with pwrite at their own offsets, so their order does not matter; at most kInFlight wait or are
    // being written (the caller blocks beyond that); finish() writes the superblock after the last piece is in.
    static constexpr int kFlushers = 2;
    static constexpr size_t kInFlight = 3;
    int fd_ = -1;
    bool memory_ = false;        // make_stamp's scratch files: everything stays in buf_
    std::atomic<bool> failed_{false};
    Bytes buf_;
    std::function<void()> before_flush_;   // blocks reserved but not yet filled: fill them now
    uint64_t base_ = 0;   // file offset of buf_[0]
    uint64_t last_data_ = 0;
    struct Piece {
        Bytes bytes;
        uint64_t offset;
    };
    std::vector<std::thread> flushers_;
    std::mutex mutex_;
    std::condition_variable work_, room_;
    std::deque<Piece> queue_;
    std::vector<Bytes> spare_;
    size_t in_flight_ = 0;
    bool stopping_ = false;

    uint64_t tell() const { return base_ + buf_.size(); }
    void flusher_loop() {
        for (;;) {
            Piece piece;
            {
                std::unique_lock<std::mutex> lock(mutex_);
                work_.wait(lock, [this]() { return stopping_ || !queue_.empty(); });
                if (queue_.empty()) return;
                piece = std::move(queue_.front());
                queue_.pop_front();
            }
            size_t done = 0;
            while (done < piece.bytes.size() && !failed_.load()) {
                const ssize_t n = ::pwrite(fd_, piece.bytes.data() + done, piece.bytes.size() - done,
                                           (off_t)(piece.offset + done));
                if (n <= 0) failed_.store(true);
                else done += (size_t)n;
            }
            {
                std::lock_guard<std::mutex> lock(mutex_);
                piece.bytes.clear();
                spare_.push_back(std::move(piece.bytes));
                --in_flight_;
            }
            room_.notify_all();
        }
    }
    void stop_flushers() {
        {
            std::unique_lock<std::mutex> lock(mutex_);
            room_.wait(lock, [this]() { return in_flight_ == 0; });
            stopping_ = true;
        }
        work_.notify_all();
        for (auto& t : flushers_) t.join();
        flushers_.clear();
        stopping_ = false;
    }
    void flush() {
        if (buf_.empty() || memory_) return;
        if (before_flush_) before_flush_();
        if (flushers_.empty()) {            // (after finish() has stopped them: nothing writes then)
            size_t done = 0;
            while (done < buf_.size() && !failed_.load()) {
                const ssize_t n = ::pwrite(fd_, buf_.data() + done, buf_.size() - done, (off_t)(base_ + done));
                if (n <= 0) failed_.store(true);
                else done += (size_t)n;
            }
            base_ += buf_.size();
            buf_.clear();
            return;
        }
        Bytes next;
        {
            std::unique_lock<std::mutex> lock(mutex_);
            room_.wait(lock, [this]() { return in_flight_ < kInFlight; });
            if (!spare_.empty()) {
                next = std::move(spare_.back());
                spare_.pop_back();
            }
            ++in_flight_;
            queue_.push_back(Piece{std::move(buf_), base_});
            base_ += queue_.back().bytes.size();
        }
        work_.notify_one();
        buf_ = std::move(next);
        buf_.clear();
        buf_.reserve(kFlush + (1 << 20));
    }
    void put(const void* p, size_t n) {
        buf_.append(p, n);
        if (buf_.size() >= kFlush) flush();
    }
    void pad(size_t n) { buf_.zeros(n); }
    void align8() { pad((8 - (tell() & 7)) & 7); }
    void put8(uint8_t v) { buf_.push_back(v); }
    void put16(uint16_t v) { put(&v, 2); }
    void put32(uint32_t v) { put(&v, 4); }
    void put64(uint64_t v) { put(&v, 8); }

    void header_prefix(int nmsgs, size_t msg_bytes) {   // version 1 object header prefix, 16 bytes
        put8(1); put8(0); put16((uint16_t)nmsgs); put32(1); put32((uint32_t)msg_bytes); pad(4);
    }
    void msg_head(uint16_t type, uint16_t size) { put16(type); put16(size); put8(0); pad(3); }
    void msg_dataspace(int rank, const uint64_t* dims) {
        msg_head(0x0001, (uint16_t)(8 + 8 * rank));
        put8(1); put8((uint8_t)rank); put8(0); put8(0); put32(0);
        for (int i = 0; i < rank; ++i) put64(dims[i]);
    }
    void msg_datatype(int size, bool is_signed) {   // class 0 (fixed point), version 1, little endian
        msg_head(0x0003, 16);
        put8(0x10); put8(is_signed ? 0x08 : 0x00); put8(0); put8(0);
        put32((uint32_t)size); put16(0); put16((uint16_t)(8 * size)); pad(4);
    }
    void msg_fill(int alloc_time) {   // version 2: allocation time, write time "if set", default value
        msg_head(0x0005, 8);
        put8(2); put8((uint8_t)alloc_time); put8(2); put8(1); put32(0);
    }
};

}  // namespace h5emit
