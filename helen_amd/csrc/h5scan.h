// h5scan.h -- a minimal read-only HDF5 *scanner* for MarginPolish image files.
//
// The reference reader opens six tiny datasets per window (dataloader_predict.py:64-70); libhdf5 spends ~300 us on
// that the first (and only) time a window is read -- object opens, B-tree and heap look-ups through its metadata
// cache -- which caps a reader process at ~3 k windows/s against 81 k windows/s of device throughput.  This walks
// the same on-disk structures directly in a read-only mapping of the file: superblock version 0/1, version 1
// object headers (with continuation chunks), "old style" groups (local heap + version 1 B-tree + symbol table
// nodes), contiguous and compact layouts, fixed-point / IEEE float / fixed- and variable-length string types --
// what the HDF5 C library writes with default settings, which is what MarginPolish uses.  ANYTHING else (newer
// superblocks, version 2 object headers, link-message groups, chunked or filtered datasets, big-endian or shared
// datatypes) makes `open` / `dataset` return false, and the caller falls back to libhdf5 for that file: the fast
// path is an optimisation, never a second source of truth.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace h5scan {

struct Dataset {
    int cls = -1;          // 0 fixed point, 1 float, 3 string, 9 variable-length string
    int size = 0;          // bytes per element (string: length of the fixed string)
    bool is_signed = false;
    int rank = 0;
    uint64_t dims[4] = {0, 0, 0, 0};
    const uint8_t* data = nullptr;   // in the mapping: raw data (contiguous or compact)
    uint64_t bytes = 0;
    uint64_t count() const {   // saturates: a damaged dataspace must not wrap around to a small count
        unsigned __int128 n = 1;
        for (int i = 0; i < rank; ++i) {
            n *= dims[i];
            if (n > ~0ull) return ~0ull;
        }
        return (uint64_t)n;
    }
    bool holds(uint64_t per_element) const {   // bytes >= count() * per_element, without overflow
        return (unsigned __int128)count() * per_element <= bytes;
    }
};

class File {
   public:
    ~File() { close(); }
    void close() {
        if (map_) munmap((void*)map_, size_);
        map_ = nullptr;
        if (fd_ >= 0) ::close(fd_);
        fd_ = -1;
    }
    // false: not a file this scanner takes (the caller uses libhdf5)
    bool open(const char* path) {
        fd_ = ::open(path, O_RDONLY);
        if (fd_ < 0) return false;
        struct stat st;
        if (fstat(fd_, &st) != 0 || st.st_size < 96) return false;
        size_ = (size_t)st.st_size;
        void* p = mmap(nullptr, size_, PROT_READ, MAP_SHARED, fd_, 0);
        if (p == MAP_FAILED) return false;
        map_ = (const uint8_t*)p;
        static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
        if (memcmp(map_, sig, 8) != 0) return false;                 // (a user block would move it: not handled)
        const uint8_t ver = map_[8];
        if (ver > 1) return false;                                    // superblock 2 / 3: new-style files
        if (map_[13] != 8 || map_[14] != 8) return false;             // 8-byte offsets and lengths only
        size_t o = 16 + 4 + 4;                                        // K values, consistency flags
        if (ver == 1) o += 4;                                         // indexed storage K + reserved
        if (u64(o) != 0) return false;                                // base address
        if (u64(o + 16) > size_) return false;                        // end-of-file address: a truncated file
        o += 32;                                                      // base, free space, end of file, driver info
        // root group symbol table entry
        root_header_ = u64(o + 8);
        return ok(root_header_, 16);
    }

    uint64_t root() const { return root_header_; }
    size_t mapped_bytes() const { return map_ ? size_ : 0; }

    // members of an old-style group in name order; false if the object is not such a group
    bool children(uint64_t header, std::vector<std::pair<std::string, uint64_t>>* out) const {
        uint64_t btree = 0, heap = 0;
        if (!symbol_table(header, &btree, &heap)) return false;
        const Names names = heap_data(heap);
        uint64_t budget = size_ / 8 + 16;    // no well-formed file has more nodes than that: loops end here
        return names.p && walk(btree, names, out, 0, &budget);
    }
    // one member by name (descends the B-tree by key comparison); false if absent or not an old-style group
    bool lookup(uint64_t header, const char* name, uint64_t* child) const {
        uint64_t node = 0, heap = 0;
        if (!symbol_table(header, &node, &heap)) return false;
        const Names names = heap_data(heap);
        if (!names.p) return false;
        for (int guard = 0; guard < 64; ++guard) {
            if (!ok(node, 24)) return false;
            const uint8_t* p = map_ + node;
            if (memcmp(p, "TREE", 4) == 0) {
                if (p[4] != 0) return false;
                const int n = u16(node + 6);
                if (!ok(node, 24 + 8 + (size_t)n * 16)) return false;
                // child k holds names in (key[k], key[k+1]]
                int k = 0;
                for (; k < n; ++k) {
                    const char* key = names.at(u64(node + 24 + 16 * (size_t)(k + 1)));
                    if (!key) return false;
                    if (strcmp(name, key) <= 0) break;
                }
                if (k == n) return false;
                node = u64(node + 24 + 8 + 16 * (size_t)k);
            } else if (memcmp(p, "SNOD", 4) == 0) {
                const int n = u16(node + 6);
                if (!ok(node, 8 + (size_t)n * 40)) return false;
                for (int k = 0; k < n; ++k) {
                    const uint64_t e = node + 8 + 40 * (size_t)k;
                    const char* member = names.at(u64(e));
                    if (!member) return false;
                    if (strcmp(name, member) == 0) {
                        *child = u64(e + 8);
                        return true;
                    }
                }
                return false;
            } else {
                return false;
            }
        }
        return false;
    }

    // type, shape and raw data of a dataset; false if it is not stored in a way this scanner takes
    bool dataset(uint64_t header, Dataset* d) const {
        bool have_space = false, have_type = false, have_layout = false, bad = false;
        auto visit = [&](uint16_t type, uint8_t flags, const uint8_t* m, size_t len) {
            if (flags & 0x02) {                       // shared message (committed datatype, ...)
                if (type == 0x0001 || type == 0x0003 || type == 0x0008) bad = true;
                return;
            }
            switch (type) {
                case 0x0001: {                        // dataspace
                    if (len < 8) { bad = true; break; }
                    const int ver = m[0], rank = m[1];
                    const size_t at = ver == 1 ? 8 : ver == 2 ? 4 : 0;
                    if (at == 0 || rank > 4 || len < at + 8 * (size_t)rank) { bad = true; break; }
                    if (ver == 2 && m[3] == 2) { bad = true; break; }   // null dataspace
                    d->rank = rank;
                    for (int i = 0; i < rank; ++i) memcpy(&d->dims[i], m + at + 8 * i, 8);
                    have_space = true;
                    break;
                }
                case 0x0003: {                        // datatype
                    if (len < 8) { bad = true; break; }
                    const int cls = m[0] & 0x0F;
                    uint32_t size;
                    memcpy(&size, m + 4, 4);
                    d->cls = cls;
                    d->size = (int)size;
                    if (cls == 0) {
                        if (m[1] & 0x01) bad = true;                      // big-endian
                        d->is_signed = (m[1] & 0x08) != 0;
                        uint16_t off, prec;
                        if (len < 12) { bad = true; break; }
                        memcpy(&off, m + 8, 2);
                        memcpy(&prec, m + 10, 2);
                        if (off != 0 || prec != 8 * size || (size != 1 && size != 2 && size != 4 && size != 8)) bad = true;
                    } else if (cls == 1) {
                        if ((m[1] & 0x01) || (size != 4 && size != 8)) bad = true;   // IEEE little-endian only
                    } else if (cls == 3) {
                        /* fixed-length string of `size` bytes */
                    } else if (cls == 9) {
                        if ((m[1] & 0x0F) != 1) bad = true;               // variable-length STRING only
                    } else {
                        bad = true;
                    }
                    have_type = true;
                    break;
                }
                case 0x0008: {                        // layout
                    if (len < 2 || m[0] != 3) { bad = true; break; }      // version 3 (libhdf5 1.6 .. 1.10 default)
                    if (m[1] == 1 && len >= 18) {                         // contiguous
                        uint64_t addr, bytes;
                        memcpy(&addr, m + 2, 8);
                        memcpy(&bytes, m + 10, 8);
                        if (addr == ~0ull) { d->data = nullptr; d->bytes = 0; }       // never written
                        else if (!ok(addr, bytes)) bad = true;
                        else { d->data = map_ + addr; d->bytes = bytes; }
                    } else if (m[1] == 0 && len >= 4) {                   // compact
                        uint16_t bytes;
                        memcpy(&bytes, m + 2, 2);
                        if (len < 4 + (size_t)bytes) { bad = true; break; }
                        d->data = m + 4;
                        d->bytes = bytes;
                    } else {
                        bad = true;                                       // chunked: libhdf5's business
                    }
                    have_layout = true;
                    break;
                }
                case 0x000B: bad = true; break;       // filter pipeline
                case 0x0007: bad = true; break;       // external data files
                default: break;
            }
        };
        if (!messages(header, visit) || bad || !have_space || !have_type || !have_layout) return false;
        if (d->size <= 0) return false;
        if (d->cls == 9 && d->size != 16) return false;       // {length u32, collection address u64, index u32}
        if (d->data && !d->holds((uint64_t)d->size)) return false;
        return true;
    }

    // element 0 of a string dataset (fixed or variable length) -> out; false if it cannot be had
    bool first_string(const Dataset& d, std::string* out) const {
        if (!d.data || d.count() < 1) return false;
        if (d.cls == 3) {
            out->assign((const char*)d.data, strnlen((const char*)d.data, (size_t)d.size));
            return true;
        }
        if (d.cls != 9) return false;
        uint32_t len, index;
        uint64_t col;
        memcpy(&len, d.data, 4);
        memcpy(&col, d.data + 4, 8);
        memcpy(&index, d.data + 12, 4);
        if (len == 0 && col == 0) { out->clear(); return true; }
        if (!ok(col, 16) || memcmp(map_ + col, "GCOL", 4) != 0) return false;
        const uint64_t csize = u64(col + 8);
        if (!ok(col, csize)) return false;
        uint64_t o = col + 16;
        while (o + 16 <= col + csize) {                   // heap objects: index u16, refcount u16, reserved u32, size u64
            const uint16_t idx = u16(o);
            const uint64_t sz = u64(o + 8);
            if (idx == 0) break;                          // the free-space object ends the list
            if (sz > col + csize - (o + 16)) return false;
            if (idx == index) {
                const size_t n = len <= sz ? len : (size_t)sz;
                out->assign((const char*)map_ + o + 16, strnlen((const char*)map_ + o + 16, n));
                return true;
            }
            o += 16 + ((sz + 7) & ~7ull);
        }
        return false;
    }

   private:
    int fd_ = -1;
    const uint8_t* map_ = nullptr;
    size_t size_ = 0;
    uint64_t root_header_ = 0;

    bool ok(uint64_t addr, uint64_t len) const { return addr != ~0ull && addr <= size_ && len <= size_ - addr; }
    uint16_t u16(uint64_t a) const { uint16_t v; memcpy(&v, map_ + a, 2); return v; }
    uint64_t u64(uint64_t a) const { uint64_t v; memcpy(&v, map_ + a, 8); return v; }

    // every message of a version 1 object header, continuation chunks included
    template <typename F>
    bool messages(uint64_t header, F&& visit) const {
        if (!ok(header, 16) || map_[header] != 1) return false;      // version 2 headers start with "OHDR"
        const int total = u16(header + 2);
        uint32_t first;
        memcpy(&first, map_ + header + 8, 4);
        std::vector<std::pair<uint64_t, uint64_t>> chunks{{header + 16, first}};
        int seen = 0;
        for (size_t c = 0; c < chunks.size() && c < 64; ++c) {
            uint64_t o = chunks[c].first;
            const uint64_t end = o + chunks[c].second;
            if (!ok(o, chunks[c].second)) return false;
            while (o + 8 <= end && seen < total) {
                const uint16_t type = u16(o), len = u16(o + 2);
                const uint8_t flags = map_[o + 4];
                if (o + 8 + len > end) return false;
                if (type == 0x0010 && len >= 16) chunks.push_back({u64(o + 8), u64(o + 16)});   // continuation
                else visit(type, flags, map_ + o + 8, (size_t)len);
                o += 8 + len;
                ++seen;
            }
        }
        return true;
    }
    bool symbol_table(uint64_t header, uint64_t* btree, uint64_t* heap) const {
        bool found = false;
        const bool good = messages(header, [&](uint16_t type, uint8_t, const uint8_t* m, size_t len) {
            if (type == 0x0011 && len >= 16) {
                memcpy(btree, m, 8);
                memcpy(heap, m + 8, 8);
                found = true;
            }
        });
        return good && found;
    }
    struct Names {              // a local heap's data segment; at(off) = NUL-terminated name or nullptr
        const uint8_t* p = nullptr;
        uint64_t bytes = 0;
        const char* at(uint64_t off) const {
            if (!p || off >= bytes || !memchr(p + off, 0, bytes - off)) return nullptr;
            return (const char*)p + off;
        }
    };
    Names heap_data(uint64_t heap) const {
        Names n;
        if (!ok(heap, 32) || memcmp(map_ + heap, "HEAP", 4) != 0) return n;
        const uint64_t bytes = u64(heap + 8), addr = u64(heap + 24);
        if (ok(addr, bytes)) {
            n.p = map_ + addr;
            n.bytes = bytes;
        }
        return n;
    }
    bool walk(uint64_t node, const Names& names, std::vector<std::pair<std::string, uint64_t>>* out, int depth,
              uint64_t* budget) const {
        if (depth > 16 || !ok(node, 24) || *budget == 0) return false;
        --*budget;
        const uint8_t* p = map_ + node;
        if (memcmp(p, "SNOD", 4) == 0) {
            const int n = u16(node + 6);
            if (!ok(node, 8 + (size_t)n * 40)) return false;
            for (int k = 0; k < n; ++k) {
                const uint64_t e = node + 8 + 40 * (size_t)k;
                const char* member = names.at(u64(e));
                if (!member) return false;
                out->emplace_back(member, u64(e + 8));
            }
            return true;
        }
        if (memcmp(p, "TREE", 4) != 0 || p[4] != 0) return false;
        const int n = u16(node + 6);
        if (!ok(node, 24 + 8 + (size_t)n * 16)) return false;
        for (int k = 0; k < n; ++k)
            if (!walk(u64(node + 24 + 8 + 16 * (size_t)k), names, out, depth + 1, budget)) return false;
        return true;
    }
};

}  // namespace h5scan
