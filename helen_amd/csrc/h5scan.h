// h5scan.h -- a read-only HDF5 *scanner* for MarginPolish image files (and this package's prediction files).
//
// The reference reader opens six tiny datasets per window (dataloader_predict.py:64-70); libhdf5 spends ~300 us on
// that the first (and only) time a window is read -- object opens, B-tree and heap look-ups through its metadata
// cache -- which caps a reader at ~3 k windows/s against 81 k windows/s of device throughput.  This walks the same
// on-disk structures directly in a read-only mapping of the file (HDF5 File Format Specification, version 3):
//
//   superblock      version 0 / 1 (libhdf5 defaults up to 1.8 "earliest") and 2 / 3 (libver = latest, SWMR)
//   object headers  version 1 (with continuation chunks) and version 2 ("OHDR" / "OCHK", any chunk-size width,
//                   creation-order fields skipped; checksums are not verified)
//   groups          old style (symbol-table message: local heap + version 1 B-tree + symbol table nodes) and new style
//                   (link-info message: link messages in the header, or "dense" storage = fractal heap + version 2
//                   B-tree of name hashes); hard links only
//   layouts         message version 3 and 4: compact, contiguous, chunked -- chunk index = version 1 B-tree (v3), or
//                   single chunk / implicit / fixed array without paging (v4)
//   filters         deflate (zlib), shuffle, fletcher32 (stripped, not verified), in any order; pipeline message v1 / v2
//   types           fixed point of 1/2/4/8 bytes, IEEE float / double, fixed- and variable-length strings; little endian
//
// -- i.e. what the HDF5 C library and h5py write with default settings, with `chunks=` / `compression="gzip"` /
// `shuffle=True`, or with `libver="latest"`.  ANYTHING else (extensible-array or B-tree-2 chunk indexes, i.e. unlimited
// dimensions; other filters; soft / external links; big-endian or committed datatypes; a user block; a damaged file) makes
// `open` / `children` / `lookup` / `dataset` return false, and the caller falls back to libhdf5 for that file: the fast
// path is an optimisation, never a second source of truth.  All methods are const and may be called from several threads.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace h5scan {

// zlib streams of deflated chunks through libdeflate when the system has it (loaded at run time, no build dependency:
// its inflate is two to three times zlib's, and a deflated image is 170 us of zlib per window -- the readers of a rank,
// not its MI355X, set the pace of a run on compressed MarginPolish output); zlib itself otherwise, and whenever
// libdeflate refuses a stream, so that what a damaged file does is unchanged.
struct FastInflate {
    void* (*alloc)() = nullptr;
    int (*zlib_decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*release)(void*) = nullptr;
    bool ok = false;
    FastInflate() {
        void* lib = nullptr;
        for (const char* name : {"libdeflate.so.0", "libdeflate.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return;
        alloc = (void* (*)())dlsym(lib, "libdeflate_alloc_decompressor");
        zlib_decompress = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(lib, "libdeflate_zlib_decompress");
        release = (void (*)(void*))dlsym(lib, "libdeflate_free_decompressor");
        ok = alloc && zlib_decompress && release;
    }
};
inline const FastInflate& fast_inflate() {
    static const FastInflate f;
    return f;
}
struct ThreadInflater {          // (a libdeflate decompressor serves one thread at a time)
    void* d = nullptr;
    ~ThreadInflater() {
        if (d) fast_inflate().release(d);
    }
};
// -> true when `full` bytes came out of the zlib stream [src, src + n)
inline bool inflate_exactly(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t full) {
    const FastInflate& f = fast_inflate();
    if (f.ok) {
        thread_local ThreadInflater t;
        if (!t.d) t.d = f.alloc();
        size_t got = 0;
        if (t.d && f.zlib_decompress(t.d, src, (size_t)n, dst, (size_t)full, &got) == 0 && got == full) return true;
    }
    uLongf got = (uLongf)full;
    return uncompress(dst, &got, src, (uLong)n) == Z_OK && got == full;
}

struct Dataset {
    int cls = -1;          // 0 fixed point, 1 float, 3 string, 9 variable-length string
    int size = 0;          // bytes per element (string: length of the fixed string)
    bool is_signed = false;
    int rank = 0;
    uint64_t dims[4] = {0, 0, 0, 0};
    const uint8_t* data = nullptr;   // raw element bytes: in the mapping (contiguous / compact) or in `owned` (chunked)
    uint64_t bytes = 0;
    std::shared_ptr<std::vector<uint8_t>> owned;   // chunks put together (and unfiltered) by File::dataset
    bool decoded = false;  // true: `owned` holds the data (the storage was chunked)
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
        if (ver <= 1) {
            if (map_[13] != 8 || map_[14] != 8) return false;         // 8-byte offsets and lengths only
            size_t o = 16 + 4 + 4;                                    // K values, consistency flags
            if (ver == 1) o += 4;                                     // indexed storage K + reserved
            if (u64(o) != 0) return false;                            // base address
            if (u64(o + 16) > size_) return false;                    // end-of-file address: a truncated file
            o += 32;                                                  // base, free space, end of file, driver info
            root_header_ = u64(o + 8);                                // root group symbol table entry
        } else if (ver <= 3) {
            if (map_[9] != 8 || map_[10] != 8) return false;          // sizes of offsets / lengths
            if (u64(12) != 0) return false;                           // base address
            if (u64(28) > size_) return false;                        // end-of-file address
            root_header_ = u64(36);
        } else {
            return false;
        }
        return ok(root_header_, 16);
    }

    uint64_t root() const { return root_header_; }
    size_t mapped_bytes() const { return map_ ? size_ : 0; }

    // members of a group in name order (bytewise, as strcmp: what H5Literate by name / h5py's keys() yield); false if
    // the object is not a group this scanner takes
    bool children(uint64_t header, std::vector<std::pair<std::string, uint64_t>>* out) const {
        GroupInfo g;
        if (!group_info(header, &g)) return false;
        if (g.old_style) {
            const Names names = heap_data(g.heap);
            uint64_t budget = size_ / 8 + 16;    // no well-formed file has more nodes than that: loops end here
            return names.p && walk(g.btree, names, out, 0, &budget);
        }
        if (g.dense) {
            if (!dense_links(g, out)) return false;
        } else {
            *out = g.compact;
        }
        std::sort(out->begin(), out->end());
        return true;
    }
    // one member by name; false if absent or not a group this scanner takes
    bool lookup(uint64_t header, const char* name, uint64_t* child) const {
        GroupInfo g;
        if (!group_info(header, &g)) return false;
        if (!g.old_style) {
            if (!g.dense) {
                for (auto& kv : g.compact)
                    if (kv.first == name) {
                        *child = kv.second;
                        return true;
                    }
                return false;
            }
            // dense storage: listed once per group, then served from the table
            std::shared_ptr<std::map<std::string, uint64_t>> table;
            {
                std::lock_guard<std::mutex> lock(dense_mutex_);
                auto it = dense_cache_.find(header);
                if (it != dense_cache_.end()) table = it->second;
            }
            if (!table) {
                std::vector<std::pair<std::string, uint64_t>> all;
                if (!dense_links(g, &all)) return false;
                table.reset(new std::map<std::string, uint64_t>(all.begin(), all.end()));
                std::lock_guard<std::mutex> lock(dense_mutex_);
                if (dense_cache_.size() > 64) dense_cache_.clear();
                dense_cache_[header] = table;
            }
            auto it = table->find(name);
            if (it == table->end()) return false;
            *child = it->second;
            return true;
        }
        uint64_t node = g.btree;
        const Names names = heap_data(g.heap);
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

    // type, shape and raw data of a dataset; false if it is not stored in a way this scanner takes.  Chunked storage is
    // put together (filters undone) into d->owned.
    bool dataset(uint64_t header, Dataset* d) const {
        bool have_space = false, have_type = false, have_layout = false, bad = false;
        Chunked ck;
        Pipeline pipe;
        auto visit = [&](uint16_t type, uint8_t flags, const uint8_t* m, size_t len) {
            if (flags & 0x02) {                       // shared message (committed datatype, ...)
                if (type == 0x0001 || type == 0x0003 || type == 0x0008 || type == 0x000B) bad = true;
                return;
            }
            switch (type) {
                case 0x0001: {                        // dataspace
                    if (len < 4) { bad = true; break; }
                    const int ver = m[0], rank = m[1];
                    const size_t at = ver == 1 ? 8 : ver == 2 ? 4 : 0;
                    if (at == 0 || rank > 4 || len < at + 8 * (size_t)rank) { bad = true; break; }
                    if (ver == 2 && m[3] == 2) { bad = true; break; }   // null dataspace
                    d->rank = rank;
                    for (int i = 0; i < rank; ++i) memcpy(&d->dims[i], m + at + 8 * i, 8);
                    have_space = true;
                    break;
                }
                case 0x0003: {                        // datatype (versions 1-3 share these fields)
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
                    if (len < 2 || (m[0] != 3 && m[0] != 4)) { bad = true; break; }   // version 3 (1.6 .. 1.10 default), 4 (latest)
                    const int cls = m[1];
                    if (cls == 1 && len >= 18) {                          // contiguous
                        uint64_t addr, bytes;
                        memcpy(&addr, m + 2, 8);
                        memcpy(&bytes, m + 10, 8);
                        if (addr == ~0ull) { d->data = nullptr; d->bytes = 0; }       // never written
                        else if (!ok(addr, bytes)) bad = true;
                        else { d->data = map_ + addr; d->bytes = bytes; }
                    } else if (cls == 0 && len >= 4) {                    // compact
                        uint16_t bytes;
                        memcpy(&bytes, m + 2, 2);
                        if (len < 4 + (size_t)bytes) { bad = true; break; }
                        d->data = m + 4;
                        d->bytes = bytes;
                    } else if (cls == 2) {
                        if (!parse_chunked(m, len, &ck)) bad = true;
                    } else {
                        bad = true;                                       // virtual
                    }
                    have_layout = true;
                    break;
                }
                case 0x000B:                          // filter pipeline
                    if (!parse_pipeline(m, len, &pipe)) bad = true;
                    break;
                case 0x0007: bad = true; break;       // external data files
                default: break;
            }
        };
        if (!messages(header, visit) || bad || !have_space || !have_type || !have_layout) return false;
        if (d->size <= 0) return false;
        if (d->cls == 9 && d->size != 16) return false;       // {length u32, collection address u64, index u32}
        if (ck.present) {
            if (!read_chunked(ck, pipe, d)) return false;
        } else if (!pipe.filters.empty()) {
            return false;                                     // filters on anything but chunked storage: not HDF5
        }
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

    // how a dataset is stored, for reports (check_images): 0 compact, 1 contiguous, 2 chunked; number of filters
    bool storage(uint64_t header, int* layout_class, int* n_filters, bool* deflated) const {
        int cls = -1, nf = 0;
        bool bad = false, z = false;
        const bool good = messages(header, [&](uint16_t type, uint8_t, const uint8_t* m, size_t len) {
            if (type == 0x0008 && len >= 2) cls = m[1];
            if (type == 0x000B) {
                Pipeline p;
                if (parse_pipeline(m, len, &p)) {
                    nf = (int)p.filters.size();
                    for (int id : p.filters) z = z || id == 1;
                } else {
                    bad = true;
                }
            }
        });
        if (!good || bad || cls < 0) return false;
        *layout_class = cls;
        *n_filters = nf;
        *deflated = z;
        return true;
    }

   private:
    int fd_ = -1;
    const uint8_t* map_ = nullptr;
    size_t size_ = 0;
    uint64_t root_header_ = 0;
    mutable std::mutex dense_mutex_;
    mutable std::map<uint64_t, std::shared_ptr<std::map<std::string, uint64_t>>> dense_cache_;

    bool ok(uint64_t addr, uint64_t len) const { return addr != ~0ull && addr <= size_ && len <= size_ - addr; }
    uint16_t u16(uint64_t a) const { uint16_t v; memcpy(&v, map_ + a, 2); return v; }
    uint32_t u32(uint64_t a) const { uint32_t v; memcpy(&v, map_ + a, 4); return v; }
    uint64_t u64(uint64_t a) const { uint64_t v; memcpy(&v, map_ + a, 8); return v; }
    uint64_t uvar(uint64_t a, int n) const {     // little-endian unsigned of n <= 8 bytes
        uint64_t v = 0;
        memcpy(&v, map_ + a, (size_t)n);
        return v;
    }
    static int log2_floor(uint64_t v) {
        int r = -1;
        while (v) { v >>= 1; ++r; }
        return r;
    }
    static int enc_size(uint64_t limit) { return log2_floor(limit) / 8 + 1; }   // H5VM_limit_enc_size

    // ---- object header messages: version 1 (continuation chunks) and version 2 ("OHDR" / "OCHK") ----
    template <typename F>
    bool messages(uint64_t header, F&& visit) const {
        if (!ok(header, 16)) return false;
        if (memcmp(map_ + header, "OHDR", 4) == 0) return messages_v2(header, visit);
        if (map_[header] != 1) return false;
        const int total = u16(header + 2);
        const uint32_t first = u32(header + 8);
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
    template <typename F>
    bool messages_v2(uint64_t header, F&& visit) const {
        if (map_[header + 4] != 2) return false;
        const uint8_t hflags = map_[header + 5];
        uint64_t o = header + 6;
        if (hflags & 0x20) o += 16;                   // access, modification, change, birth times
        if (hflags & 0x10) o += 4;                    // max compact / min dense attributes
        const int width = 1 << (hflags & 3);
        if (!ok(o, (uint64_t)width)) return false;
        const uint64_t size0 = uvar(o, width);
        o += width;
        const size_t mh = 4 + ((hflags & 0x04) ? 2 : 0);          // message header: type, size, flags (+ creation order)
        std::vector<std::pair<uint64_t, uint64_t>> chunks{{o, size0}};   // message bytes of each chunk (checksum excluded)
        for (size_t c = 0; c < chunks.size() && c < 64; ++c) {
            uint64_t p = chunks[c].first;
            const uint64_t end = p + chunks[c].second;
            if (!ok(p, chunks[c].second + 4)) return false;
            while (p + mh <= end) {
                const uint8_t type = map_[p];
                const uint16_t len = u16(p + 1);
                const uint8_t flags = map_[p + 3];
                if (p + mh + len > end) return false;
                const uint8_t* m = map_ + p + mh;
                if (type == 0x10 && len >= 16) {                  // continuation: an "OCHK" block, signature and checksum
                    uint64_t at, bytes;                           // included in its length
                    memcpy(&at, m, 8);
                    memcpy(&bytes, m + 8, 8);
                    if (bytes < 8 || !ok(at, bytes) || memcmp(map_ + at, "OCHK", 4) != 0) return false;
                    chunks.push_back({at + 4, bytes - 8});
                } else if (type != 0) {
                    visit((uint16_t)type, flags, m, (size_t)len);
                }
                p += mh + len;
            }
        }
        return true;
    }

    // ---- groups ----
    struct GroupInfo {
        bool old_style = false, dense = false;
        uint64_t btree = 0, heap = 0;                                 // old style: version 1 B-tree, local heap
        uint64_t fheap = ~0ull, name_index = ~0ull;                   // dense: fractal heap, version 2 B-tree
        std::vector<std::pair<std::string, uint64_t>> compact;        // links stored in the object header
    };
    // a link message (also the objects of a dense group's fractal heap); false: not a hard link this scanner takes
    static bool parse_link(const uint8_t* m, size_t len, std::string* name, uint64_t* addr) {
        if (len < 4 || m[0] != 1) return false;
        const uint8_t f = m[1];
        size_t o = 2;
        int type = 0;
        if (f & 0x08) {
            if (o >= len) return false;
            type = m[o++];
        }
        if (f & 0x04) o += 8;                          // creation order
        if (f & 0x10) o += 1;                          // character set
        const int w = 1 << (f & 3);
        if (o + w > len) return false;
        uint64_t n = 0;
        memcpy(&n, m + o, (size_t)w);
        o += w;
        if (n > len - o || type != 0) return false;   // soft / external / user-defined links: libhdf5's business
        name->assign((const char*)m + o, (size_t)n);
        o += n;
        if (o + 8 > len) return false;
        memcpy(addr, m + o, 8);
        return true;
    }
    bool group_info(uint64_t header, GroupInfo* g) const {
        bool symtab = false, linfo = false, bad = false;
        const bool good = messages(header, [&](uint16_t type, uint8_t flags, const uint8_t* m, size_t len) {
            if (type == 0x0011 && len >= 16) {                        // symbol table
                memcpy(&g->btree, m, 8);
                memcpy(&g->heap, m + 8, 8);
                symtab = true;
            } else if (type == 0x0002) {                              // link info
                if (len < 18 || m[0] != 0) { bad = true; return; }
                size_t o = 2;
                if (m[1] & 0x01) o += 8;                              // maximum creation index
                if (len < o + 16) { bad = true; return; }
                memcpy(&g->fheap, m + o, 8);
                memcpy(&g->name_index, m + o + 8, 8);
                linfo = true;
            } else if (type == 0x0006) {                              // link
                if (flags & 0x02) { bad = true; return; }
                std::string name;
                uint64_t addr;
                if (!parse_link(m, len, &name, &addr)) { bad = true; return; }
                g->compact.emplace_back(std::move(name), addr);
            }
        });
        if (!good || bad) return false;
        if (symtab) {
            g->old_style = true;
            return true;
        }
        if (!linfo) return false;                                     // not a group
        g->dense = g->fheap != ~0ull;
        return true;
    }

    // ---- dense link storage: a fractal heap of link messages, indexed by a version 2 B-tree of name hashes ----
    struct FractalHeap {
        int id_len = 0, off_bytes = 0, len_bytes = 0;
        unsigned width = 0, max_heap_bits = 0, cur_rows = 0;
        uint64_t start_block = 0, max_direct = 0, max_managed = 0, root = 0;
        bool checksummed = false;
    };
    bool parse_heap(uint64_t a, FractalHeap* h) const {
        // "FRHP" version(1) id length(2) filter length(2) flags(1) max managed(4) next huge id(8) huge B-tree(8)
        // free space(8) free-space manager(8) managed space(8) allocated(8) iterator offset(8) managed objects(8)
        // huge size(8) huge count(8) tiny size(8) tiny count(8) table width(2) starting block size(8)
        // max direct block size(8) max heap size(2) starting rows(2) root address(8) current rows(2) ...
        if (!ok(a, 144) || memcmp(map_ + a, "FRHP", 4) != 0 || map_[a + 4] != 0) return false;
        h->id_len = u16(a + 5);
        if (u16(a + 7) != 0) return false;            // filtered heap
        h->checksummed = (map_[a + 9] & 0x02) != 0;
        h->max_managed = u32(a + 10);
        uint64_t o = a + 14 + 8 * 12;
        h->width = u16(o);
        h->start_block = u64(o + 2);
        h->max_direct = u64(o + 10);
        h->max_heap_bits = u16(o + 18);
        h->root = u64(o + 22);
        h->cur_rows = u16(o + 30);
        if (h->width == 0 || h->start_block == 0 || (h->start_block & (h->start_block - 1)) ||
            (h->max_direct & (h->max_direct - 1)) || h->max_direct < h->start_block || h->max_heap_bits > 64 ||
            h->max_heap_bits < 8 || h->cur_rows > h->max_heap_bits ||
            log2_floor(h->start_block) + (int)h->cur_rows > 62 || h->width > (1u << 20))
            return false;
        h->off_bytes = (int)(h->max_heap_bits + 7) / 8;
        // H5HFhdr.c: min(bytes of an offset inside the largest direct block, bytes that hold the largest managed object size)
        h->len_bytes = std::min((log2_floor(h->max_direct) + 7) / 8, enc_size(h->max_managed ? h->max_managed : 1));
        return h->id_len >= 1 + h->off_bytes + h->len_bytes;
    }
    uint64_t row_block_size(const FractalHeap& h, unsigned row) const {
        return row < 2 ? h.start_block : h.start_block << (row - 1);
    }
    // the bytes of a managed object at heap offset `off`
    bool heap_locate(const FractalHeap& h, uint64_t off, uint64_t len, const uint8_t** p) const {
        uint64_t block = h.root, base = 0, bsize = h.start_block;
        unsigned rows = h.cur_rows;
        const unsigned max_direct_rows = (unsigned)(log2_floor(h.max_direct) - log2_floor(h.start_block)) + 2;
        const uint64_t ihead = 4 + 1 + 8 + (uint64_t)h.off_bytes;
        for (int depth = 0; rows != 0; ++depth) {                     // descend indirect blocks
            if (depth > 8 || !ok(block, ihead) || memcmp(map_ + block, "FHIB", 4) != 0) return false;
            uint64_t row_base = base;
            bool found = false;
            for (unsigned r = 0; r < rows; ++r) {
                if (r >= 2 && (unsigned)log2_floor(h.start_block) + (r - 1) > 62) return false;       // the shift below
                const uint64_t bs = row_block_size(h, r);
                if (bs > (~0ull >> 1) / h.width) return false;                                          // bs * width would wrap
                const uint64_t span = bs * h.width;
                if (off < row_base + span) {
                    const uint64_t col = (off - row_base) / bs;
                    // entries: one address per block, direct rows first, indirect rows behind them (row-major)
                    const uint64_t entry = (uint64_t)r * h.width + col;
                    if (!ok(block + ihead, (entry + 1) * 8)) return false;
                    const uint64_t child = u64(block + ihead + entry * 8);
                    if (child == ~0ull) return false;
                    base = row_base + col * bs;
                    block = child;
                    if (r < max_direct_rows) {
                        rows = 0;
                        bsize = bs;
                    } else {                                          // an indirect block spanning `bs` bytes of heap space
                        rows = (unsigned)(log2_floor(bs) - log2_floor(h.start_block * h.width)) + 1;
                    }
                    found = true;
                    break;
                }
                row_base += span;
            }
            if (!found) return false;
        }
        // direct block: "FHDB" version(1) heap header address(8) block offset (off_bytes) [checksum(4)] data ...
        // heap offsets count from the block's first byte, its header included
        const uint64_t dhead = 4 + 1 + 8 + (uint64_t)h.off_bytes + (h.checksummed ? 4 : 0);
        if (!ok(block, bsize) || memcmp(map_ + block, "FHDB", 4) != 0) return false;
        if (off < base + dhead || off - base > bsize || len > bsize - (off - base)) return false;
        *p = map_ + block + (off - base);
        return true;
    }
    // every record of a version 2 B-tree (any order); false on anything unexpected
    bool btree2_records(uint64_t a, int want_type, std::vector<const uint8_t*>* recs, unsigned* rec_size) const {
        if (!ok(a, 38) || memcmp(map_ + a, "BTHD", 4) != 0 || map_[a + 4] != 0 || map_[a + 5] != want_type) return false;
        const uint32_t node_size = u32(a + 6);
        const unsigned rsize = u16(a + 10), depth = u16(a + 12);
        const uint64_t root = u64(a + 16);
        const unsigned root_n = u16(a + 24);
        const uint64_t total = u64(a + 26);
        if (rsize == 0 || node_size < 16 + rsize || depth > 8 || total > size_ / rsize) return false;
        *rec_size = rsize;
        // H5B2hdr.c: records per node and the widths of the child-pointer fields, level by level
        std::vector<uint64_t> max_nrec(depth + 1), cum_max(depth + 1);
        std::vector<int> cum_size(depth + 1);
        max_nrec[0] = (node_size - 10) / rsize;
        cum_max[0] = max_nrec[0];
        cum_size[0] = 0;
        const int nrec_size = enc_size(max_nrec[0]);
        for (unsigned u = 1; u <= depth; ++u) {
            const uint64_t ptr = 8 + (uint64_t)nrec_size + (uint64_t)cum_size[u - 1];
            if (node_size < 10 + ptr + rsize) return false;
            max_nrec[u] = (node_size - (10 + ptr)) / (rsize + ptr);
            cum_max[u] = (max_nrec[u] + 1) * cum_max[u - 1] + max_nrec[u];
            cum_size[u] = enc_size(cum_max[u]);
        }
        if (root == ~0ull) return total == 0;
        uint64_t budget = size_ / 16 + 16;
        struct Walk {
            const File* f;
            unsigned rsize;
            int nrec_size;
            const std::vector<int>* cum_size;
            std::vector<const uint8_t*>* recs;
            uint64_t* budget;
            uint64_t total;
            bool node(uint64_t at, uint64_t nrec, unsigned depth) {
                if (*budget == 0) return false;
                --*budget;
                // (child pointers that alias one node must not grow the list past what the header promises)
                if (nrec > total || recs->size() + nrec > total) return false;
                const uint64_t body = 6 + nrec * rsize;
                if (!f->ok(at, body)) return false;
                if (depth == 0) {
                    if (memcmp(f->map_ + at, "BTLF", 4) != 0) return false;
                    for (uint64_t i = 0; i < nrec; ++i) recs->push_back(f->map_ + at + 6 + i * rsize);
                    return true;
                }
                if (memcmp(f->map_ + at, "BTIN", 4) != 0) return false;
                for (uint64_t i = 0; i < nrec; ++i) recs->push_back(f->map_ + at + 6 + i * rsize);
                const uint64_t ptr = 8 + (uint64_t)nrec_size + (depth > 1 ? (uint64_t)(*cum_size)[depth - 1] : 0);
                if (!f->ok(at + body, (nrec + 1) * ptr)) return false;
                for (uint64_t i = 0; i <= nrec; ++i) {
                    const uint64_t p = at + body + i * ptr;
                    if (!node(f->u64(p), f->uvar(p + 8, nrec_size), depth - 1)) return false;
                }
                return true;
            }
        } w{this, rsize, nrec_size, &cum_size, recs, &budget, total};
        if (!w.node(root, root_n, depth)) return false;
        return recs->size() == total;
    }
    bool dense_links(const GroupInfo& g, std::vector<std::pair<std::string, uint64_t>>* out) const {
        FractalHeap h;
        if (!parse_heap(g.fheap, &h)) return false;
        std::vector<const uint8_t*> recs;
        unsigned rsize = 0;
        if (!btree2_records(g.name_index, 5, &recs, &rsize)) return false;     // type 5: link name hash (4) + heap id
        if (rsize != 4 + (unsigned)h.id_len) return false;
        out->reserve(out->size() + recs.size());
        for (const uint8_t* r : recs) {
            const uint8_t* id = r + 4;
            if ((id[0] & 0xC0) != 0 || (id[0] & 0x30) != 0) return false;     // version 0, managed objects only
            uint64_t off = 0, len = 0;
            memcpy(&off, id + 1, (size_t)h.off_bytes);
            memcpy(&len, id + 1 + h.off_bytes, (size_t)h.len_bytes);
            const uint8_t* p;
            if (!heap_locate(h, off, len, &p)) return false;
            std::string name;
            uint64_t addr;
            if (!parse_link(p, (size_t)len, &name, &addr)) return false;
            out->emplace_back(std::move(name), addr);
        }
        return true;
    }

    // ---- chunked storage ----
    struct Chunked {
        bool present = false;
        int version = 0, rank = 0, index = 0;         // index: 0 = version 1 B-tree (layout v3); v4: 1 single, 2 implicit, 3 fixed array
        uint64_t addr = ~0ull;
        uint64_t dims[5] = {0, 0, 0, 0, 0};           // chunk shape (elements) [+ element size for v3]
        bool single_filtered = false;
        uint64_t single_bytes = 0;
        uint32_t single_mask = 0;
    };
    struct Pipeline {
        std::vector<int> filters;                     // ids in the order they were applied when writing
    };
    bool parse_chunked(const uint8_t* m, size_t len, Chunked* c) const {
        c->present = true;
        c->version = m[0];
        if (m[0] == 3) {
            // version(1) class(1) dimensionality(1) = rank + 1, B-tree address(8), dimension sizes (4 each; last = element size)
            if (len < 11) return false;
            const int nd = m[2];
            if (nd < 2 || nd > 5 || len < 11 + 4 * (size_t)nd) return false;
            c->rank = nd - 1;
            memcpy(&c->addr, m + 3, 8);
            for (int i = 0; i < nd; ++i) {
                uint32_t v;
                memcpy(&v, m + 11 + 4 * i, 4);
                c->dims[i] = v;
            }
            c->index = 0;
            return true;
        }
        // version 4: version(1) class(1) flags(1) dimensionality(1) encoded length of a dimension(1) dimensions ...
        // chunk index type(1) type-specific information, address(8)
        if (len < 5) return false;
        const uint8_t flags = m[2];
        const int nd = m[3], enc = m[4];
        if (nd < 2 || nd > 5 || enc < 1 || enc > 8) return false;
        size_t o = 5;
        if (len < o + (size_t)nd * enc + 1) return false;
        c->rank = nd - 1;
        for (int i = 0; i < nd; ++i) {
            uint64_t v = 0;
            memcpy(&v, m + o, (size_t)enc);
            c->dims[i] = v;
            o += enc;
        }
        c->index = m[o++];
        if (c->index == 1) {                          // single chunk
            if (flags & 0x02) {                       // ... which is filtered: its stored size and filter mask
                if (len < o + 12) return false;
                memcpy(&c->single_bytes, m + o, 8);
                memcpy(&c->single_mask, m + o + 8, 4);
                c->single_filtered = true;
                o += 12;
            }
        } else if (c->index == 2) {                   // implicit: nothing
        } else if (c->index == 3) {                   // fixed array: page bits
            o += 1;
        } else {
            return false;                             // extensible array / version 2 B-tree: unlimited dimensions
        }
        if (len < o + 8) return false;
        memcpy(&c->addr, m + o, 8);
        return true;
    }
    static bool parse_pipeline(const uint8_t* m, size_t len, Pipeline* p) {
        if (len < 2) return false;
        const int ver = m[0], n = m[1];
        size_t o = ver == 1 ? 8 : ver == 2 ? 2 : 0;
        if (o == 0 || n > 32) return false;
        for (int i = 0; i < n; ++i) {
            uint16_t id, name_len = 0, ncd;
            if (o + 2 > len) return false;
            memcpy(&id, m + o, 2);
            o += 2;
            if (ver == 1 || id >= 256) {
                if (o + 2 > len) return false;
                memcpy(&name_len, m + o, 2);
                o += 2;
            }
            if (o + 4 > len) return false;
            o += 2;                                   // flags
            memcpy(&ncd, m + o, 2);
            o += 2;
            o += ver == 1 ? ((size_t)name_len + 7) / 8 * 8 : name_len;
            o += 4 * (size_t)ncd;
            if (ver == 1 && (ncd & 1)) o += 4;
            if (o > len) return false;
            if (id != 1 && id != 2 && id != 3) return false;        // deflate, shuffle, fletcher32
            p->filters.push_back(id);
        }
        return true;
    }
    // one stored chunk -> `full` bytes of elements
    bool unfilter(const uint8_t* raw, uint64_t nbytes, uint32_t mask, const Pipeline& pipe, uint64_t full, int elem,
                  std::vector<uint8_t>* a, std::vector<uint8_t>* b, const uint8_t** out) const {
        const uint8_t* cur = raw;
        uint64_t cur_n = nbytes;
        for (int i = (int)pipe.filters.size() - 1; i >= 0; --i) {
            if (mask & (1u << i)) continue;           // this filter was skipped for this chunk
            const int id = pipe.filters[i];
            if (id == 3) {                            // fletcher32: four checksum bytes behind the data
                if (cur_n < 4) return false;
                cur_n -= 4;
            } else if (id == 1) {                     // deflate
                std::vector<uint8_t>* dst = (cur == a->data()) ? b : a;
                // what follows decides the size: only fletcher32 / shuffle keep it, so the result is the full chunk
                dst->resize(full);
                if (!inflate_exactly(cur, cur_n, dst->data(), full)) return false;
                cur = dst->data();
                cur_n = full;
            } else {                                  // shuffle: byte k of every element stored together
                if (elem > 1) {
                    std::vector<uint8_t>* dst = (cur == a->data()) ? b : a;
                    dst->resize(cur_n);
                    const uint64_t n = cur_n / (uint64_t)elem;
                    for (int k = 0; k < elem; ++k) {
                        const uint8_t* s = cur + (uint64_t)k * n;
                        uint8_t* t = dst->data() + k;
                        for (uint64_t e = 0; e < n; ++e) t[e * (uint64_t)elem] = s[e];
                    }
                    memcpy(dst->data() + n * (uint64_t)elem, cur + n * (uint64_t)elem, cur_n - n * (uint64_t)elem);
                    cur = dst->data();
                }
            }
        }
        if (cur_n != full) return false;
        *out = cur;
        return true;
    }
    bool read_chunked(const Chunked& c, const Pipeline& pipe, Dataset* d) const {
        if (c.rank != d->rank || d->rank < 1 || d->rank > 2) return false;       // images, positions and labels: rank 1 / 2
        const uint64_t n = d->count();
        const int elem = d->size;
        if (n == ~0ull || n > ((uint64_t)1 << 28) / (uint64_t)elem) return false;   // 256 MiB: nothing on this path is larger
        if (c.dims[c.rank] != (uint64_t)elem) return false;     // the last "dimension" of a chunk is the element size
        uint64_t cdim[2] = {c.dims[0], d->rank == 2 ? c.dims[1] : 1};
        const uint64_t ddim[2] = {d->dims[0], d->rank == 2 ? d->dims[1] : 1};
        if (cdim[0] == 0 || cdim[1] == 0) return false;
        if (cdim[0] > ((uint64_t)1 << 28) || cdim[1] > ((uint64_t)1 << 28) || cdim[0] * cdim[1] > ((uint64_t)1 << 28) / (uint64_t)elem)
            return false;
        const uint64_t chunk_bytes = cdim[0] * cdim[1] * (uint64_t)elem;
        d->owned.reset(new std::vector<uint8_t>(n * (uint64_t)elem, 0));
        d->decoded = true;
        d->data = d->owned->data();
        d->bytes = d->owned->size();
        if (n == 0) return true;
        if (c.addr == ~0ull) return false;            // never written: fill values are libhdf5's business
        const uint64_t nchunks[2] = {(ddim[0] + cdim[0] - 1) / cdim[0], (ddim[1] + cdim[1] - 1) / cdim[1]};
        std::vector<uint8_t> a, b;
        uint64_t placed = 0;
        auto place = [&](uint64_t o0, uint64_t o1, uint64_t addr, uint64_t nbytes, uint32_t mask) -> bool {
            if (o0 >= ddim[0] || o1 >= ddim[1] || o0 % cdim[0] || o1 % cdim[1] || !ok(addr, nbytes)) return false;
            const uint8_t* src;
            if (pipe.filters.empty()) {
                if (nbytes != chunk_bytes) return false;
                src = map_ + addr;
            } else if (!unfilter(map_ + addr, nbytes, mask, pipe, chunk_bytes, elem, &a, &b, &src)) {
                return false;
            }
            const uint64_t rows = std::min(cdim[0], ddim[0] - o0), cols = std::min(cdim[1], ddim[1] - o1);
            uint8_t* dst = d->owned->data();
            if (cols == ddim[1] && cdim[1] == ddim[1]) {
                memcpy(dst + o0 * ddim[1] * elem, src, rows * cols * elem);
            } else {
                for (uint64_t r = 0; r < rows; ++r)
                    memcpy(dst + ((o0 + r) * ddim[1] + o1) * elem, src + r * cdim[1] * elem, cols * elem);
            }
            ++placed;
            return true;
        };
        if (c.index == 0) {                           // version 1 B-tree, node type 1 (raw data chunks)
            uint64_t budget = size_ / 32 + 16;
            if (!chunk_btree(c.addr, c.rank, place, 0, &budget)) return false;
        } else if (c.index == 1) {                    // single chunk
            if (nchunks[0] * nchunks[1] != 1) return false;
            const uint64_t nb = c.single_filtered ? c.single_bytes : chunk_bytes;
            if (!c.single_filtered && !pipe.filters.empty()) return false;
            if (!place(0, 0, c.addr, nb, c.single_mask)) return false;
        } else if (c.index == 2) {                    // implicit: all chunks one after the other, unfiltered
            if (!pipe.filters.empty()) return false;
            uint64_t at = c.addr;
            for (uint64_t i = 0; i < nchunks[0]; ++i)
                for (uint64_t j = 0; j < nchunks[1]; ++j) {
                    if (!place(i * cdim[0], j * cdim[1], at, chunk_bytes, 0)) return false;
                    at += chunk_bytes;
                }
        } else {                                      // fixed array
            // "FAHD" version(1) client(1) entry size(1) page bits(1) entries(8) data block address(8) checksum(4)
            if (!ok(c.addr, 28) || memcmp(map_ + c.addr, "FAHD", 4) != 0 || map_[c.addr + 4] != 0) return false;
            const int client = map_[c.addr + 5], esize = map_[c.addr + 6], page_bits = map_[c.addr + 7];
            const uint64_t entries = u64(c.addr + 8), db = u64(c.addr + 16);
            if (entries != nchunks[0] * nchunks[1] || page_bits > 32 || entries > ((uint64_t)1 << page_bits)) return false;   // paged: not taken
            const bool filtered = client == 1;
            if (filtered == pipe.filters.empty()) return false;
            if (esize < 8 || (filtered && (esize < 13 || esize > 20))) return false;
            // "FADB" version(1) client(1) header address(8) elements ... checksum(4)
            if (db == ~0ull || !ok(db, 14 + entries * (uint64_t)esize) || memcmp(map_ + db, "FADB", 4) != 0) return false;
            for (uint64_t k = 0; k < entries; ++k) {
                const uint64_t e = db + 14 + k * (uint64_t)esize;
                const uint64_t addr = u64(e);
                uint64_t nb = chunk_bytes;
                uint32_t mask = 0;
                if (filtered) {
                    nb = uvar(e + 8, esize - 12);
                    mask = u32(e + 8 + (uint64_t)(esize - 12));
                }
                if (addr == ~0ull) return false;
                if (!place((k / nchunks[1]) * cdim[0], (k % nchunks[1]) * cdim[1], addr, nb, mask)) return false;
            }
        }
        return placed == nchunks[0] * nchunks[1];     // a chunk that was never written: fill values are libhdf5's business
    }
    template <typename P>
    bool chunk_btree(uint64_t node, int rank, P&& place, int depth, uint64_t* budget) const {
        // "TREE" type(1) = 1, level(1), entries(2), left(8), right(8), then {key: bytes(4) mask(4) offsets(8 x (rank + 1)); child(8)} ...
        if (depth > 16 || *budget == 0 || !ok(node, 24) || memcmp(map_ + node, "TREE", 4) != 0 || map_[node + 4] != 1) return false;
        --*budget;
        const int level = map_[node + 5], n = u16(node + 6);
        const uint64_t key = 8 + 8 * (uint64_t)(rank + 1);
        if (!ok(node + 24, (uint64_t)n * (key + 8) + key)) return false;
        for (int k = 0; k < n; ++k) {
            const uint64_t e = node + 24 + (uint64_t)k * (key + 8);
            const uint64_t child = u64(e + key);
            if (level > 0) {
                if (!chunk_btree(child, rank, place, depth + 1, budget)) return false;
            } else {
                const uint64_t o0 = u64(e + 8), o1 = rank == 2 ? u64(e + 16) : 0;
                if (!place(o0, o1, child, u32(e), u32(e + 4))) return false;
            }
        }
        return true;
    }

    // ---- old-style groups ----
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
