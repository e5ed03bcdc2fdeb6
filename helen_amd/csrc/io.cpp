// io.cpp -- libhelen_io.so: native HDF5 reader/writer for the two file formats of the polish path.
//
// The per-window work of the reference's loader and writer is a handful of tiny HDF5 datasets
// (reader: helen/modules/python/models/dataloader_predict.py:64-82; writer:
// helen/modules/python/DataStore.py:99-133).  Doing that through Python costs ~400 us per window,
// 30x more than the MI355X needs for the window itself, so the same semantics are restated here
// against the HDF5 C API: whole batches per call, file handles kept open, types converted by HDF5.
// C ABI (extern "C"), no exceptions across it; helen_io_last_error() describes the last failure.
#include "../../include/helen_io.h"
#include <hdf5.h>

#include "h5emit.h"
#include "h5scan.h"

#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <type_traits>
#include <unordered_set>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

constexpr int kSeq = 1000;   // ImageSizeOptions.SEQ_LENGTH  (Options.py:16)
constexpr int kFeat = 90;    // ImageSizeOptions.IMAGE_HEIGHT (Options.py:14)
constexpr int kName = 256;   // bytes reserved per contig name in the batch arrays (a longer name is an error, never cut)

struct Quiet {
    Quiet() { H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr); }
};

// The reader entry points may be called from several threads at once (helen_io_read_image_runs starts its own):
// the caches below are guarded by g_cache_mutex; libhdf5 (not built thread-safe here) by g_library_mutex, one
// caller at a time; the direct scanner only reads a read-only mapping and needs no lock.
std::mutex g_cache_mutex;
std::recursive_mutex g_library_mutex;

// ---- reader side: per-process cache of open files -------------------------------------------
// What a cached handle / mapping was opened on.  A path can be rewritten while this process lives (a second
// polish into the same output, a test that reuses a name): the writers here truncate in place (same inode,
// smaller size: touching the old mapping past the new end is a SIGBUS) or a caller unlinks and recreates
// (new inode: the old one would be served silently).  Every cache hit is checked against stat().
struct FileIdentity {
    dev_t dev = 0;
    ino_t ino = 0;
    off_t size = -1;
    long long mtime_ns = 0;
    bool read(const char* path) {
        struct stat st;
        if (stat(path, &st) != 0) return false;
        dev = st.st_dev;
        ino = st.st_ino;
        size = st.st_size;
        mtime_ns = (long long)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
        return true;
    }
    bool operator==(const FileIdentity& o) const {
        return dev == o.dev && ino == o.ino && size == o.size && mtime_ns == o.mtime_ns;
    }
};
struct OpenFile {
    hid_t id = -1;
    FileIdentity identity;
};
std::map<std::string, OpenFile>& open_files() {
    static std::map<std::string, OpenFile> m;
    return m;
}

hid_t get_file(const char* path) {
    static Quiet q;
    auto& m = open_files();
    FileIdentity now;
    const bool present = now.read(path);
    auto it = m.find(path);
    if (it != m.end()) {
        if (present && it->second.identity == now) return it->second.id;
        H5Fclose(it->second.id);     // the path names another file now (or none)
        m.erase(it);
    }
    if (m.size() >= 64) {
        for (auto& kv : m) H5Fclose(kv.second.id);
        m.clear();
    }
    hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f >= 0) {
        OpenFile of;
        of.id = f;
        of.identity = now;
        m[path] = of;
    }
    return f;
}

// a path is about to be (or has just been) written by this process: no reader state of it may survive
void forget_path(const char* path);

// read a 1-element-or-more integer dataset, return element 0 as int64
int read_i64_first(hid_t loc, const char* name, int64_t* out) {
    hid_t d = H5Dopen2(loc, name, H5P_DEFAULT);
    if (d < 0) return -1;
    hid_t s = H5Dget_space(d);
    const hssize_t n = H5Sget_simple_extent_npoints(s);
    H5Sclose(s);
    int rc = -1;
    if (n >= 1) {
        std::vector<int64_t> buf((size_t)n);
        if (H5Dread(d, H5T_NATIVE_INT64, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.data()) >= 0) {
            *out = buf[0];
            rc = 0;
        }
    }
    H5Dclose(d);
    return rc;
}

// first element of a string dataset (fixed- or variable-length), NUL-terminated into out[cap];
// -1: missing / not a string, -2: does not fit into cap bytes (two names must never merge by truncation)
int read_str_first(hid_t loc, const char* name, char* out, size_t cap) {
    hid_t d = H5Dopen2(loc, name, H5P_DEFAULT);
    if (d < 0) return -1;
    hid_t t = H5Dget_type(d);
    hid_t s = H5Dget_space(d);
    const hssize_t n = H5Sget_simple_extent_npoints(s);
    int rc = -1;
    out[0] = 0;
    if (H5Tget_class(t) == H5T_STRING && n >= 1) {
        if (H5Tis_variable_str(t) > 0) {
            hid_t mt = H5Tcopy(H5T_C_S1);
            H5Tset_size(mt, H5T_VARIABLE);
            std::vector<char*> ptrs((size_t)n, nullptr);
            if (H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, ptrs.data()) >= 0) {
                rc = 0;
                if (ptrs[0]) {
                    if (strlen(ptrs[0]) >= cap) rc = -2;
                    snprintf(out, cap, "%s", ptrs[0]);
                }
                H5Dvlen_reclaim(mt, s, H5P_DEFAULT, ptrs.data());
            }
            H5Tclose(mt);
        } else {
            const size_t sz = H5Tget_size(t);
            std::vector<char> buf(sz * (size_t)n + 1, 0);
            hid_t mt = H5Tcopy(t);
            if (H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf.data()) >= 0) {
                const size_t len = strnlen(buf.data(), sz);
                const size_t c = len < cap - 1 ? len : cap - 1;
                memcpy(out, buf.data(), c);
                out[c] = 0;
                rc = len > cap - 1 ? -2 : 0;
            }
            H5Tclose(mt);
        }
    }
    H5Sclose(s);
    H5Tclose(t);
    H5Dclose(d);
    return rc;
}

// 2-D dataset [rows, cols] -> dst (row-major, `cols` wide) as `memtype`; rows returned
int read_2d(hid_t loc, const char* name, hid_t memtype, int cols, int max_rows, void* dst, int* rows) {
    hid_t d = H5Dopen2(loc, name, H5P_DEFAULT);
    if (d < 0) return -1;
    hid_t s = H5Dget_space(d);
    hsize_t dims[2] = {0, 0};
    int rc = -1;
    if (H5Sget_simple_extent_ndims(s) == 2) {
        H5Sget_simple_extent_dims(s, dims, nullptr);
        if ((int)dims[1] == cols && (int)dims[0] <= max_rows) {
            *rows = (int)dims[0];
            rc = (dims[0] == 0 || H5Dread(d, memtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, dst) >= 0) ? 0 : -1;
        } else {
            *rows = (int)dims[0];
            rc = -2;  // shape the path cannot take
        }
    }
    H5Sclose(s);
    H5Dclose(d);
    return rc;
}

// ---- reader side, fast path: the file's own structures walked in a read-only mapping (h5scan.h) ----------
// One entry per image file: the scanner and the object header of its `images` group; `usable` is false for a
// file the scanner does not take (libhdf5 reads it instead).  $HELEN_IO_READER=libhdf5 turns the fast path off;
// =direct turns the FALLBACK off (what the scanner declines is an error: the fuzz tests' way of exercising the
// scanner alone on damaged files).
std::atomic<long long> g_fast_windows{0}, g_library_windows{0};   // images read by the scanner / by libhdf5 in this process
struct Scanned {
    h5scan::File file;
    FileIdentity identity;
    bool usable = false;
    bool has_images = false;
    uint64_t images = 0;
    uint64_t last_used = 0;
};
std::map<std::string, std::shared_ptr<Scanned>>& scanned_files() {
    static std::map<std::string, std::shared_ptr<Scanned>> m;
    return m;
}
void forget_index(const char* path);
void forget_path(const char* path) {
    forget_index(path);
    std::lock_guard<std::recursive_mutex> lib(g_library_mutex);
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    auto& of = open_files();
    auto it = of.find(path);
    if (it != of.end()) {
        H5Fclose(it->second.id);
        of.erase(it);
    }
    scanned_files().erase(path);
}
bool reader_mode_is(const char* what) {
    const char* e = getenv("HELEN_IO_READER");
    return e && strcmp(e, what) == 0;
}
std::shared_ptr<Scanned> scan_file_shared(const char* path) {
    static const bool off = reader_mode_is("libhdf5");
    if (off) return nullptr;
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    // A reader walks its files one after the other: only the newest few GIGABYTES stay mapped.  (A process that had
    // read a 27 GB image directory through eight mappings took up to 2 s to exit -- the kernel unmaps page by
    // page -- and predict waits for its readers; dropped when the reader moves on, that cost runs beside the device
    // instead.  Prediction files are an eighth of that size, and stitch alternates between all of them: up to 64 of
    // those stay.)
    static uint64_t tick = 0;
    constexpr size_t kKeepBytes = (size_t)4 << 30;
    auto& m = scanned_files();
    FileIdentity now;
    const bool present = now.read(path);
    auto it = m.find(path);
    if (it != m.end()) {
        if (present && it->second->identity == now) {
            it->second->last_used = ++tick;
            return it->second->usable ? it->second : nullptr;
        }
        m.erase(it);     // rewritten, replaced or removed since it was mapped: never touch the old mapping again
    }
    for (;;) {
        size_t held = 0;
        for (auto& kv : m) held += kv.second->file.mapped_bytes();
        if (m.size() < 64 && (held < kKeepBytes || m.size() < 3)) break;
        auto oldest = m.begin();
        for (auto k = m.begin(); k != m.end(); ++k)
            if (k->second->last_used < oldest->second->last_used) oldest = k;
        m.erase(oldest);
    }
    std::shared_ptr<Scanned> sc(new Scanned());
    sc->identity = now;
    if (present && sc->file.open(path)) {
        std::vector<std::pair<std::string, uint64_t>> top;
        if (sc->file.children(sc->file.root(), &top)) {
            sc->usable = true;
            for (auto& kv : top)
                if (kv.first == "images") {
                    sc->has_images = true;
                    sc->images = kv.second;
                }
        }
    }
    if (!sc->usable) sc->file.close();
    sc->last_used = ++tick;
    m[path] = sc;
    return sc->usable ? sc : nullptr;
}
// the mapping stays alive (whatever the cache evicts meanwhile) until this thread's next call
Scanned* scan_file(const char* path) {
    thread_local std::shared_ptr<Scanned> hold;
    hold = scan_file_shared(path);
    return hold.get();
}

// element 0 of an integer dataset as int64 (any width, signed or not); false: not a plain integer dataset
bool scan_i64_first(const h5scan::File& f, uint64_t group, const char* name, int64_t* out) {
    uint64_t h;
    h5scan::Dataset d;
    if (!f.lookup(group, name, &h) || !f.dataset(h, &d) || d.cls != 0 || !d.data || d.count() < 1) return false;
    switch (d.size * 2 + (d.is_signed ? 1 : 0)) {
        case 2: *out = *(const uint8_t*)d.data; break;
        case 3: *out = *(const int8_t*)d.data; break;
        case 4: { uint16_t v; memcpy(&v, d.data, 2); *out = v; break; }
        case 5: { int16_t v; memcpy(&v, d.data, 2); *out = v; break; }
        case 8: { uint32_t v; memcpy(&v, d.data, 4); *out = v; break; }
        case 9: { int32_t v; memcpy(&v, d.data, 4); *out = v; break; }
        case 16: { uint64_t v; memcpy(&v, d.data, 8); *out = (int64_t)v; break; }
        case 17: { int64_t v; memcpy(&v, d.data, 8); *out = v; break; }
        default: return false;
    }
    return true;
}

// a [rows, cols] dataset converted like np.array(x, dtype=T) does: T = uint8_t (image) or int64_t (position)
template <typename T>
bool scan_2d(const h5scan::Dataset& d, T* dst) {
    const uint64_t n = d.count();
    const uint8_t* p = d.data;
    if (n && !p) return false;
#define HELEN_CONVERT(S) { for (uint64_t i = 0; i < n; ++i) { S v; memcpy(&v, p + i * sizeof(S), sizeof(S)); dst[i] = (T)v; } return true; }
    if (d.cls == 0) {
        if (d.size == (int)sizeof(T) && (sizeof(T) == 1 || d.is_signed == std::is_signed<T>::value)) {
            memcpy(dst, p, n * sizeof(T));
            return true;
        }
        switch (d.size * 2 + (d.is_signed ? 1 : 0)) {
            case 2: HELEN_CONVERT(uint8_t)
            case 3: HELEN_CONVERT(int8_t)
            case 4: HELEN_CONVERT(uint16_t)
            case 5: HELEN_CONVERT(int16_t)
            case 8: HELEN_CONVERT(uint32_t)
            case 9: HELEN_CONVERT(int32_t)
            case 16: HELEN_CONVERT(uint64_t)
            case 17: HELEN_CONVERT(int64_t)
            default: return false;
        }
    }
    if (d.cls == 1) {
        if (d.size == 4) { for (uint64_t i = 0; i < n; ++i) { float v; memcpy(&v, p + 4 * i, 4); dst[i] = (T)(int64_t)v; } return true; }
        if (d.size == 8) { for (uint64_t i = 0; i < n; ++i) { double v; memcpy(&v, p + 8 * i, 8); dst[i] = (T)(int64_t)v; } return true; }
    }
#undef HELEN_CONVERT
    return false;
}

// The contig string as the reference's reader returns it (dataloader_predict.py:64):
//     np.array2string(name.astype(np.str)).replace("'", '')
// array2string prints a string scalar as Python's repr does -- in single quotes, or in DOUBLE quotes when the name holds a
// single quote and no double quote; backslashes doubled, a quote of the delimiter's kind and control characters escaped --
// and the replace then drops every single quote.  For an ordinary name that is the name; "c'q" comes back as "cq" WITH
// its double quotes (checked against the reference's reader itself: tests/golden/make_golden_io.py).
std::string reference_contig_text(const std::string& raw) {
    const bool has_sq = raw.find('\'') != std::string::npos, has_dq = raw.find('"') != std::string::npos;
    const char quote = (has_sq && !has_dq) ? '"' : '\'';
    std::string r(1, quote);
    char esc[8];
    for (unsigned char c : raw) {
        if (c == (unsigned char)quote || c == '\\') {
            r += '\\';
            r += (char)c;
        } else if (c == '\t') {
            r += "\\t";
        } else if (c == '\n') {
            r += "\\n";
        } else if (c == '\r') {
            r += "\\r";
        } else if (c < 0x20 || c == 0x7f) {
            snprintf(esc, sizeof(esc), "\\x%02x", c);
            r += esc;
        } else {
            r += (char)c;
        }
    }
    r += quote;
    std::string out;
    for (char c : r)
        if (c != '\'') out.push_back(c);
    return out;
}

// helen_io_read_images through the scanner.  0: done; -1: the reader's error (message set); 1: something this
// scanner does not take -- the caller reads the batch through libhdf5 instead.
// One image group (object header `g`) into row 0 of the output arrays.  0: done; -1: the reader's error; 1: not for
// the scanner.
int fast_read_one(const h5scan::File& f, uint64_t g, const char* path, const char* name, uint8_t* img, int64_t* pos,
                  int64_t* meta, char* c, std::string* contig) {
    uint64_t h;
    h5scan::Dataset dc, di, dp;
    if (!f.lookup(g, "contig", &h) || !f.dataset(h, &dc) || !f.first_string(dc, contig)) return 1;
    if (!scan_i64_first(f, g, "contig_start", meta + 0) || !scan_i64_first(f, g, "contig_end", meta + 1) ||
        !scan_i64_first(f, g, "feature_chunk_idx", meta + 2))
        return 1;
    if (!f.lookup(g, "image", &h) || !f.dataset(h, &di)) return 1;
    if (!f.lookup(g, "position", &h) || !f.dataset(h, &dp)) return 1;
    if (di.cls > 1 || dp.cls != 0) return 1;
    const int rows = di.rank == 2 && di.dims[0] <= (uint64_t)kSeq ? (int)di.dims[0] : -1;
    if (di.rank != 2 || di.dims[1] != (uint64_t)kFeat || di.dims[0] > (uint64_t)kSeq || dp.rank != 2 || dp.dims[1] != 3 ||
        dp.dims[0] != di.dims[0])
        return fail("IMAGE SIZE ERROR: %s (%d, %d)", path, rows, kFeat);   // dataloader_predict.py:85-86
    if (!scan_2d<uint8_t>(di, img) || !scan_2d<int64_t>(dp, pos)) return 1;
    if (rows < kSeq) {                                                        // :74-82
        memset(img + (size_t)rows * kFeat, 0, (size_t)(kSeq - rows) * kFeat);
        for (int k = rows * 3; k < kSeq * 3; ++k) pos[k] = -1;
    }
    const std::string clean = reference_contig_text(*contig);
    if (clean.size() > (size_t)kName - 1)
        return fail("%s: image '%s': contig name longer than %d bytes", path, name, kName - 1);
    memcpy(c, clean.c_str(), clean.size() + 1);
    return 0;
}

// helen_io_read_images through the scanner.  0: done; -1: the reader's error (message set); 1: something this
// scanner does not take -- the caller reads the batch through libhdf5 instead.
int fast_read_images(Scanned* sc, const char* path, const char* names, int n, uint8_t* images, int64_t* positions,
                     int64_t* meta, char* contigs) {
    const h5scan::File& f = sc->file;
    if (!sc->has_images) return 1;
    const char* p = names;
    std::string contig;
    for (int i = 0; i < n; ++i) {
        const char* e = strchr(p, '\n');
        const std::string name = e ? std::string(p, e - p) : std::string(p);
        p = e ? e + 1 : p + name.size();
        uint64_t g;
        if (!f.lookup(sc->images, name.c_str(), &g)) return 1;
        const int rc = fast_read_one(f, g, path, name.c_str(), images + (size_t)i * kSeq * kFeat,
                                     positions + (size_t)i * kSeq * 3, meta + (size_t)i * 3, contigs + (size_t)i * kName,
                                     &contig);
        if (rc) return rc;
    }
    return 0;
}

// ---- image index: the members of `images` of one file in name order, listed once -----------------------------------
// The loader walks a file's images in name order (dataloader_predict.py:38-52); with the index a window is addressed by
// (file, position in that order): no name travels between the languages and the scanner needs no B-tree descent per image.
struct ImageIndex {
    FileIdentity identity;
    bool has_images = false;
    bool through_library = false;                           // the scanner declined the file: names only
    std::shared_ptr<Scanned> scanned;                       // keeps the mapping of `headers` alive
    std::vector<std::string> names;
    std::vector<uint64_t> headers;                          // object header of each image group (scanner)
};
std::map<std::string, std::shared_ptr<ImageIndex>>& image_indexes() {
    static std::map<std::string, std::shared_ptr<ImageIndex>> m;
    return m;
}
void forget_index(const char* path) {
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    if (path) image_indexes().erase(path);
    else image_indexes().clear();
}

// ---- writer side ------------------------------------------------------------------------------
struct Region {                      // predictions/<contig>/<contig-start-end>
    std::vector<h5emit::Child> kids; // contig_start, contig_end, then one group per chunk id
    uint64_t header = 0;             // its group as last emitted ...
    bool dirty = true;               // ... which lacks members added since
};
// The block of one window -- position uint32 [1000, 3], bases uint8 [1000], rles uint8 [1000] and the group over the three
// (DataStore.py:126-133) -- as h5emit emits it, made once; marks = where the three datasets' data goes.
const h5emit::File::Stamp& window_stamp() {
    static const h5emit::File::Stamp stamp = h5emit::File::make_stamp([](h5emit::File& f, std::vector<uint64_t>& marks) {
        static_assert((kSeq * 3) % 4 == 0 && kSeq % 8 == 0, "the stamped block keeps its datasets 8-aligned");
        const std::vector<uint8_t> zeros((size_t)kSeq * 12, 0);
        const uint64_t dp[2] = {(uint64_t)kSeq, 3}, dl[1] = {(uint64_t)kSeq};
        std::vector<h5emit::Child> kids(3);
        auto mark = [&](size_t bytes) { marks.push_back(f.last_data_address()); marks.push_back(bytes); };
        kids[0] = {"position", f.dataset(zeros.data(), (size_t)kSeq * 12, 4, false, 2, dp)};
        mark((size_t)kSeq * 12);
        kids[1] = {"bases", f.dataset(zeros.data(), kSeq, 1, false, 1, dl)};
        mark(kSeq);
        kids[2] = {"rles", f.dataset(zeros.data(), kSeq, 1, false, 1, dl)};
        mark(kSeq);
        return f.group(kids);
    });
    return stamp;
}

// A window's block of the prediction file, reserved by the writer's own thread (which does the bookkeeping and fixes the
// addresses) and FILLED -- the block's bytes, 3000 positions int64 -> uint32, two label rows -- by that thread and the
// helpers below, a flush-full of blocks at a time.
struct FillJob {
    uint8_t* block;
    uint64_t addr;
    const int64_t* positions;
    const uint8_t *bases, *rles;
};
void fill_window(const FillJob& j) {
    const h5emit::File::Stamp& st = window_stamp();
    h5emit::File::fill_block(st, j.block, j.addr);
    // int64 -> uint32: a -1 padding row wraps to 4294967295 (DataStore.py:128); aligned in the FILE, not in memory
    uint8_t* q = j.block + st.marks[0];
    const int64_t* p = j.positions;
    for (int k = 0; k < kSeq * 3; k += 4) {
        const uint32_t four[4] = {(uint32_t)p[k], (uint32_t)p[k + 1], (uint32_t)p[k + 2], (uint32_t)p[k + 3]};
        memcpy(q + 4 * k, four, 16);
    }
    memcpy(j.block + st.marks[2], j.bases, kSeq);
    memcpy(j.block + st.marks[4], j.rles, kSeq);
}
class FillPool {
   public:
    explicit FillPool(int helpers) {
        for (int t = 0; t < helpers; ++t) threads_.emplace_back([this]() { loop(); });
    }
    ~FillPool() {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            stop_ = true;
        }
        wake_.notify_all();
        for (auto& t : threads_) t.join();
    }
    void add(const FillJob& j) { jobs_.push_back(j); }
    size_t pending() const { return jobs_.size(); }
    // every queued block filled when this returns (the caller works too), and no helper still looking at the job list: a
    // helper that wakes late for a round that is over finds left_ == 0 under the lock and goes back to sleep
    void run() {
        if (jobs_.empty()) return;
        if (threads_.empty() || jobs_.size() < 8) {
            for (const FillJob& j : jobs_) fill_window(j);
            jobs_.clear();
            return;
        }
        {
            std::lock_guard<std::mutex> lock(mutex_);
            next_.store(0);
            left_ = jobs_.size();
            ++round_;
        }
        wake_.notify_all();
        work();
        std::unique_lock<std::mutex> lock(mutex_);
        done_.wait(lock, [this]() { return left_ == 0 && busy_ == 0; });
        jobs_.clear();
    }

   private:
    void work() {
        size_t mine = 0;
        const size_t n = jobs_.size();          // (fixed for the round: nobody adds while a round runs)
        for (;;) {
            const size_t k = next_.fetch_add(1);
            if (k >= n) break;
            fill_window(jobs_[k]);
            ++mine;
        }
        if (mine) {
            std::lock_guard<std::mutex> lock(mutex_);
            left_ -= mine;
            if (left_ == 0) done_.notify_all();
        }
    }
    void loop() {
        unsigned long seen = 0;
        std::unique_lock<std::mutex> lock(mutex_);
        for (;;) {
            wake_.wait(lock, [&]() { return stop_ || round_ != seen; });
            if (stop_) return;
            seen = round_;
            if (left_ == 0) continue;           // that round is over already
            ++busy_;
            lock.unlock();
            work();
            lock.lock();
            if (--busy_ == 0 && left_ == 0) done_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::vector<FillJob> jobs_;
    std::mutex mutex_;
    std::condition_variable wake_, done_;
    std::atomic<size_t> next_{0};
    size_t left_ = 0, busy_ = 0;
    unsigned long round_ = 0;
    bool stop_ = false;
};

struct Writer {
    hid_t file = -1;
    hid_t lcpl = -1, dcpl = -1;
    hid_t space_pos = -1, space_lab = -1, space_scalar = -1;
    std::unordered_set<std::string> regions;   // DataStore.py:115  meta['predictions_contig']
    std::unordered_set<std::string> images;    // DataStore.py:123  meta['predictions']
    std::vector<uint32_t> pos32;
    // direct emitter (h5emit.h): the default; $HELEN_IO_WRITER=libhdf5 selects the library path above
    h5emit::File* fast = nullptr;
    FillPool* pool = nullptr;        // helpers that fill the reserved window blocks ($HELEN_IO_WRITER_THREADS - 1; default: one)
    std::string path;
    std::map<std::string, std::map<std::string, Region>> tree;   // contig -> region name -> members
    Region* open_region = nullptr;   // the region of the newest window: its group is emitted when the next begins
};
// A region's group goes out as soon as the stream moves on to another region (windows arrive file by file, a
// region's chunk ids together), so that close only has the contig groups left to write -- not 300 k region
// groups.  A region that does come back later is simply emitted again at close with all its members: the earlier
// copy of its group structure is then unreferenced space in the file.
void settle_region(Writer* w, Region* r) {
    if (r && r->dirty) {
        r->header = w->fast->group(r->kids);
        r->dirty = false;
    }
}

int write_ds(Writer* w, hid_t loc, const std::string& path, hid_t ftype, hid_t mtype, hid_t space, const void* buf) {
    hid_t d = H5Dcreate2(loc, path.c_str(), ftype, space, w->lcpl, w->dcpl, H5P_DEFAULT);
    if (d < 0) return fail("cannot create dataset '%s'", path.c_str());
    const herr_t rc = H5Dwrite(d, mtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf);
    H5Dclose(d);
    return rc < 0 ? fail("cannot write dataset '%s'", path.c_str()) : 0;
}

// helen_io_read_images through libhdf5 (the caller holds g_library_mutex)
int library_read_images(hid_t f, const char* path, const char* names, int n, uint8_t* images, int64_t* positions,
                        int64_t* meta, char* contigs) {
    const char* p = names;
    for (int i = 0; i < n; ++i) {
        const char* e = strchr(p, '\n');
        std::string name = e ? std::string(p, e - p) : std::string(p);
        p = e ? e + 1 : p + name.size();
        const std::string gpath = "images/" + name;
        hid_t g = H5Gopen2(f, gpath.c_str(), H5P_DEFAULT);
        if (g < 0) return fail("%s: no image '%s'", path, name.c_str());
        uint8_t* img = images + (size_t)i * kSeq * kFeat;
        int64_t* pos = positions + (size_t)i * kSeq * 3;
        int rows = 0, prow = 0;
        int rc = read_str_first(g, "contig", contigs + (size_t)i * kName, kName);
        if (rc == -2) {
            H5Gclose(g);
            return fail("%s: image '%s': contig name longer than %d bytes", path, name.c_str(), kName - 1);
        }
        rc |= read_i64_first(g, "contig_start", meta + (size_t)i * 3 + 0);
        rc |= read_i64_first(g, "contig_end", meta + (size_t)i * 3 + 1);
        rc |= read_i64_first(g, "feature_chunk_idx", meta + (size_t)i * 3 + 2);
        if (rc) {
            H5Gclose(g);
            return fail("%s: image '%s' lacks contig / contig_start / contig_end / feature_chunk_idx", path,
                        name.c_str());
        }
        const int r1 = read_2d(g, "image", H5T_NATIVE_UINT8, kFeat, kSeq, img, &rows);
        const int r2 = read_2d(g, "position", H5T_NATIVE_INT64, 3, kSeq, pos, &prow);
        H5Gclose(g);
        if (r1 || r2 || prow != rows)
            return fail("IMAGE SIZE ERROR: %s (%d, %d)", path, rows, kFeat);   // :85-86
        if (rows < kSeq) {                                                        // :74-82
            memset(img + (size_t)rows * kFeat, 0, (size_t)(kSeq - rows) * kFeat);
            for (int k = rows * 3; k < kSeq * 3; ++k) pos[k] = -1;
        }
        char* c = contigs + (size_t)i * kName;
        const std::string clean = reference_contig_text(c);
        if (clean.size() > (size_t)kName - 1)
            return fail("%s: image '%s': contig name longer than %d bytes", path, name.c_str(), kName - 1);
        memcpy(c, clean.c_str(), clean.size() + 1);
    }
    return 0;
}


}  // namespace

extern "C" {

const char* helen_io_last_error(void) { return g_err; }

int helen_io_abi_version(void) { return 1; }

/* Names of the members of group `images` of one file, in name order (what h5py's .keys() yields;
 * dataloader_predict.py:41), '\n'-separated into `out` (capacity `cap`).  *n_out = count, or 0 and
 * return 1 if the file has no `images` group (the reader warns and skips, :47-49).
 * Returns -1 on error, -2 if `cap` is too small (*n_out then holds the bytes needed). */
int helen_io_list_images(const char* path, char* out, size_t cap, long long* n_out) {
    if (Scanned* sc = scan_file(path)) {
        *n_out = 0;
        if (!sc->has_images) return 1;
        std::vector<std::pair<std::string, uint64_t>> kids;
        if (sc->file.children(sc->images, &kids)) {      // B-tree order = name order = what h5py's .keys() yields
            size_t used = 0;
            for (auto& kv : kids) {
                if (used + kv.first.size() + 1 <= cap) {
                    memcpy(out + used, kv.first.data(), kv.first.size());
                    out[used + kv.first.size()] = '\n';
                }
                used += kv.first.size() + 1;
            }
            if (used > cap) {
                *n_out = (long long)used;
                return -2;
            }
            *n_out = (long long)kids.size();
            return 0;
        }
    }
    static const bool direct_only = reader_mode_is("direct");
    if (direct_only) return fail("%s: not a file the direct scanner takes", path);
    std::lock_guard<std::recursive_mutex> lib(g_library_mutex);
    hid_t f = get_file(path);
    if (f < 0) return fail("cannot open '%s'", path);
    *n_out = 0;
    const htri_t has_images = H5Lexists(f, "images", H5P_DEFAULT);
    if (has_images < 0) return fail("%s: cannot look up 'images'", path);   // damaged, not absent
    if (has_images == 0) return 1;
    hid_t g = H5Gopen2(f, "images", H5P_DEFAULT);
    if (g < 0) return fail("%s: cannot open group 'images'", path);
    // one pass in name order (H5Lget_name_by_idx per member is O(n) each on symbol-table groups)
    struct Acc {
        char* out;
        size_t cap, used;
        long long count;
    } acc = {out, cap, 0, 0};
    auto cb = [](hid_t, const char* name, const H5L_info_t*, void* ud) -> herr_t {
        Acc* a = (Acc*)ud;
        const size_t n = strlen(name);
        if (a->used + n + 1 <= a->cap) {
            memcpy(a->out + a->used, name, n);
            a->out[a->used + n] = '\n';
        }
        a->used += n + 1;
        a->count += 1;
        return 0;
    };
    const herr_t it = H5Literate(g, H5_INDEX_NAME, H5_ITER_INC, nullptr, cb, &acc);
    H5Gclose(g);
    if (it < 0) return fail("%s: cannot list 'images'", path);
    if (acc.used > cap) {
        *n_out = (long long)acc.used;
        return -2;
    }
    *n_out = acc.count;
    return 0;
}

/* Read `n` images of one file (names '\n'-separated) with the reader's semantics
 * (dataloader_predict.py:54-88): image -> uint8 [1000, 90], position -> int64 [1000, 3]; a short image
 * is padded with zero rows and (-1,-1,-1) position rows; anything that is not [<=1000, 90] / [l, 3]
 * is the reader's "IMAGE SIZE ERROR".
 *   images    [n, 1000, 90] uint8      positions [n, 1000, 3] int64
 *   meta      [n, 3] int64 = contig_start, contig_end, feature_chunk_idx
 *   contigs   [n, 256] char, NUL-terminated */
int helen_io_read_images(const char* path, const char* names, int n, uint8_t* images, int64_t* positions,
                         int64_t* meta, char* contigs) {
    if (Scanned* sc = scan_file(path)) {
        const int rc = fast_read_images(sc, path, names, n, images, positions, meta, contigs);
        if (rc == 0) g_fast_windows += n;
        if (rc <= 0) return rc;       // done, or the reader's own error; 1 = not for the scanner: libhdf5 below
    }
    static const bool direct_only = reader_mode_is("direct");
    if (direct_only) return fail("%s: not a file the direct scanner takes", path);
    g_library_windows += n;
    std::lock_guard<std::recursive_mutex> lib(g_library_mutex);
    hid_t f = get_file(path);
    if (f < 0) return fail("cannot open '%s'", path);
    return library_read_images(f, path, names, n, images, positions, meta, contigs);
}

/* ---- the same reader addressed by position: (file, first image, count) runs, several threads ------------------- */
static std::shared_ptr<ImageIndex> image_index(const char* path) {
    FileIdentity now;
    const bool present = now.read(path);
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        auto it = image_indexes().find(path);
        if (it != image_indexes().end()) {
            if (present && it->second->identity == now) return it->second;
            image_indexes().erase(it);
        }
    }
    std::shared_ptr<ImageIndex> ix(new ImageIndex());
    ix->identity = now;
    bool listed = false;
    if (std::shared_ptr<Scanned> sc = scan_file_shared(path)) {
        if (!sc->has_images) {
            listed = true;
        } else {
            std::vector<std::pair<std::string, uint64_t>> kids;
            if (sc->file.children(sc->images, &kids)) {
                listed = true;
                ix->has_images = true;
                ix->scanned = sc;
                ix->names.reserve(kids.size());
                ix->headers.reserve(kids.size());
                for (auto& kv : kids) {
                    ix->names.push_back(std::move(kv.first));
                    ix->headers.push_back(kv.second);
                }
            }
        }
    }
    if (!listed) {
        static const bool direct_only = reader_mode_is("direct");
        if (direct_only) {
            fail("%s: not a file the direct scanner takes", path);
            return nullptr;
        }
        std::vector<char> buf((size_t)1 << 20);
        long long n = 0;
        int rc;
        while ((rc = helen_io_list_images(path, buf.data(), buf.size(), &n)) == -2) buf.resize((size_t)n + 16);
        if (rc < 0) return nullptr;
        ix->through_library = true;
        ix->has_images = rc == 0;
        const char* p = buf.data();
        for (long long i = 0; i < n; ++i) {
            const char* e = (const char*)memchr(p, '\n', buf.data() + buf.size() - p);
            if (!e) break;
            ix->names.emplace_back(p, e - p);
            p = e + 1;
        }
    }
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    if (image_indexes().size() >= 4096) image_indexes().clear();
    image_indexes()[path] = ix;
    return ix;
}

/* Number of images of one file (*n_out) and whether libhdf5 has to read them (*through_library).  Returns 1 (and
 * *n_out = 0) if the file has no `images` group. */
int helen_io_index_images(const char* path, long long* n_out, int* through_library) {
    *n_out = 0;
    if (through_library) *through_library = 0;
    const std::shared_ptr<ImageIndex> ix = image_index(path);
    if (!ix) return -1;
    if (through_library) *through_library = ix->through_library ? 1 : 0;
    if (!ix->has_images) return 1;
    *n_out = (long long)ix->names.size();
    return 0;
}

/* Names of images [first, first + count) of the index, '\n'-separated; -2 if `cap` is too small (*needed = bytes). */
int helen_io_image_names(const char* path, long long first, long long count, char* out, size_t cap, long long* needed) {
    const std::shared_ptr<ImageIndex> ix = image_index(path);
    if (!ix) return -1;
    if (first < 0 || count < 0 || (size_t)(first + count) > ix->names.size()) return fail("%s: image range out of bounds", path);
    size_t used = 0;
    for (long long i = first; i < first + count; ++i) {
        const std::string& nm = ix->names[(size_t)i];
        if (used + nm.size() + 1 <= cap) {
            memcpy(out + used, nm.data(), nm.size());
            out[used + nm.size()] = '\n';
        }
        used += nm.size() + 1;
    }
    if (needed) *needed = (long long)used;
    if (used + 1 > cap) return -2;
    out[used] = 0;
    return 0;
}

/* helen_io_read_images for images [first, first + count) of the file's index.  Thread-safe. */
int helen_io_read_image_range(const char* path, long long first, int count, uint8_t* images, int64_t* positions,
                              int64_t* meta, char* contigs, long long* through_library) {
    const std::shared_ptr<ImageIndex> ix = image_index(path);
    if (!ix) return -1;
    if (first < 0 || count < 0 || (size_t)(first + count) > ix->names.size()) return fail("%s: image range out of bounds", path);
    int done = 0;
    if (!ix->through_library) {
        const h5scan::File& f = ix->scanned->file;
        std::string contig;
        for (; done < count; ++done) {
            const size_t k = (size_t)first + done;
            const int rc = fast_read_one(f, ix->headers[k], path, ix->names[k].c_str(), images + (size_t)done * kSeq * kFeat,
                                         positions + (size_t)done * kSeq * 3, meta + (size_t)done * 3,
                                         contigs + (size_t)done * kName, &contig);
            if (rc < 0) return rc;
            if (rc > 0) break;          // this image is not for the scanner: the rest of the range goes to libhdf5
        }
        g_fast_windows += done;
        if (done == count) return 0;
        static const bool direct_only = reader_mode_is("direct");
        if (direct_only) return fail("%s: not a file the direct scanner takes", path);
    }
    std::string names;
    for (int i = done; i < count; ++i) {
        names += ix->names[(size_t)first + i];
        names += '\n';
    }
    if (through_library) *through_library += count - done;
    // (helen_io_read_images tries the scanner first: switched off for this call by going to the library code directly)
    std::lock_guard<std::recursive_mutex> lib(g_library_mutex);
    g_library_windows += count - done;
    hid_t f = get_file(path);
    if (f < 0) return fail("cannot open '%s'", path);
    return library_read_images(f, path, names.c_str(), count - done, images + (size_t)done * kSeq * kFeat,
                               positions + (size_t)done * kSeq * 3, meta + (size_t)done * 3, contigs + (size_t)done * kName);
}

/* How a file's images are stored, from its first image (a MarginPolish run writes all of them alike): out[0] = 0 the direct
 * scanner reads the file, 1 libhdf5 has to; out[1] = layout class of `image` (0 compact, 1 contiguous, 2 chunked, -1 not
 * inspected); out[2] = number of filters; out[3] = 1 if deflate is among them.  Returns 1 if the file has no images. */
int helen_io_fast_inflate(void) { return h5scan::fast_inflate().ok ? 1 : 0; }

int helen_io_image_storage(const char* path, int* out) {
    out[0] = 0;
    out[1] = -1;
    out[2] = 0;
    out[3] = 0;
    const std::shared_ptr<ImageIndex> ix = image_index(path);
    if (!ix) return -1;
    if (!ix->has_images || ix->names.empty()) return 1;
    if (ix->through_library) {
        out[0] = 1;
        return 0;
    }
    const h5scan::File& f = ix->scanned->file;
    uint64_t h;
    int cls = -1, nf = 0;
    bool z = false;
    h5scan::Dataset d;
    if (!f.lookup(ix->headers[0], "image", &h) || !f.storage(h, &cls, &nf, &z) || !f.dataset(h, &d)) {
        out[0] = 1;                 // the groups are the scanner's, the datasets are not: libhdf5 reads the images
        return 0;
    }
    out[1] = cls;
    out[2] = nf;
    out[3] = z ? 1 : 0;
    return 0;
}

/* A reader that has moved past a file lets go of its index and its mapping (unmapping gigabytes of touched pages takes
 * the kernel tenths of a second: call this from a thread that has nothing better to do). */
void helen_io_forget_images(const char* path) {
    std::shared_ptr<ImageIndex> ix;
    std::shared_ptr<Scanned> sc;
    {
        std::lock_guard<std::mutex> lock(g_cache_mutex);
        auto it = image_indexes().find(path);
        if (it != image_indexes().end()) {
            ix = it->second;
            image_indexes().erase(it);
        }
        auto js = scanned_files().find(path);
        if (js != scanned_files().end()) {
            sc = js->second;
            scanned_files().erase(js);
        }
    }
    ix.reset();
    sc.reset();      // the last reference unmaps here, outside the lock
}

/* `n_runs` ranges (paths[i], firsts[i], counts[i]) read into consecutive rows of the output arrays by `threads`
 * threads of this call (pieces of at most 32 images, taken in order).  *through_library = images libhdf5 read (those are
 * serialised: the library is not thread-safe).  The first failing piece's error is this call's. */
int helen_io_read_image_runs(int n_runs, const char* const* paths, const long long* firsts, const int* counts, int threads,
                             uint8_t* images, int64_t* positions, int64_t* meta, char* contigs,
                             long long* through_library) {
    struct Piece {
        int run;
        long long first;
        int count;
        size_t row;
    };
    std::vector<Piece> pieces;
    size_t row = 0;
    for (int r = 0; r < n_runs; ++r)
        for (int o = 0; o < counts[r]; o += 32) {
            const int c = std::min(32, counts[r] - o);
            pieces.push_back({r, firsts[r] + o, c, row});
            row += (size_t)c;
        }
    if (through_library) *through_library = 0;
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    std::atomic<long long> lib_total{0};
    std::mutex err_mutex;
    std::string first_error;
    auto work = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= pieces.size() || failed.load()) return;
            const Piece& p = pieces[k];
            long long lib = 0;
            const int rc = helen_io_read_image_range(paths[p.run], p.first, p.count, images + p.row * kSeq * kFeat,
                                                     positions + p.row * kSeq * 3, meta + p.row * 3,
                                                     contigs + p.row * kName, &lib);
            lib_total += lib;
            if (rc != 0) {
                std::lock_guard<std::mutex> lock(err_mutex);
                if (!failed.exchange(true)) first_error = g_err;
                return;
            }
        }
    };
    const int nt = std::max(1, std::min<int>(threads, (int)pieces.size()));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (through_library) *through_library = lib_total.load();
    if (failed.load()) return fail("%s", first_error.c_str());
    return 0;
}

/* Labeled images for the evaluation path (`models/dataloader.py:48-61`, the loader of `helen_train test`): for `n`
 * images of one file (names '\n'-separated) the image as uint8 [1000, 90] and label_base / label_run_length as uint8
 * [1000], exactly as stored -- that loader does not pad, so anything else is an "IMAGE SIZE ERROR".
 *   images [n, 1000, 90] uint8      label_base, label_rle [n, 1000] uint8 */
int helen_io_read_labeled(const char* path, const char* names, int n, uint8_t* images, uint8_t* label_base,
                          uint8_t* label_rle) {
    Scanned* sc = scan_file(path);
    bool fast = sc && sc->has_images;
    hid_t f = -1;
    const char* p = names;
    for (int i = 0; i < n; ++i) {
        const char* e = strchr(p, '\n');
        const std::string name = e ? std::string(p, e - p) : std::string(p);
        p = e ? e + 1 : p + name.size();
        uint8_t* img = images + (size_t)i * kSeq * kFeat;
        uint8_t* lb = label_base + (size_t)i * kSeq;
        uint8_t* lr = label_rle + (size_t)i * kSeq;
        if (fast) {
            const h5scan::File& sf = sc->file;
            uint64_t g, h;
            h5scan::Dataset di, db, dr;
            bool took = sf.lookup(sc->images, name.c_str(), &g) && sf.lookup(g, "image", &h) && sf.dataset(h, &di) &&
                        sf.lookup(g, "label_base", &h) && sf.dataset(h, &db) && sf.lookup(g, "label_run_length", &h) &&
                        sf.dataset(h, &dr) && di.cls <= 1 && db.cls == 0 && dr.cls == 0;
            if (took) {
                if (di.rank != 2 || di.dims[0] != (uint64_t)kSeq || di.dims[1] != (uint64_t)kFeat || db.rank != 1 ||
                    db.dims[0] != (uint64_t)kSeq || dr.rank != 1 || dr.dims[0] != (uint64_t)kSeq)
                    return fail("IMAGE SIZE ERROR: %s image '%s'", path, name.c_str());
                took = scan_2d<uint8_t>(di, img) && scan_2d<uint8_t>(db, lb) && scan_2d<uint8_t>(dr, lr);
            }
            if (took) {
                ++g_fast_windows;
                continue;
            }
            fast = false;     // this file is libhdf5's from here on
        }
        static const bool direct_only = reader_mode_is("direct");
        if (direct_only) return fail("%s: not a file the direct scanner takes", path);
        std::lock_guard<std::recursive_mutex> lib(g_library_mutex);
        if (f < 0) f = get_file(path);
        if (f < 0) return fail("cannot open '%s'", path);
        const std::string gpath = "images/" + name;
        hid_t g = H5Gopen2(f, gpath.c_str(), H5P_DEFAULT);
        if (g < 0) return fail("%s: no image '%s'", path, name.c_str());
        int rows = 0;
        const int r1 = read_2d(g, "image", H5T_NATIVE_UINT8, kFeat, kSeq, img, &rows);
        bool good = r1 == 0 && rows == kSeq;
        for (int which = 0; good && which < 2; ++which) {
            hid_t d = H5Dopen2(g, which ? "label_run_length" : "label_base", H5P_DEFAULT);
            good = d >= 0;
            if (good) {
                hid_t sp = H5Dget_space(d);
                hsize_t dim = 0;
                good = H5Sget_simple_extent_ndims(sp) == 1 && H5Sget_simple_extent_dims(sp, &dim, nullptr) == 1 &&
                       dim == (hsize_t)kSeq &&
                       H5Dread(d, H5T_NATIVE_UINT8, H5S_ALL, H5S_ALL, H5P_DEFAULT, which ? lr : lb) >= 0;
                H5Sclose(sp);
                H5Dclose(d);
            }
        }
        H5Gclose(g);
        if (!good) return fail("IMAGE SIZE ERROR: %s image '%s'", path, name.c_str());
        ++g_library_windows;
    }
    return 0;
}

/* Synthetic MarginPolish image file through the direct emitter (helen_amd.synthetic: benchmark inputs; the files the
 * reader TESTS use are written by libhdf5): `n` windows named <contig>-<start>-<end>-<chunk> under `images/`, each with
 * the six datasets of dataloader_predict.py:64-70 -- contig (fixed string [1]), contig_start / contig_end /
 * feature_chunk_idx (int64 [1]), image uint8 [L, 90], position int64 [L, 3] = (start + k, 0, 0).
 *   starts, chunks  int64 [n];  lengths int32 [n] (rows stored, <= 1000);  images uint8 [n, 1000, 90] */
int helen_io_emit_images(const char* path, int n, const char* contig, const int64_t* starts, const int64_t* chunks,
                         const int32_t* lengths, const uint8_t* images) {
    forget_path(path);
    h5emit::File f;
    if (!f.open(path)) return fail("cannot create '%s'", path);
    std::vector<h5emit::Child> all;
    std::vector<int64_t> pos((size_t)kSeq * 3);
    char name[512];
    for (int i = 0; i < n; ++i) {
        const int L = lengths[i];
        if (L < 0 || L > kSeq) return fail("bad length %d", L);
        for (int k = 0; k < L; ++k) {
            pos[3 * k] = starts[i] + k;
            pos[3 * k + 1] = pos[3 * k + 2] = 0;
        }
        const uint64_t one[1] = {1}, di[2] = {(uint64_t)L, (uint64_t)kFeat}, dp[2] = {(uint64_t)L, 3};
        const int64_t cs = starts[i], ce = starts[i] + 1000, ch = chunks[i];
        std::vector<h5emit::Child> kids(6);
        kids[0] = {"contig", f.string1(contig)};
        kids[1] = {"contig_start", f.dataset(&cs, 8, 8, true, 1, one)};
        kids[2] = {"contig_end", f.dataset(&ce, 8, 8, true, 1, one)};
        kids[3] = {"feature_chunk_idx", f.dataset(&ch, 8, 8, true, 1, one)};
        kids[4] = {"image", f.dataset(images + (size_t)i * kSeq * kFeat, (size_t)L * kFeat, 1, false, 2, di)};
        kids[5] = {"position", f.dataset(pos.data(), (size_t)L * 24, 8, true, 2, dp)};
        snprintf(name, sizeof(name), "%s-%lld-%lld-%lld", contig, (long long)cs, (long long)ce, (long long)ch);
        all.push_back({name, f.group(kids)});
    }
    std::vector<h5emit::Child> top{{"images", f.group(all)}};
    uint64_t bt = 0, hp = 0;
    const uint64_t root = f.group(top, &bt, &hp);
    return f.finish(root, bt, hp) ? 0 : fail("writing '%s' failed", path);
}

/* The general form of helen_io_emit_images: every window names its own contig, contig_end and position rows (the
 * simulated assemblies of helen_amd.synthetic.write_assembly_dir: several contigs per file, insert and split rows).
 *   contigs char [n, 256];  starts, ends, chunks int64 [n];  lengths int32 [n];  images uint8 [n, 1000, 90];
 *   positions int64 [n, 1000, 3] (the first lengths[i] rows of each are stored) */
int helen_io_emit_image_windows(const char* path, int n, const char* contigs, const int64_t* starts, const int64_t* ends,
                                const int64_t* chunks, const int32_t* lengths, const uint8_t* images,
                                const int64_t* positions) {
    forget_path(path);
    h5emit::File f;
    if (!f.open(path)) return fail("cannot create '%s'", path);
    std::vector<h5emit::Child> all;
    char name[kName + 96];
    for (int i = 0; i < n; ++i) {
        const int L = lengths[i];
        if (L < 0 || L > kSeq) return fail("bad length %d", L);
        const char* contig = contigs + (size_t)i * kName;
        if (strnlen(contig, kName) == (size_t)kName) return fail("contig name of window %d is not NUL-terminated", i);
        const uint64_t one[1] = {1}, di[2] = {(uint64_t)L, (uint64_t)kFeat}, dp[2] = {(uint64_t)L, 3};
        const int64_t cs = starts[i], ce = ends[i], ch = chunks[i];
        std::vector<h5emit::Child> kids(6);
        kids[0] = {"contig", f.string1(contig)};
        kids[1] = {"contig_start", f.dataset(&cs, 8, 8, true, 1, one)};
        kids[2] = {"contig_end", f.dataset(&ce, 8, 8, true, 1, one)};
        kids[3] = {"feature_chunk_idx", f.dataset(&ch, 8, 8, true, 1, one)};
        kids[4] = {"image", f.dataset(images + (size_t)i * kSeq * kFeat, (size_t)L * kFeat, 1, false, 2, di)};
        kids[5] = {"position", f.dataset(positions + (size_t)i * kSeq * 3, (size_t)L * 24, 8, true, 2, dp)};
        snprintf(name, sizeof(name), "%s-%lld-%lld-%lld", contig, (long long)cs, (long long)ce, (long long)ch);
        all.push_back({name, f.group(kids)});
    }
    std::vector<h5emit::Child> top{{"images", f.group(all)}};
    uint64_t bt = 0, hp = 0;
    const uint64_t root = f.group(top, &bt, &hp);
    return f.finish(root, bt, hp) ? 0 : fail("writing '%s' failed", path);
}

/* Images this process has read through the direct scanner (out[0]) and through libhdf5 (out[1]). */
void helen_io_reader_counts(long long* out) {
    out[0] = g_fast_windows;
    out[1] = g_library_windows;
}

/* The lock around every libhdf5 call of this library, for a thread of the same process that enters libhdf5 through
 * another binding: lock, use the library, unlock -- on one thread (the mutex is recursive). */
void helen_io_library_lock(void) { g_library_mutex.lock(); }
void helen_io_library_unlock(void) { g_library_mutex.unlock(); }

/* Drop every cached read handle of this process. */
void helen_io_close_readers(void) {
    forget_index(nullptr);
    std::lock_guard<std::recursive_mutex> lib(g_library_mutex);
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    for (auto& kv : open_files()) H5Fclose(kv.second.id);
    open_files().clear();
    scanned_files().clear();
}

/* Prediction file writer (DataStore(filename, 'w'), predict_gpu.py:55). */
void* helen_io_writer_open(const char* path) {
    static Quiet q;
    forget_path(path);
    Writer* w = new Writer();
    w->path = path;
    w->pos32.resize((size_t)kSeq * 3);
    const char* which = getenv("HELEN_IO_WRITER");
    if (!(which && strcmp(which, "libhdf5") == 0)) {
        w->fast = new h5emit::File();
        if (!w->fast->open(path)) {
            fail("cannot create '%s'", path);
            delete w->fast;
            delete w;
            return nullptr;
        }
        if (window_stamp().bytes.empty() || window_stamp().marks.size() != 6) {     // (never: the block's layout is fixed)
            fail("internal: the window block of a prediction file is not stampable");
            delete w->fast;
            delete w;
            return nullptr;
        }
        int helpers = 1;
        if (const char* t = getenv("HELEN_IO_WRITER_THREADS")) helpers = std::max(0, std::min(15, atoi(t) - 1));
        w->pool = new FillPool(helpers);
        FillPool* pool = w->pool;
        w->fast->set_before_flush([pool]() { pool->run(); });
        return w;
    }
    std::lock_guard<std::recursive_mutex> lib(g_library_mutex);
    hid_t fapl = H5Pcreate(H5P_FILE_ACCESS);
    // thousands of tiny objects: allocate metadata in 1 MiB blocks (+10 % writes/s, same format)
    H5Pset_meta_block_size(fapl, (hsize_t)1 << 20);
    w->file = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, fapl);
    H5Pclose(fapl);
    if (w->file < 0) {
        fail("cannot create '%s'", path);
        delete w;
        return nullptr;
    }
    w->lcpl = H5Pcreate(H5P_LINK_CREATE);
    H5Pset_create_intermediate_group(w->lcpl, 1);
    w->dcpl = H5Pcreate(H5P_DATASET_CREATE);
    hsize_t dp[2] = {kSeq, 3}, dl[1] = {kSeq};
    w->space_pos = H5Screate_simple(2, dp, nullptr);
    w->space_lab = H5Screate_simple(1, dl, nullptr);
    w->space_scalar = H5Screate(H5S_SCALAR);
    w->pos32.resize((size_t)kSeq * 3);
    return w;
}

/* DataStore.write_prediction for `n` windows (DataStore.py:83-133): scalar int64 contig_start /
 * contig_end once per region, then position uint32 [1000,3] (-1 wraps to 4294967295), bases uint8
 * [1000], rles uint8 [1000] once per (contig, region, chunk id); repeats are skipped silently.
 * `sel` (optional) lists the rows of the batch arrays to write -- several writer processes share one
 * batch, each taking the regions assigned to it. */
int helen_io_write_predictions_sel(void* handle, int n_sel, const int32_t* sel, const char* contigs,
                                   const int64_t* meta, const int64_t* positions, const uint8_t* bases,
                                   const uint8_t* rles) {
    Writer* w = (Writer*)handle;
    if (!w) return fail("null writer");
    char num[64];
    for (int j = 0; j < n_sel; ++j) {
        const int i = sel ? sel[j] : j;
        const std::string contig = contigs + (size_t)i * kName;
        const int64_t cs = meta[(size_t)i * 3], ce = meta[(size_t)i * 3 + 1], chunk = meta[(size_t)i * 3 + 2];
        snprintf(num, sizeof(num), "-%lld-%lld", (long long)cs, (long long)ce);
        const std::string prefix = contig + num;
        snprintf(num, sizeof(num), "%lld", (long long)chunk);
        const std::string suffix = num;
        if (w->fast) {
            // the same bookkeeping (DataStore.py:115-124), the bytes written directly: datasets now, groups at close
            Region& reg = w->tree[contig][prefix];
            if (&reg != w->open_region) {
                settle_region(w, w->open_region);
                w->open_region = &reg;
            }
            if (w->regions.insert(prefix).second) {
                reg.kids.push_back({"contig_start", w->fast->scalar_i64(cs)});
                reg.kids.push_back({"contig_end", w->fast->scalar_i64(ce)});
                reg.dirty = true;
            }
            if (w->images.insert(contig + prefix + suffix).second) {
                // a window's three datasets and their group are one stamped block (h5emit.h): the block's bytes, then the
                // data -- int64 -> uint32 positions (a -1 padding row wraps to 4294967295, DataStore.py:128), bases, rles
                uint64_t addr, header;
                uint8_t* block = w->fast->reserve_block(window_stamp(), &addr, &header);
                w->pool->add({block, addr, positions + (size_t)i * kSeq * 3, bases + (size_t)i * kSeq, rles + (size_t)i * kSeq});
                w->fast->stamped();          // (a full buffer is handed on: its blocks are filled first, before_flush)
                reg.kids.push_back({suffix, header});
                reg.dirty = true;
            }
            if (!w->fast->ok()) return fail("write failed (disk full?)");
            continue;
        }
        std::lock_guard<std::recursive_mutex> lib(g_library_mutex);
        const std::string root = "predictions/" + contig + "/" + prefix;
        if (w->regions.insert(prefix).second) {
            if (write_ds(w, w->file, root + "/contig_start", H5T_STD_I64LE, H5T_NATIVE_INT64, w->space_scalar, &cs)) return -1;
            if (write_ds(w, w->file, root + "/contig_end", H5T_STD_I64LE, H5T_NATIVE_INT64, w->space_scalar, &ce)) return -1;
        }
        if (w->images.insert(contig + prefix + suffix).second) {
            const int64_t* p = positions + (size_t)i * kSeq * 3;
            for (int k = 0; k < kSeq * 3; ++k) w->pos32[k] = (uint32_t)p[k];
            const std::string base = root + "/" + suffix;
            hid_t g = H5Gcreate2(w->file, base.c_str(), w->lcpl, H5P_DEFAULT, H5P_DEFAULT);
            if (g < 0) return fail("cannot create group '%s'", base.c_str());
            int bad = write_ds(w, g, "position", H5T_STD_U32LE, H5T_NATIVE_UINT32, w->space_pos, w->pos32.data()) ||
                      write_ds(w, g, "bases", H5T_STD_U8LE, H5T_NATIVE_UINT8, w->space_lab, bases + (size_t)i * kSeq) ||
                      write_ds(w, g, "rles", H5T_STD_U8LE, H5T_NATIVE_UINT8, w->space_lab, rles + (size_t)i * kSeq);
            H5Gclose(g);
            if (bad) return -1;
        }
    }
    if (w->fast) {
        w->pool->run();              // the caller's arrays are only good until this call returns
        if (!w->fast->ok()) return fail("write failed (disk full?)");
    }
    return 0;
}

int helen_io_write_predictions(void* handle, int n, const char* contigs, const int64_t* meta,
                               const int64_t* positions, const uint8_t* bases, const uint8_t* rles) {
    return helen_io_write_predictions_sel(handle, n, nullptr, contigs, meta, positions, bases, rles);
}

/* Sequence of one region of a prediction file, as Stitch.small_chunk_stitch builds it
 * (helen/modules/python/Stitch.py:204-247): the region's chunk ids are visited in STRING-sorted order;
 * every (pos, indx, split) position key keeps its first writer; keys are sorted numerically and each
 * contributes label_decoder[base] repeated rle times ('' A C G T for 0..4, Options.py:3).  Positions
 * are stored as uint32, so the (-1,-1,-1) rows of padded images arrive as 4294967295 and are NOT
 * skipped by the reference's `< 0` test: all of them share one key, which sorts last -- reproduced.
 * Writes the NUL-terminated sequence into out (capacity cap); returns its length, -2 if cap is too
 * small (call again with a larger buffer), -1 on error. */
struct Rec {
    int64_t pos, indx, split;
    uint8_t base, rle;
    uint32_t order;
};

// One region's records through the scanner (prediction files written by this package's emitter, by libhdf5 with
// default settings, or by h5py: old-style groups, contiguous datasets).  0: done; 1: not for the scanner.
static int fast_region_records(Scanned* sc, const char* contig, const char* region, std::vector<Rec>* recs) {
    const h5scan::File& f = sc->file;
    uint64_t g;
    if (!f.lookup(f.root(), "predictions", &g) || !f.lookup(g, contig, &g) || !f.lookup(g, region, &g)) return 1;
    std::vector<std::pair<std::string, uint64_t>> kids;
    if (!f.children(g, &kids)) return 1;
    // chunk ids in STRING order (sorted(set of str)): the B-tree order of an old-style group is exactly that
    uint32_t order = 0;
    std::vector<int64_t> pos;
    std::vector<uint8_t> bases, rles;
    for (auto& kv : kids) {
        if (kv.first == "contig_start" || kv.first == "contig_end") continue;
        uint64_t h;
        h5scan::Dataset db, dr, dp;
        if (!f.lookup(kv.second, "bases", &h) || !f.dataset(h, &db)) return 1;
        if (!f.lookup(kv.second, "rles", &h) || !f.dataset(h, &dr)) return 1;
        if (!f.lookup(kv.second, "position", &h) || !f.dataset(h, &dp)) return 1;
        const uint64_t n = db.count();
        if (db.cls != 0 || dr.cls != 0 || dp.cls != 0 || n == 0 || n > (1u << 24) || dr.count() != n ||
            dp.count() != 3 * n)
            return 1;
        pos.resize((size_t)n * 3);
        bases.resize((size_t)n);
        rles.resize((size_t)n);
        if (!scan_2d<uint8_t>(db, bases.data()) || !scan_2d<uint8_t>(dr, rles.data()) ||
            !scan_2d<int64_t>(dp, pos.data()))
            return 1;
        // values as the file's own type holds them: the reference's `indx < 0 or pos < 0` test never fires on the
        // uint32 its DataStore writes (wrapped -1 padding included) and does on a signed file of another writer
        for (uint64_t k = 0; k < n; ++k) {
            if (pos[(size_t)k * 3] < 0 || pos[(size_t)k * 3 + 1] < 0) continue;
            recs->push_back({pos[(size_t)k * 3], pos[(size_t)k * 3 + 1], pos[(size_t)k * 3 + 2], bases[(size_t)k],
                             rles[(size_t)k], order++});
        }
    }
    return 0;
}

static long long decode_records(std::vector<Rec>& recs, char* out, long long cap);

/* The regions of predictions/<contig> of one file, in name order (what sorted(h5py keys) yields), with the
 * contig_start / contig_end each of them stores (StitchInterface.py:84-95 reads these two scalars per region:
 * 300 k regions x 2 dataset opens through a Python binding is minutes).  Two calls: with names == NULL only
 * sizes[0] = number of regions and sizes[1] = bytes of the '\n'-joined names are set; then names (capacity
 * sizes[1] + 1), starts and ends (sizes[0] entries each) are filled.  Returns 0, 1 if the file has no such
 * contig, -1 on error. */
int helen_io_list_regions(const char* path, const char* contig, long long* sizes, char* names, int64_t* starts,
                          int64_t* ends) {
    std::vector<std::string> region;
    std::vector<int64_t> st, en;
    bool listed = false;
    if (Scanned* sc = scan_file(path)) {
        const h5scan::File& f = sc->file;
        uint64_t g;
        std::vector<std::pair<std::string, uint64_t>> kids;
        if (f.lookup(f.root(), "predictions", &g)) {
            if (!f.lookup(g, contig, &g)) {
                // absent, or not an old-style group: only the first is an answer
                std::vector<std::pair<std::string, uint64_t>> top;
                uint64_t pg;
                if (f.lookup(f.root(), "predictions", &pg) && f.children(pg, &top)) return 1;
            } else if (f.children(g, &kids)) {
                listed = true;
                for (auto& kv : kids) {
                    int64_t a, b;
                    if (!scan_i64_first(f, kv.second, "contig_start", &a) || !scan_i64_first(f, kv.second, "contig_end", &b)) {
                        listed = false;
                        break;
                    }
                    region.push_back(kv.first);
                    st.push_back(a);
                    en.push_back(b);
                }
            }
        }
    }
    if (!listed) {
        static const bool direct_only = reader_mode_is("direct");
        if (direct_only) return fail("%s: not a file the direct scanner takes", path);
        region.clear();
        st.clear();
        en.clear();
        std::lock_guard<std::recursive_mutex> lib(g_library_mutex);
        hid_t f = get_file(path);
        if (f < 0) return fail("cannot open '%s'", path);
        const std::string gpath = std::string("predictions/") + contig;
        const htri_t has_p = H5Lexists(f, "predictions", H5P_DEFAULT);
        const htri_t has_c = has_p > 0 ? H5Lexists(f, gpath.c_str(), H5P_DEFAULT) : has_p;
        if (has_p < 0 || has_c < 0) return fail("%s: cannot look up '%s'", path, gpath.c_str());   // damaged, not absent
        if (has_c == 0) return 1;
        hid_t g = H5Gopen2(f, gpath.c_str(), H5P_DEFAULT);
        if (g < 0) return fail("%s: cannot open group '%s'", path, gpath.c_str());
        auto cb = [](hid_t, const char* name, const H5L_info_t*, void* ud) -> herr_t {
            ((std::vector<std::string>*)ud)->push_back(name);
            return 0;
        };
        const herr_t it = H5Literate(g, H5_INDEX_NAME, H5_ITER_INC, nullptr, cb, &region);
        bool good = it >= 0;
        for (size_t k = 0; good && k < region.size(); ++k) {
            hid_t rg = H5Gopen2(g, region[k].c_str(), H5P_DEFAULT);
            int64_t a = 0, b = 0;
            good = rg >= 0 && read_i64_first(rg, "contig_start", &a) == 0 && read_i64_first(rg, "contig_end", &b) == 0;
            if (rg >= 0) H5Gclose(rg);
            st.push_back(a);
            en.push_back(b);
        }
        H5Gclose(g);
        if (!good) return fail("%s: cannot list the regions of '%s'", path, gpath.c_str());
    }
    size_t bytes = 0;
    for (auto& r : region) bytes += r.size() + 1;
    if (!names) {
        sizes[0] = (long long)region.size();
        sizes[1] = (long long)bytes;
        return 0;
    }
    if (sizes[0] != (long long)region.size() || sizes[1] < (long long)bytes) return fail("%s: changed between two calls", path);
    size_t o = 0;
    for (size_t k = 0; k < region.size(); ++k) {
        memcpy(names + o, region[k].data(), region[k].size());
        o += region[k].size();
        names[o++] = '\n';
        starts[k] = st[k];
        ends[k] = en[k];
    }
    names[o ? o - 1 : 0] = 0;
    return 0;
}

long long helen_io_region_sequence(const char* path, const char* contig, const char* region, char* out,
                                   long long cap) {
    std::vector<Rec> recs;
    if (Scanned* sc = scan_file(path)) {
        if (fast_region_records(sc, contig, region, &recs) == 0) return decode_records(recs, out, cap);
        recs.clear();
    }
    static const bool direct_only = reader_mode_is("direct");
    if (direct_only) return fail("%s: not a file the direct scanner takes", path);
    std::lock_guard<std::recursive_mutex> lib(g_library_mutex);
    hid_t f = get_file(path);
    if (f < 0) return fail("cannot open '%s'", path);
    const std::string gpath = std::string("predictions/") + contig + "/" + region;
    hid_t g = H5Gopen2(f, gpath.c_str(), H5P_DEFAULT);
    if (g < 0) return fail("%s: no region '%s'", path, gpath.c_str());
    H5G_info_t info;
    H5Gget_info(g, &info);
    std::vector<std::string> chunks;
    std::vector<char> name(256);
    for (hsize_t i = 0; i < info.nlinks; ++i) {
        ssize_t n = H5Lget_name_by_idx(g, ".", H5_INDEX_NAME, H5_ITER_INC, i, name.data(), name.size(), H5P_DEFAULT);
        if (n < 0 || (size_t)n >= name.size()) continue;
        std::string s(name.data());
        if (s != "contig_start" && s != "contig_end") chunks.push_back(s);
    }
    std::sort(chunks.begin(), chunks.end());   // sorted(set of str): lexicographic
    std::vector<int64_t> pos;
    std::vector<uint8_t> bases, rles;
    uint32_t order = 0;
    for (const std::string& c : chunks) {
        hid_t cg = H5Gopen2(g, c.c_str(), H5P_DEFAULT);
        if (cg < 0) continue;
        hid_t d = H5Dopen2(cg, "bases", H5P_DEFAULT);
        hssize_t n = 0;
        if (d >= 0) {
            hid_t sp = H5Dget_space(d);
            n = H5Sget_simple_extent_npoints(sp);
            H5Sclose(sp);
        }
        bool ok = d >= 0 && n > 0;
        if (ok) {
            pos.resize((size_t)n * 3);
            bases.resize((size_t)n);
            rles.resize((size_t)n);
            ok = H5Dread(d, H5T_NATIVE_UINT8, H5S_ALL, H5S_ALL, H5P_DEFAULT, bases.data()) >= 0;
        }
        if (d >= 0) H5Dclose(d);
        if (ok) {
            hid_t dr = H5Dopen2(cg, "rles", H5P_DEFAULT);
            hid_t dp = H5Dopen2(cg, "position", H5P_DEFAULT);
            ok = dr >= 0 && dp >= 0 &&
                 H5Dread(dr, H5T_NATIVE_UINT8, H5S_ALL, H5S_ALL, H5P_DEFAULT, rles.data()) >= 0 &&
                 H5Dread(dp, H5T_NATIVE_INT64, H5S_ALL, H5S_ALL, H5P_DEFAULT, pos.data()) >= 0;
            if (dr >= 0) H5Dclose(dr);
            if (dp >= 0) H5Dclose(dp);
        }
        H5Gclose(cg);
        if (!ok) {
            H5Gclose(g);
            return fail("%s: cannot read chunk '%s/%s'", path, gpath.c_str(), c.c_str());
        }
        for (hssize_t k = 0; k < n; ++k) {
            if (pos[(size_t)k * 3] < 0 || pos[(size_t)k * 3 + 1] < 0) continue;   // Stitch.py:226 (see above)
            recs.push_back({pos[(size_t)k * 3], pos[(size_t)k * 3 + 1], pos[(size_t)k * 3 + 2], bases[(size_t)k],
                            rles[(size_t)k], order++});
        }
    }
    H5Gclose(g);
    return decode_records(recs, out, cap);
}

// first writer wins per (pos, indx, split) key, keys in numeric order, base x run length
static long long decode_records(std::vector<Rec>& recs, char* out, long long cap) {
    auto before = [](const Rec& a, const Rec& b) {
        if (a.pos != b.pos) return a.pos < b.pos;
        if (a.indx != b.indx) return a.indx < b.indx;
        if (a.split != b.split) return a.split < b.split;
        return a.order < b.order;
    };
    // (an image's rows come in key order: a region of one chunk id needs no sorting at all)
    if (!std::is_sorted(recs.begin(), recs.end(), before)) std::sort(recs.begin(), recs.end(), before);
    static const char kDecode[5] = {0, 'A', 'C', 'G', 'T'};
    long long len = 0;
    for (size_t k = 0; k < recs.size(); ++k) {
        if (k > 0 && recs[k].pos == recs[k - 1].pos && recs[k].indx == recs[k - 1].indx &&
            recs[k].split == recs[k - 1].split)
            continue;   // first writer wins
        const char ch = recs[k].base < 5 ? kDecode[recs[k].base] : 0;
        if (!ch) continue;
        for (int r = 0; r < recs[k].rle; ++r) {
            if (len < cap - 1) out[len] = ch;
            ++len;
        }
    }
    if (len >= cap) {
        fail("buffer too small: need %lld", len + 1);
        return -2;
    }
    out[len] = 0;
    return len;
}

/* helen_io_region_sequence for regions whose images are still in memory (the labels a device call has just delivered:
 * helen_amd.stitch_stream decodes a region as soon as its last image has been written, beside the device stage).
 * Region r is made of the windows rows[first[r] .. first[r + 1]) of the arrays -- the caller lists them in the STRING
 * order of their chunk ids, each chunk id once (what the prediction file holds, Stitch.py:211 and DataStore.py:123).  The
 * position values are taken as the prediction file stores them: uint32 (a -1 padding row wraps to 4294967295 and
 * is a key like any other, Stitch.py:226 never skips it).  Sequences go to `out` back to back, offsets[r] .. offsets[r + 1];
 * `threads` threads of this call share the regions.  Returns the total length, -2 if `cap` is too small. */
long long helen_io_decode_regions(int n_regions, const int32_t* first, const int32_t* rows, const int64_t* positions,
                                  const uint8_t* bases, const uint8_t* rles, int threads, char* out, long long cap,
                                  int64_t* offsets) {
    if (n_regions <= 0) {
        if (offsets) offsets[0] = 0;
        return 0;
    }
    // An image's rows come in key order and a region's images follow each other on the contig, so a region is built by
    // APPENDING image after image; only where two images share rows (or a writer interleaved them) is there a merge,
    // and only an image that is not in key order itself is sorted.  Keys as the file stores them: three uint32.
    struct Item {
        uint32_t pos, indx, split;
        uint8_t base, rle;
    };
    auto less = [](const Item& a, const Item& b) {
        if (a.pos != b.pos) return a.pos < b.pos;
        if (a.indx != b.indx) return a.indx < b.indx;
        return a.split < b.split;
    };
    auto same = [](const Item& a, const Item& b) { return a.pos == b.pos && a.indx == b.indx && a.split == b.split; };
    std::vector<std::string> seqs((size_t)n_regions);
    std::atomic<int> next{0};
    std::atomic<bool> bad{false};
    auto work = [&]() {
        std::vector<Item> acc, img, merged;
        std::string local;
        static const char kDecode[5] = {0, 'A', 'C', 'G', 'T'};
        for (;;) {
            const int r = next.fetch_add(1);
            if (r >= n_regions) return;
            acc.clear();
            for (int k = first[r]; k < first[r + 1]; ++k) {
                const size_t w = (size_t)rows[k];
                const int64_t* p = positions + w * kSeq * 3;
                const uint8_t* b = bases + w * kSeq;
                const uint8_t* l = rles + w * kSeq;
                img.resize(kSeq);
                bool ordered = true;
                for (int i = 0; i < kSeq; ++i) {
                    img[i] = {(uint32_t)p[3 * i], (uint32_t)p[3 * i + 1], (uint32_t)p[3 * i + 2], b[i], l[i]};
                    if (i > 0 && less(img[i], img[i - 1])) ordered = false;
                }
                if (!ordered) std::stable_sort(img.begin(), img.end(), less);     // (equal keys keep their row order)
                if (acc.empty() || less(acc.back(), img.front())) {
                    acc.insert(acc.end(), img.begin(), img.end());
                } else {                                                           // ties: the earlier image first
                    merged.resize(acc.size() + img.size());
                    std::merge(acc.begin(), acc.end(), img.begin(), img.end(), merged.begin(), less);
                    acc.swap(merged);
                }
            }
            // (built in this thread's own string and moved into place: neighbouring elements of `seqs` share cache lines,
            // and a string that grows there is written to by every append)
            size_t need = 0;
            for (const Item& x : acc) need += x.rle;
            local.resize(need);
            char* at = &local[0];
            for (size_t k = 0; k < acc.size(); ++k) {
                if (k > 0 && same(acc[k], acc[k - 1])) continue;                    // first writer wins
                const char ch = acc[k].base < 5 ? kDecode[acc[k].base] : 0;
                if (!ch) continue;
                for (int c = acc[k].rle; c > 0; --c) *at++ = ch;
            }
            local.resize((size_t)(at - local.data()));
            seqs[(size_t)r] = local;
        }
    };
    const int T = std::max(1, std::min(threads, n_regions));
    std::vector<std::thread> pool;
    for (int t = 1; t < T; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (bad) return fail("region decode failed");
    long long total = 0;
    for (auto& s : seqs) total += (long long)s.size();
    if (total > cap) {
        fail("buffer too small: need %lld", total);
        return -2;
    }
    long long at = 0;
    for (int r = 0; r < n_regions; ++r) {
        offsets[r] = at;
        memcpy(out + at, seqs[(size_t)r].data(), seqs[(size_t)r].size());
        at += (long long)seqs[(size_t)r].size();
    }
    offsets[n_regions] = at;
    return total;
}

/* The overlap alignments of `n` joins in one call (Stitch.py:104-134 per join): join k aligns query
 * blob[r_off[k] .. + r_len[k]) against reference blob[l_off[k] .. + l_len[k]) as helen_ssw_align does and reduces the
 * result to what alignment_stitch uses: out[3 k] = best score, out[3 k + 1], out[3 k + 2] = (reference index, query
 * index) of the first M run (= and X merged) of at least `min_run`, or (-1, -1) (`get_confident_positions`,
 * Stitch.py:34-94).  An empty side gives score 0.  Single-threaded and re-entrant: callers run several at once. */
int helen_ssw_join_batch(int n, const char* blob, const int64_t* l_off, const int32_t* l_len, const int64_t* r_off,
                         const int32_t* r_len, int match, int mismatch, int gap_open, int gap_extend, int min_run,
                         int32_t* out) {
    std::vector<char> cigar;
    for (int k = 0; k < n; ++k) {
        int32_t* o = out + 3 * (size_t)k;
        o[0] = 0;
        o[1] = o[2] = -1;
        if (l_len[k] <= 0 || r_len[k] <= 0) continue;
        cigar.resize((size_t)16 * ((size_t)l_len[k] + (size_t)r_len[k]) + 64);
        int res[6];
        const int rc = helen_ssw_align(blob + l_off[k], l_len[k], blob + r_off[k], r_len[k], match, mismatch, gap_open,
                                       gap_extend, res, cigar.data(), (int)cigar.size());
        if (rc < 0) return fail("join %d: the aligner failed", k);
        o[0] = res[0];
        if (rc != 0 || res[0] == 0) continue;
        // runs of the extended CIGAR with = and X merged into M
        long long ref_index = res[1], read_index = 0, run = 0;
        char run_op = 0;
        bool found = false;
        auto close_run = [&]() {          // the run that just ended (or the last one)
            if (!run_op || found) return;
            if (run_op == 'M' && run >= min_run) {
                o[1] = (int32_t)ref_index;
                o[2] = (int32_t)read_index;
                found = true;
                return;
            }
            if (run_op == 'S' || run_op == 'I') read_index += run;
            else if (run_op == 'D') ref_index += run;
            else if (run_op == 'M') { ref_index += run; read_index += run; }
        };
        const char* c = cigar.data();
        while (*c && !found) {
            long long len = 0;
            while (*c >= '0' && *c <= '9') len = len * 10 + (*c++ - '0');
            char op = *c ? *c++ : 0;
            if (op == '=' || op == 'X') op = 'M';
            if (op == run_op) {
                run += len;
            } else {
                close_run();
                run_op = op;
                run = len;
            }
        }
        close_run();
    }
    return 0;
}

int helen_io_writer_close(void* handle) {
    Writer* w = (Writer*)handle;
    if (!w) return 0;
    if (w->fast) {
        // now every name is known: region groups, contig groups, `predictions`, the root group, the superblock.
        // A file that received no window at all gets an empty root group -- no `predictions` member, which is what
        // the reference's DataStore leaves behind in that case (and what stitch reports as an invalid file).
        // `predictions/{contig}/{contig}-{start}-{end}/...` is an HDF5 PATH for the reference's h5py writer and for
        // libhdf5 with intermediate-group creation (DataStore.py:117-133): a contig named 'a/b' makes groups
        // a -> b -> a -> 'b-0-1000', empty components and '.' vanish.  The same tree here: every region's full path
        // split into a trie of groups, the region's group attached under its last component.
        struct Node {
            std::map<std::string, Node> sub;
            std::vector<h5emit::Child> regions;
            const std::map<std::string, Region>* table = nullptr;    // an ordinary contig: its regions as the writer holds them
        };
        Node rootn;
        std::vector<std::string> parts;
        auto plain = [](const std::string& s) { return !s.empty() && s != "." && s.find('/') == std::string::npos; };
        for (auto& c : w->tree) {
            // (the ordinary contig name -- no '/', not empty, not '.' -- is one component, and so are its regions' names:
            // the group is written straight from the writer's table, which is in name order because the map is)
            bool direct = plain(c.first) && rootn.sub.find(c.first) == rootn.sub.end();
            for (auto& r : c.second) {
                settle_region(w, &r.second);
                direct = direct && plain(r.first);
            }
            if (direct) {
                rootn.sub[c.first].table = &c.second;
                continue;
            }
            for (auto& r : c.second) {
                const std::string full = c.first + "/" + r.first;
                parts.clear();
                for (size_t b = 0; b <= full.size();) {
                    size_t e = full.find('/', b);
                    if (e == std::string::npos) e = full.size();
                    const std::string part = full.substr(b, e - b);
                    if (!part.empty() && part != ".") parts.push_back(part);
                    b = e + 1;
                }
                Node* n = &rootn;
                for (size_t k = 0; k + 1 < parts.size(); ++k) n = &n->sub[parts[k]];
                if (n->table) {                 // a later contig puts something under an ordinary one ('a' and 'a/b'): spell it out
                    for (auto& r2 : *n->table) n->regions.push_back({r2.first, r2.second.header});
                    n->table = nullptr;
                }
                n->regions.push_back({parts.empty() ? std::string(".") : parts.back(), r.second.header});
            }
        }
        bool clash = false;
        std::function<uint64_t(Node&)> emit = [&](Node& n) -> uint64_t {
            if (n.table && n.sub.empty() && n.regions.empty()) {
                std::vector<const std::pair<const std::string, Region>*> rows;
                rows.reserve(n.table->size());
                for (auto& r : *n.table) rows.push_back(&r);
                return w->fast->group_sorted(rows.size(), [&](size_t i) -> const std::string& { return rows[i]->first; },
                                             [&](size_t i) { return rows[i]->second.header; });
            }
            std::vector<h5emit::Child> kids;
            kids.swap(n.regions);
            if (n.table)
                for (auto& r2 : *n.table) kids.push_back({r2.first, r2.second.header});
            if (!n.sub.empty()) {
                std::set<std::string> names;
                for (auto& k : kids) names.insert(k.name);
                for (auto& kv : n.sub) {
                    if (!names.insert(kv.first).second) clash = true;   // a region and a contig component of one name
                    kids.push_back({kv.first, emit(kv.second)});
                }
            }
            return w->fast->group(kids);
        };
        std::vector<h5emit::Child> top;
        if (!w->tree.empty()) top.push_back({"predictions", emit(rootn)});
        uint64_t bt = 0, hp = 0;
        const uint64_t root = w->fast->group(top, &bt, &hp);
        const bool good = w->fast->finish(root, bt, hp);
        forget_path(w->path.c_str());
        // the file is complete and closed; what is left is giving back a million small allocations (a 300,000-region
        // run: two name sets and the region table, ~0.1 s): not on the caller's clock
        delete w->pool;              // (its helpers are idle: every call filled what it reserved)
        w->pool = nullptr;
        std::thread([w]() {
            delete w->fast;
            delete w;
        }).detach();
        if (clash) return fail("a contig name component equals a region name of the same group: the prediction file is ambiguous");
        return good ? 0 : fail("writing the prediction file failed");
    }
    H5Sclose(w->space_pos);
    H5Sclose(w->space_lab);
    H5Sclose(w->space_scalar);
    H5Pclose(w->lcpl);
    H5Pclose(w->dcpl);
    const herr_t rc = H5Fclose(w->file);
    delete w;
    return rc < 0 ? fail("closing the prediction file failed") : 0;
}

}  // extern "C"
