/*
 * helen_io.h -- C ABI of libhelen_io.so, the host-side companion of libhelen_hip.so: the HDF5 reader and writer of
 * the `call_consensus` path and the two native pieces of `stitch`.
 *
 * Plain C++ on the host (no GPU, no torch).  Files are walked directly where their layout allows it -- MarginPolish
 * image files and prediction files as the HDF5 C library / h5py write them by default (h5scan.h), prediction files
 * emitted byte by byte (h5emit.h) -- with libhdf5 behind both for anything else.  Citations are file:line into the
 * reference tree.
 *
 * Conventions
 *   - functions returning int give 0 on success, -1 on error (helen_io_last_error() then describes it; the string
 *     is thread-local), and the small positive codes documented per function; nothing throws across the ABI;
 *   - all buffers are the caller's, C-contiguous; strings are NUL-terminated UTF-8;
 *   - the file functions (helen_io_*) keep per-process caches of open files and mappings behind a lock; the READER
 *     functions may be called from any number of threads (the direct scanner runs in parallel, whatever goes through
 *     libhdf5 -- not built thread-safe -- is serialised; helen_io_read_image_runs starts its own threads); a WRITER handle
 *     belongs to one thread at a time; helen_ssw_align is re-entrant;
 *   - HELEN_IO_SEQ (1000) positions and HELEN_IO_FEATURES (90) features per window (`Options.py:13-21`);
 *     HELEN_IO_NAME (256) bytes per contig name slot.
 */
#ifndef HELEN_IO_H
#define HELEN_IO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HELEN_IO_ABI_VERSION 1
#define HELEN_IO_SEQ 1000
#define HELEN_IO_FEATURES 90
#define HELEN_IO_NAME 256

int helen_io_abi_version(void);
const char* helen_io_last_error(void);

/* ---- reader: `SequenceDataset` (`models/dataloader_predict.py:18-95`) ------------------------------------------ */

/* Names of the members of group `images` of one file in name order -- what h5py's .keys() yields (:38-52) --
 * '\n'-separated into `out` (capacity `cap`); *n_out = their number.  Returns 1 (and *n_out = 0) if the file has no
 * `images` group (the reference warns and skips it, :47-49), -2 if `cap` is too small (*n_out = bytes needed). */
int helen_io_list_images(const char* path, char* out, size_t cap, long long* n_out);

/* `__getitem__` (:54-88) for `n` images of one file (names '\n'-separated): image -> uint8, position -> int64,
 * short images padded with zero rows / (-1,-1,-1) rows (:74-82); an image that is not [<= 1000, 90] with a
 * [same, 3] position is the reference's "IMAGE SIZE ERROR" (:85-86; the message starts with that text).
 *   images    [n, 1000, 90] uint8       positions [n, 1000, 3] int64
 *   meta      [n, 3] int64 = contig_start, contig_end, feature_chunk_idx        contigs [n, 256] char */
int helen_io_read_images(const char* path, const char* names, int n, uint8_t* images, int64_t* positions,
                         int64_t* meta, char* contigs);

/* The same reader addressed by POSITION in the file's name-ordered image list (what the loader's index is,
 * `dataloader_predict.py:38-52`): no names cross the boundary and the scanner needs no look-up per image.
 *   helen_io_index_images     *n_out = images of the file, *through_library = 1 if libhdf5 has to read them (the direct
 *                             scanner does not take the file's storage); returns 1 if there is no `images` group
 *   helen_io_image_names      names of images [first, first + count), '\n'-separated; -2: cap too small (*needed)
 *   helen_io_read_image_range helen_io_read_images for images [first, first + count); *through_library += images
 *                             libhdf5 read
 *   helen_io_read_image_runs  n_runs ranges (paths[i], firsts[i], counts[i]) into consecutive rows of the arrays, read
 *                             by `threads` threads of this call
 *   helen_io_forget_images    drop the index and the mapping of one file (the reader has moved on) */
int helen_io_index_images(const char* path, long long* n_out, int* through_library);
int helen_io_image_names(const char* path, long long first, long long count, char* out, size_t cap, long long* needed);
int helen_io_read_image_range(const char* path, long long first, int count, uint8_t* images, int64_t* positions,
                              int64_t* meta, char* contigs, long long* through_library);
int helen_io_read_image_runs(int n_runs, const char* const* paths, const long long* firsts, const int* counts,
                             int threads, uint8_t* images, int64_t* positions, int64_t* meta, char* contigs,
                             long long* through_library);
void helen_io_forget_images(const char* path);
/* How a file's images are stored, judged by its first image: out[0] = 0 the direct scanner reads them, 1 libhdf5 has to;
 * out[1] = layout class of `image` (0 compact, 1 contiguous, 2 chunked; -1 not inspected); out[2] = number of filters;
 * out[3] = 1 if deflate is among them.  Returns 1 if the file has no images.  (helen_amd.host_plan prices a run with it.) */
int helen_io_image_storage(const char* path, int* out);
/* 1 when deflated chunks are inflated through libdeflate (found at run time; 2-3 x zlib's rate), 0 when through zlib. */
int helen_io_fast_inflate(void);

/* The loader of `helen_train test` (`models/dataloader.py:48-61`): image uint8 [1000, 90], label_base and
 * label_run_length uint8 [1000] of `n` images of one file, exactly as stored (that loader does not pad: any other
 * shape is an "IMAGE SIZE ERROR").  images [n, 1000, 90]; label_base, label_rle [n, 1000]. */
int helen_io_read_labeled(const char* path, const char* names, int n, uint8_t* images, uint8_t* label_base,
                          uint8_t* label_rle);

/* libhdf5 is not assumed thread-safe: every call this library makes into it is under one process-wide recursive lock.
 * A thread that uses libhdf5 through another binding beside this library takes the same lock for the duration
 * (lock and unlock on the same thread). */
void helen_io_library_lock(void);
void helen_io_library_unlock(void);

/* out[0] / out[1] = images this process has read through the direct scanner / through libhdf5. */
void helen_io_reader_counts(long long* out);
/* Drop every cached file handle and mapping of this process. */
void helen_io_close_readers(void);

/* Benchmark inputs: a MarginPolish-shaped image file through the direct emitter (`n` windows named
 * <contig>-<start>-<end>-<chunk>, the six datasets of :64-70).  starts, chunks int64 [n]; lengths int32 [n] (rows
 * stored, <= 1000); images uint8 [n, 1000, 90]. */
int helen_io_emit_images(const char* path, int n, const char* contig, const int64_t* starts, const int64_t* chunks,
                         const int32_t* lengths, const uint8_t* images);

/* The same with per-window contig names (char [n, 256]), contig_end and position rows (int64 [n, 1000, 3], the first
 * lengths[i] rows stored): the simulated assemblies of helen_amd.synthetic.write_assembly_dir. */
int helen_io_emit_image_windows(const char* path, int n, const char* contigs, const int64_t* starts, const int64_t* ends,
                                const int64_t* chunks, const int32_t* lengths, const uint8_t* images,
                                const int64_t* positions);

/* ---- writer: `DataStore.write_prediction` (`DataStore.py:83-133`) ---------------------------------------------- */

/* DataStore(filename, 'w') (`predict_gpu.py:55`); NULL on error. */
void* helen_io_writer_open(const char* path);
/* `n` windows: scalar int64 contig_start / contig_end once per region predictions/<contig>/<contig-start-end>
 * (:115-120), then position uint32 [1000, 3] (-1 wraps to 4294967295), bases uint8 [1000], rles uint8 [1000] once per
 * (region, chunk id) (:123-133); repeats are skipped silently.
 *   contigs [n, 256] char; meta [n, 3] int64 = contig_start, contig_end, chunk id; positions [n, 1000, 3] int64;
 *   bases, rles [n, 1000] uint8.  `_sel`: only the rows sel[0..n_sel) of those arrays. */
int helen_io_write_predictions(void* writer, int n, const char* contigs, const int64_t* meta,
                               const int64_t* positions, const uint8_t* bases, const uint8_t* rles);
int helen_io_write_predictions_sel(void* writer, int n_sel, const int32_t* sel, const char* contigs,
                                   const int64_t* meta, const int64_t* positions, const uint8_t* bases,
                                   const uint8_t* rles);
/* Writes the group structures and closes the file; a failure here (disk full) is an error, not a short file. */
int helen_io_writer_close(void* writer);

/* ---- stitch (`Stitch.py:14-301`, `StitchInterface.py:40-106`) -------------------------------------------------- */

/* The regions of predictions/<contig> of one prediction file in name order with the contig_start / contig_end each
 * stores (`StitchInterface.py:84-95`).  Two calls: with names == NULL only sizes[0] = number of regions and
 * sizes[1] = bytes of the '\n'-joined names are set; then names (capacity sizes[1] + 1), starts and ends (sizes[0]
 * entries) are filled.  Returns 1 if the file has no such contig. */
int helen_io_list_regions(const char* path, const char* contig, long long* sizes, char* names, int64_t* starts,
                          int64_t* ends);

/* The sequence of one region as `small_chunk_stitch` builds it (`Stitch.py:204-247`): chunk ids in STRING order, the
 * first image to mention a (pos, indx, split) key wins, rows with a negative pos or indx are skipped (the uint32-wrapped
 * padding of the reference's writer is not negative), keys in numeric order, label_decoder[base] x run length.
 * Returns the length (NUL-terminated in `out`), -2 if `cap` is too small. */
long long helen_io_region_sequence(const char* path, const char* contig, const char* region, char* out,
                                   long long cap);

/* helen_io_region_sequence for regions whose images are still in memory (labels a device call has just delivered): region
 * r = the windows rows[first[r] .. first[r + 1]) of positions int64 [*, 1000, 3] / bases, rles uint8 [*, 1000], listed
 * by the caller in the STRING order of their chunk ids, each id once.  Position values are taken as the prediction
 * file stores them (uint32: a -1 padding row is the key 4294967295).  Sequences go to `out` back to back with
 * offsets[n_regions + 1]; `threads` threads of this call share the work.  Returns the total length, -2 if cap is short. */
long long helen_io_decode_regions(int n_regions, const int32_t* first, const int32_t* rows, const int64_t* positions,
                                  const uint8_t* bases, const uint8_t* rles, int threads, char* out, long long cap,
                                  int64_t* offsets);

/* `n` overlap alignments in one call (`Stitch.py:104-134` per join), reduced to what `alignment_stitch` uses: join k
 * aligns query blob[r_off[k], +r_len[k]) against reference blob[l_off[k], +l_len[k]) as helen_ssw_align does;
 * out[3k] = best score, out[3k+1], out[3k+2] = (reference index, query index) of the first M run (= and X merged)
 * of at least `min_run`, or (-1, -1) (`get_confident_positions`, `Stitch.py:34-94`).  Single-threaded, re-entrant. */
int helen_ssw_join_batch(int n, const char* blob, const int64_t* l_off, const int32_t* l_len, const int64_t* r_off,
                         const int32_t* r_len, int match, int mismatch, int gap_open, int gap_extend, int min_run,
                         int32_t* out);

/* HELEN.Aligner(match, mismatch, gap_open, gap_extend) + SetReferenceSequence(ref) + Align_cpp(query, Filter(), &al, 0)
 * (`Stitch.py:110-134`; native `ssw.c` / `ssw_cpp.cpp`): the same score, begin / end cells, extended CIGAR and
 * mismatch count as the striped Smith-Waterman library the reference vendors.
 *   out[6] = score, ref_begin, ref_end, query_begin, query_end, mismatches (0-based, inclusive)
 *   cigar  = extended CIGAR with soft clips, truncated to cigar_cap
 * Returns 1 if either sequence is empty (Align_cpp returns false). */
int helen_ssw_align(const char* ref_seq, int ref_len, const char* query_seq, int query_len, int match, int mismatch,
                    int gap_open, int gap_extend, int* out, char* cigar, int cigar_cap);

/* helen_ssw_align answers the common join of stitch -- the two overlap strings share one exact run as long as their longest
 * common subsequence, A/C/G/T only -- without running the three passes: for such a pair the library's result is determined
 * (helen_amd/csrc/ssw.cpp: exact_overlap states why), and it is what this returns; a pair that fails that test but whose
 * forward pass ends on an exact run of score / match bases skips the other two passes.  On by default.
 *   helen_ssw_fast_path(0 | 1)   switches it off / on for the process (any other value only asks); returns the previous setting
 *   helen_ssw_fast_path_counts   alignments answered that way / handed on to the three passes since the library was loaded */
int helen_ssw_fast_path(int enable);
void helen_ssw_fast_path_counts(long long* hits, long long* misses);
/* ... of the hits, those answered only after the forward pass: its best cell ends an exact run of score / match bases, which
 * fixes the begin cell and the CIGAR (the backward and the banded pass are not run). */
long long helen_ssw_fast_path_after_forward(void);

#ifdef __cplusplus
}
#endif
#endif /* HELEN_IO_H */
