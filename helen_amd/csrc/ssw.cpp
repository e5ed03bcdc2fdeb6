// ssw.cpp -- local alignment for `stitch`, written from scratch to give the SAME answers as the
// striped Smith-Waterman library the reference vendors and calls through its HELEN.Aligner binding
// (reference: helen/modules/src/local_reassembly/ssw.c `ssw_align`, ssw_cpp.cpp `Aligner::Align_cpp`,
// used at helen/modules/python/Stitch.py:111-134).
//
// Why not "any" Smith-Waterman: stitch anchors two overlapping sequences at the first long match
// run of the reported alignment, so ties must break the same way and the reference's quirks matter:
//   * the score / end cell come from a STRIPED pass (query split into L interleaved lanes, L = 16 for
//     the 8-bit pass, 8 for the 16-bit pass that is used once the 8-bit score saturates at 255) whose
//     E (gap in the query direction) is updated from H *before* the lazy-F correction, and whose best
//     cell is the first reference column that raises the maximum, smallest query index in it;
//   * the begin cell comes from the same pass run backwards from the end cell until a column
//     reaches the forward score;
//   * the CIGAR comes from a banded global-in-the-box DP (band doubled until the score is reached)
//     with a fixed preference order in its traceback (diagonal on ties, deletion over insertion).
// The lanes are emulated with plain loops so that the arithmetic, including its saturation behaviour, is the
// library's; the version stitch actually runs (striped_pass_small) keeps them in the compiler's vector types.
// Checked cell-for-cell against the reference library itself (oracle/_ref/libssw_ref.so, built from
// the reference's own sources) on randomised inputs: tests/test_stitch.py.
#include "../../include/helen_io.h"
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// A,C,G,T -> 0..3 (either case), everything else 4 -- the reference's translation table
// (ssw_cpp.cpp:9-19; note that U/u also map to 0 there).
int8_t base_code(char c) {
    switch (c) {
        case 'A': case 'a': case 'U': case 'u': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

struct Best {
    int score = 0, ref = 0, read = 0;
};

inline int sat_sub_u(int a, int b) { return a > b ? a - b : 0; }  // unsigned saturating subtract

// One striped pass over ref[begin..end) (forward) or backwards, query `read` of length m.
//   lanes 16: the 8-bit pass (scores biased by `bias`, saturating at 255);
//   lanes 8:  the 16-bit pass (signed saturating add).
// terminate < 0: never stop early.  Returns the best score, its reference index and query index.
Best striped_pass(const int8_t* ref, bool backwards, int n, const int8_t* read, int m, int gap_open,
                  int gap_ext, const int8_t* mat, int lanes, int bias, int terminate) {
    const int seg = (m + lanes - 1) / lanes;
    const int cells = seg * lanes;
    // profile[r][cell]: substitution score of ref base r against the query position of that cell;
    // cell index c = s * lanes + lane  <->  query position lane * seg + s; padding scores 0.
    std::vector<int> prof((size_t)5 * cells);
    for (int r = 0; r < 5; ++r)
        for (int s = 0; s < seg; ++s)
            for (int l = 0; l < lanes; ++l) {
                const int q = l * seg + s;
                prof[(size_t)r * cells + s * lanes + l] = q < m ? mat[r * 5 + read[q]] : 0;
            }
    std::vector<int> Hs(cells, 0), Hl(cells, 0), E(cells, 0), Hbest(cells, 0), F(lanes), Hv(lanes);
    const int cap = lanes == 16 ? 255 : 32767;
    int best = 0, best_ref = lanes == 16 ? -1 : 0, best_read = m - 1;
    bool overflow = false;

    auto shift_lanes = [&](std::vector<int>& v) {  // lane l takes lane l-1's value, lane 0 gets 0
        for (int l = lanes - 1; l > 0; --l) v[l] = v[l - 1];
        v[0] = 0;
    };

    for (int step = 0; step < n; ++step) {
        const int i = backwards ? n - 1 - step : step;
        const int* P = &prof[(size_t)ref[i] * cells];
        int col_max = 0;
        std::fill(F.begin(), F.end(), 0);
        for (int l = 0; l < lanes; ++l) Hv[l] = Hs[(seg - 1) * lanes + l];
        shift_lanes(Hv);  // diagonal predecessors of the first segment
        std::swap(Hs, Hl);
        for (int s = 0; s < seg; ++s) {
            for (int l = 0; l < lanes; ++l) {
                const int c = s * lanes + l;
                int h;
                if (lanes == 16) {
                    h = std::min(Hv[l] + P[c] + bias, 255);  // adds_epu8 with the biased profile
                    h = sat_sub_u(h, bias);
                } else {
                    h = std::min(Hv[l] + P[c], 32767);        // adds_epi16 (never underflows here)
                }
                h = std::max(h, E[c]);
                h = std::max(h, F[l]);
                col_max = std::max(col_max, h);
                Hs[c] = h;
                const int open = sat_sub_u(h, gap_open);
                E[c] = std::max(sat_sub_u(E[c], gap_ext), open);   // from H before the lazy-F fix-up
                F[l] = std::max(sat_sub_u(F[l], gap_ext), open);
                Hv[l] = Hl[c];
            }
        }
        // lazy F: carry gaps across lane boundaries; E is deliberately left alone
        if (lanes == 16) {
            int s = 0;
            shift_lanes(F);
            auto settled = [&](int seg_i) {
                for (int l = 0; l < lanes; ++l)
                    if (sat_sub_u(F[l], sat_sub_u(Hs[seg_i * lanes + l], gap_open)) != 0) return false;
                return true;
            };
            while (!settled(s)) {
                for (int l = 0; l < lanes; ++l) {
                    int& h = Hs[s * lanes + l];
                    h = std::max(h, F[l]);
                    col_max = std::max(col_max, h);
                    F[l] = sat_sub_u(F[l], gap_ext);
                }
                if (++s >= seg) {
                    s = 0;
                    shift_lanes(F);
                }
            }
        } else {
            bool done = false;
            for (int k = 0; k < lanes && !done; ++k) {
                shift_lanes(F);
                for (int s = 0; s < seg && !done; ++s) {
                    bool any = false;
                    for (int l = 0; l < lanes; ++l) {
                        int& h = Hs[s * lanes + l];
                        h = std::max(h, F[l]);
                        col_max = std::max(col_max, h);
                        F[l] = sat_sub_u(F[l], gap_ext);
                        if (F[l] > sat_sub_u(h, gap_open)) any = true;
                    }
                    if (!any) done = true;
                }
            }
        }
        if (col_max > best) {
            best = col_max;
            if (lanes == 16 && best + bias >= 255) {
                overflow = true;
                break;
            }
            best_ref = i;
            Hbest = Hs;
        }
        if (col_max == terminate) break;
    }
    for (int c = 0; c < cells; ++c)
        if (Hbest[c] == best) {
            const int q = c / lanes + (c % lanes) * seg;
            if (q < best_read) best_read = q;
        }
    Best b;
    b.score = (lanes == 16 && (overflow || best + bias >= 255)) ? 255 : std::min(best, cap);
    b.ref = best_ref;
    b.read = best_read;
    return b;
}

// The same pass for scores that stay clear of the 16-bit cap (every alignment stitch makes: a few hundred bases),
// on the compiler's vector types: one vector of LANES 16-bit values per segment, so the lane loops above become
// single SSE2 / NEON operations (pmaxsw, paddw, ...).  Cell for cell the same values (tests/test_stitch.py runs this
// against the reference library and against striped_pass).  With long match runs F stays above H - gap_open for
// many cells, and the lazy-F loop -- 40-50 segment visits per column against `seg` in the main loop -- is where the
// time goes: its "is any lane still open" test is one compare and an OR of the mask's halves.
// The caller guarantees  min(n, m) * max(mat) + max(mat) + bias < 32000.
template <int LANES>
struct Lanes;
#define HELEN_SSW_INLINE static inline __attribute__((always_inline))
#define HELEN_SSW_COMMON(N, T, M)                                                            \
    typedef T V __attribute__((vector_size(16)));                                            \
    typedef long long W __attribute__((vector_size(16)));                                    \
    HELEN_SSW_INLINE V splat(int x) {                                                        \
        V v;                                                                                 \
        for (int l = 0; l < N; ++l) v[l] = (T)x;                                             \
        return v;                                                                            \
    }                                                                                        \
    HELEN_SSW_INLINE V vmax(V a, V b) { return a > b ? a : b; }                              \
    HELEN_SSW_INLINE bool any_set(V v) {                                                     \
        const W w = (W)v;                                                                    \
        return (w[0] | w[1]) != 0;                                                           \
    }                                                                                        \
    HELEN_SSW_INLINE V shift_up(V v) { /* lane l takes lane l-1's value, lane 0 gets 0 */    \
        typedef M Mask __attribute__((vector_size(16)));                                     \
        Mask k;                                                                              \
        k[0] = 0;                                                                            \
        for (int l = 1; l < N; ++l) k[l] = (M)(N + l - 1);                                   \
        return __builtin_shuffle(splat(0), v, k);       /* one byte shift of the register */ \
    }                                                                                        \
    HELEN_SSW_INLINE int hmax(V v) {                                                         \
        int m = v[0];                                                                        \
        for (int l = 1; l < N; ++l) m = std::max<int>(m, v[l]);                              \
        return m;                                                                            \
    }
template <>
struct Lanes<8> {   // the 16-bit pass: signed words, nowhere near saturation (see the caller)
    HELEN_SSW_COMMON(8, int16_t, short)
    HELEN_SSW_INLINE V sub0(V a, V b) { return vmax(a - b, splat(0)); }
    HELEN_SSW_INLINE bool any_above(V a, V b) { return any_set((V)(a > b)); }
    HELEN_SSW_INLINE V add_profile(V h, V p, V) { return h + p; }
    static int profile_entry(int score, int) { return score; }
};
template <>
struct Lanes<16> {  // the 8-bit pass: unsigned bytes, saturating at 255; profile entries carry the bias
    HELEN_SSW_COMMON(16, uint8_t, unsigned char)
    HELEN_SSW_INLINE V sub0(V a, V b) { return vmax(a, b) - b; }                       // subs_epu8
    HELEN_SSW_INLINE bool any_above(V a, V b) { return any_set(sub0(a, b)); }
    HELEN_SSW_INLINE V add_profile(V h, V p, V bias) {
        V sum = h + p;
        sum |= (V)(sum < h);                                                             // adds_epu8: 255 on wrap
        return sub0(sum, bias);
    }
    static int profile_entry(int score, int bias) { return score + bias; }
};
#undef HELEN_SSW_COMMON
#undef HELEN_SSW_INLINE

template <int LANES>
Best striped_pass_small(const int8_t* ref, bool backwards, int n, const int8_t* read, int m, int gap_open,
                        int gap_ext, const int8_t* mat, int bias, int terminate) {
    typedef Lanes<LANES> L;
    typedef typename L::V V;
    const int seg = (m + LANES - 1) / LANES;
    thread_local std::vector<V> scratch;
    scratch.assign((size_t)9 * seg, L::splat(0));
    V* const prof = scratch.data();             // [5][seg]
    V* Hs = prof + 5 * seg;                     // H of the current column (store) ...
    V* Hl = Hs + seg;                           // ... and of the previous one (load)
    V* const E = Hl + seg;
    V* const Hbest = E + seg;
    constexpr bool kByte = LANES == 16;
    for (int r = 0; r < 5; ++r)
        for (int s = 0; s < seg; ++s)
            for (int l = 0; l < LANES; ++l) {
                const int q = l * seg + s;    // padding cells score 0 (the bias alone in the 8-bit pass)
                prof[(size_t)r * seg + s][l] = L::profile_entry(q < m ? mat[r * 5 + read[q]] : 0, bias);
            }
    const V zero = L::splat(0), go = L::splat(gap_open), ge = L::splat(gap_ext), bs = L::splat(bias);
    int best = 0, best_ref = kByte ? -1 : 0, best_read = m - 1;
    bool overflow = false;
    for (int step = 0; step < n; ++step) {
        const int i = backwards ? n - 1 - step : step;
        const V* P = prof + (size_t)ref[i] * seg;
        V F = zero, cm = zero;
        V Hv = L::shift_up(Hs[seg - 1]);        // diagonal predecessors of the first segment
        std::swap(Hs, Hl);
        for (int s = 0; s < seg; ++s) {
            V h = L::add_profile(Hv, P[s], bs);
            h = L::vmax(L::vmax(h, E[s]), F);
            cm = L::vmax(cm, h);
            Hs[s] = h;
            const V open = L::sub0(h, go);
            E[s] = L::vmax(L::sub0(E[s], ge), open);   // from H before the lazy-F fix-up
            F = L::vmax(L::sub0(F, ge), open);
            Hv = Hl[s];
        }
        // lazy F: carry gaps across lane boundaries; E is deliberately left alone
        if (kByte) {
            int s = 0;
            F = L::shift_up(F);
            while (L::any_above(F, L::sub0(Hs[s], go))) {
                Hs[s] = L::vmax(Hs[s], F);
                cm = L::vmax(cm, Hs[s]);
                F = L::sub0(F, ge);
                if (++s >= seg) {
                    s = 0;
                    F = L::shift_up(F);
                }
            }
        } else {
            bool done = false;
            for (int k = 0; k < LANES && !done; ++k) {
                F = L::shift_up(F);
                for (int s = 0; s < seg && !done; ++s) {
                    Hs[s] = L::vmax(Hs[s], F);
                    cm = L::vmax(cm, Hs[s]);
                    F = L::sub0(F, ge);
                    if (!L::any_above(F, L::sub0(Hs[s], go))) done = true;
                }
            }
        }
        const int col_max = L::hmax(cm);
        if (col_max > best) {
            best = col_max;
            if (kByte && best + bias >= 255) {
                overflow = true;
                break;
            }
            best_ref = i;
            memcpy(Hbest, Hs, sizeof(V) * seg);
        }
        if (col_max == terminate) break;
    }
    for (int s = 0; s < seg; ++s)
        for (int l = 0; l < LANES; ++l)
            if (Hbest[s][l] == best) {
                const int q = s + l * seg;
                if (q < best_read) best_read = q;
            }
    Best b;
    b.score = (kByte && (overflow || best + bias >= 255)) ? 255 : best;
    b.ref = best_ref;
    b.read = best_read;
    return b;
}

// dispatch: the fast version whenever no score can come near the 16-bit cap
Best striped(const int8_t* ref, bool backwards, int n, const int8_t* read, int m, int gap_open, int gap_ext,
             const int8_t* mat, int lanes, int bias, int terminate) {
    int top = 0;
    for (int k = 0; k < 25; ++k) top = std::max<int>(top, mat[k]);
    const long long bound = (long long)std::min(n, m) * top + top + bias;
    static const bool general_only = getenv("HELEN_SSW_GENERAL") != nullptr;   // tests: exercise the version below
    // (the 8-bit lanes hold the penalties and the biased profile as bytes)
    const bool fits = lanes == 16 ? (gap_open <= 255 && gap_ext <= 255 && top + bias <= 255 && bias >= 0)
                                  : (gap_open < 32000 && gap_ext < 32000);
    if (!general_only && bound < 32000 && fits)
        return lanes == 16 ? striped_pass_small<16>(ref, backwards, n, read, m, gap_open, gap_ext, mat, bias, terminate)
                           : striped_pass_small<8>(ref, backwards, n, read, m, gap_open, gap_ext, mat, bias, terminate);
    return striped_pass(ref, backwards, n, read, m, gap_open, gap_ext, mat, lanes, bias, terminate);
}

// Does the 8-bit pass certainly saturate?  It reports 255 (and the caller turns to the 16-bit pass) as soon as a
// column's maximum + bias reaches 255; a cell at the end of a gap-free run of k equal bases holds at least k * match
// whatever E and F are (H = max(diagonal, E, F), saturating), so ONE common substring of need = ceil((255 - bias) /
// match) real bases (63 with stitch's 4 / 6) settles it -- and two overlapping ends of neighbouring regions nearly
// always have one.  Found by indexing the reference's 16-mers (2 bits a base: one 32-bit word) and extending the hits of
// every 16th 16-mer of the query... of every (need - 15)th at most, so that no run of `need` is stepped over.
bool surely_saturates(const int8_t* ref, int n, const int8_t* read, int m, int match, int bias) {
    if (match <= 0) return false;
    const int need = (255 - bias + match - 1) / match;
    constexpr int K = 16;
    if (need < K || n < need || m < need) return false;
    thread_local std::vector<int> head;            // open hash of the reference's K-mers: first position + chain
    thread_local std::vector<int> next;
    constexpr int kBuckets = 1024;
    head.assign(kBuckets, -1);
    next.assign((size_t)n, -1);
    auto bucket = [](uint32_t w) { return (int)((w * 2654435761u) >> 22); };
    uint32_t w = 0;
    int valid = 0;
    for (int i = 0; i < n; ++i) {
        if (ref[i] > 3) { valid = 0; continue; }
        w = (w << 2) | (uint32_t)ref[i];
        if (++valid >= K) {
            const int b = bucket(w), at = i - K + 1;
            next[at] = head[b];
            head[b] = at;
        }
    }
    const int step = need - K + 1;                 // a run of `need` holds a K-mer starting at a multiple of `step`
    for (int q = 0; q + K <= m; q += step) {
        uint32_t v = 0;
        bool ok = true;
        for (int k = 0; k < K; ++k) {
            if (read[q + k] > 3) { ok = false; break; }
            v = (v << 2) | (uint32_t)read[q + k];
        }
        if (!ok) continue;
        for (int at = head[bucket(v)]; at >= 0; at = next[at]) {
            if (memcmp(ref + at, read + q, K) != 0) continue;
            int lo = 0, hi = K;                    // extend the hit both ways over equal real bases
            while (at - lo > 0 && q - lo > 0 && ref[at - lo - 1] == read[q - lo - 1] && ref[at - lo - 1] <= 3) ++lo;
            while (at + hi < n && q + hi < m && ref[at + hi] == read[q + hi] && ref[at + hi] <= 3) ++hi;
            if (lo + hi >= need) return true;
        }
    }
    return false;
}

// ------------------------------------------------------------------------------------------------
// The common join, without the three passes.  Two neighbouring regions called from the same reads agree base for base
// where they overlap, so the overlap strings stitch aligns -- the last `ov` bases of the running sequence and the first
// `ov` of the next region -- usually share ONE long exact run (the left string's head is the right string's middle).
// For that case the library's answer follows from two facts that are cheap to establish:
//   (1) L = the length of the longest common SUBSEQUENCE of the two strings (bit-parallel, a few hundred word
//       operations) bounds the number of match columns of ANY alignment, local or not, gapped or not;
//   (2) the two strings have a common SUBSTRING of exactly that length L.
// Then, with match > 0 and mismatch, gap_open, gap_extend > 0 and only A, C, G, T in both strings: every alignment scores
// at most L * match, and one that reaches it has L match columns and nothing else -- it is an occurrence of a common
// substring of length L.  The striped pass computes the cells of such a gap-free run exactly (H >= diagonal + match,
// and its values never exceed the true ones: its one deviation from Smith-Waterman, E taken before the lazy-F fix-up,
// only loses paths), so its best score is L * match and its best cell -- the first reference column that raises the
// maximum, smallest query index in it -- is the end of the occurrence with the smallest reference start a, and among
// those the smallest query start b.  The backward pass over the two prefixes stops at the first column, walking down from
// that end, whose maximum is the forward score: the occurrence's own start (any other occurrence inside the box starts
// further left; in that column the smallest reversed query index is the largest start <= b, which is b).  The banded
// pass over the box aligns two EQUAL strings: all diagonal ("diagonal on ties").  Result: score L * match, reference
// [a, a + L), query [b, b + L), CIGAR  bS L= tailS, no mismatches.  Checked against the reference library on 2e5
// randomised and adversarial pairs (tests/test_stitch.py); when either fact fails the three passes run as before.
// ------------------------------------------------------------------------------------------------
std::atomic<int> g_fast_path{1};
std::atomic<long long> g_fast_hits{0}, g_fast_hits2{0}, g_fast_misses{0};

// length of the longest common subsequence of ref[0..n) and read[0..m), codes 0..3 only, n <= 64 * kLcsWords
constexpr int kLcsWords = 16;
int lcs_length(const int8_t* ref, int n, const int8_t* read, int m) {
    const int words = (n + 63) / 64;
    uint64_t mask[4][kLcsWords];
    for (int c = 0; c < 4; ++c)
        for (int w = 0; w < words; ++w) mask[c][w] = 0;
    for (int i = 0; i < n; ++i) mask[ref[i]][i >> 6] |= 1ull << (i & 63);
    uint64_t V[kLcsWords];
    for (int w = 0; w < words; ++w) V[w] = ~0ull;
    for (int j = 0; j < m; ++j) {
        const uint64_t* M = mask[read[j]];
        unsigned carry = 0;
        for (int w = 0; w < words; ++w) {       // V = (V + (V & M)) | (V & ~M), the sum carried across words
            const uint64_t v = V[w], u = v & M[w];
            const uint64_t s1 = v + u;
            const uint64_t s2 = s1 + carry;
            carry = (s1 < v) | (s2 < s1);
            V[w] = s2 | (v & ~M[w]);
        }
    }
    int zeros = 0;
    for (int w = 0; w < words; ++w) {
        uint64_t v = ~V[w];
        if (w == words - 1 && (n & 63)) v &= (1ull << (n & 63)) - 1;
        zeros += __builtin_popcountll(v);
    }
    return zeros;
}

// -> true and (a, b, L) when the certificate holds
bool exact_overlap(const int8_t* ref, int n, const int8_t* read, int m, int* a_out, int* b_out, int* len_out) {
    if (n > 64 * kLcsWords || n < 1 || m < 1) return false;
    for (int i = 0; i < n; ++i)
        if (ref[i] > 3) return false;
    for (int j = 0; j < m; ++j)
        if (read[j] > 3) return false;
    const int L = lcs_length(ref, n, read, m);
    if (L < 1) return false;
    // smallest a with ref[a, a + L) somewhere in read, and there the smallest b (memmem returns the first occurrence)
    for (int a = 0; a + L <= n; ++a) {
        const void* hit = memmem(read, (size_t)m, ref + a, (size_t)L);
        if (hit) {
            *a_out = a;
            *b_out = (int)((const int8_t*)hit - read);
            *len_out = L;
            return true;
        }
    }
    return false;
}

struct Op {
    char op;
    int len;
};

// Banded DP inside the box ref[0..n) x read[0..m) that reproduces `score`, then the traceback.
// Row i = query position, column j = reference position; the band of row i is
// [max(0, i-w), min(n-1, i+w)].  dir packs, per cell, the choices for E (query gap continues /
// opens), F (reference gap) and H.
bool banded_cigar(const int8_t* ref, const int8_t* read, int n, int m, int score, int gap_open,
                  int gap_ext, int w, const int8_t* mat, std::vector<Op>* out) {
    std::vector<int> hb, eb, hc;
    std::vector<int8_t> dir;
    int best = 0;
    int width_d = 0;
    auto left = [&](int i) { return std::max(0, i - w); };
    do {
        const int width = 2 * w + 3;
        width_d = 2 * w + 1;
        hb.assign(width + 1, 0);
        eb.assign(width + 1, 0);
        hc.assign(width + 1, 0);
        dir.assign((size_t)width_d * m * 3 + 3, 0);
        best = 0;
        for (int i = 0; i < m; ++i) {
            const int beg = std::max(0, i - w);
            const int end = std::min(n - 1, i + w);
            const int edge = std::min(end + 1, width - 1);
            int f = 0, u = 0;
            hb[0] = eb[0] = hb[edge] = eb[edge] = hc[0] = 0;
            int8_t* d = &dir[(size_t)width_d * i * 3];
            for (int j = beg; j <= end; ++j) {
                u = j - left(i) + 1;                       // this cell in this row's band
                const int up = j - left(i - 1) + 1;        // (i-1, j)   in the previous row's band
                const int lf = j - 1 - left(i) + 1;        // (i, j-1)
                const int dg = j - 1 - left(i - 1) + 1;    // (i-1, j-1)
                const int x = (j - left(i)) * 3;
                int a = (i == 0 ? 0 : hb[up]) - gap_open;
                int b = (i == 0 ? 0 : eb[up]) - gap_ext;
                if (i == 0) {
                    a = -gap_open;
                    b = -gap_ext;
                }
                eb[u] = std::max(a, b);
                d[x + 0] = a > b ? 3 : 2;
                a = hc[lf] - gap_open;
                b = f - gap_ext;
                f = std::max(a, b);
                d[x + 1] = a > b ? 5 : 4;
                const int e1 = std::max(eb[u], 0), f1 = std::max(f, 0);
                const int gap = std::max(e1, f1);
                const int diag = hb[dg] + mat[ref[j] * 5 + read[i]];
                hc[u] = std::max(gap, diag);
                best = std::max(best, hc[u]);
                if (gap <= diag)
                    d[x + 2] = 1;
                else
                    d[x + 2] = e1 > f1 ? d[x + 0] : d[x + 1];
            }
            for (int j = 1; j <= u; ++j) hb[j] = hc[j];
        }
        w *= 2;
    } while (best < score);
    w /= 2;

    // traceback from the box's far corner; stops when the query is used up
    std::vector<Op> rev;
    int i = m - 1, j = n - 1, run = 0, which = 2;
    char op = 'M', prev = 'M';
    while (i > 0) {
        const int x = (j - left(i)) * 3 + which;
        if (j - left(i) < 0 || j - left(i) >= width_d) return false;
        const int8_t t = dir[(size_t)width_d * i * 3 + x];
        switch (t) {
            case 1: --i; --j; which = 2; op = 'M'; break;
            case 2: --i; which = 0; op = 'I'; break;
            case 3: --i; which = 2; op = 'I'; break;
            case 4: --j; which = 1; op = 'D'; break;
            case 5: --j; which = 2; op = 'D'; break;
            default: return false;
        }
        if (op == prev) {
            ++run;
        } else {
            rev.push_back({prev, run});
            prev = op;
            run = 1;
        }
    }
    if (op == 'M') {
        rev.push_back({'M', run + 1});
    } else {
        rev.push_back({op, run});
        rev.push_back({'M', 1});
    }
    out->assign(rev.rbegin(), rev.rend());
    return true;
}

}  // namespace

extern "C" {

/* Local alignment of `query` against `ref` with the reference library's semantics
 * (Aligner(match, mismatch, gap_open, gap_extend); SetReferenceSequence; Align_cpp(query, Filter(), al, 0)).
 *   out[6] = best score, ref_begin, ref_end, query_begin, query_end, mismatches (0-based, inclusive)
 *   cigar  = extended CIGAR (=, X, I, D, with S soft clips), NUL-terminated, truncated to cap
 * Returns 0; 1 if either sequence is empty (the reference's Align_cpp returns false); -1 on an
 * internal inconsistency. */
int helen_ssw_align(const char* ref_seq, int ref_len, const char* query_seq, int query_len, int match,
                    int mismatch, int gap_open, int gap_extend, int* out, char* cigar, int cigar_cap) {
    for (int k = 0; k < 6; ++k) out[k] = 0;
    if (cigar_cap > 0) cigar[0] = 0;
    if (ref_len <= 0 || query_len <= 0) return 1;
    int8_t mat[25];
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b) mat[a * 5 + b] = (a < 4 && b < 4 && a == b) ? (int8_t)match : (int8_t)-mismatch;
    std::vector<int8_t> ref(ref_len), read(query_len);
    for (int k = 0; k < ref_len; ++k) ref[k] = base_code(ref_seq[k]);
    for (int k = 0; k < query_len; ++k) read[k] = base_code(query_seq[k]);

    // the common join: one long exact run shared by the two strings (see exact_overlap)
    const bool shortcuts = g_fast_path.load(std::memory_order_relaxed) && match > 0 && mismatch > 0 && gap_open > 0 && gap_extend >= 0;
    auto exact_run_result = [&](int a, int b, int len) {      // reference [a, a + len) == query [b, b + len), nothing else aligned
        out[0] = len * match;
        out[1] = a;
        out[2] = a + len - 1;
        out[3] = b;
        out[4] = b + len - 1;
        out[5] = 0;
        if (cigar_cap > 0) {
            char text[96];
            int at = 0;
            if (b > 0) at += snprintf(text + at, sizeof(text) - at, "%dS", b);
            at += snprintf(text + at, sizeof(text) - at, "%d=", len);
            const int tail = query_len - (b + len);
            if (tail > 0) at += snprintf(text + at, sizeof(text) - at, "%dS", tail);
            snprintf(cigar, cigar_cap, "%s", text);
        }
    };
    if (shortcuts) {
        int a = 0, b = 0, len = 0;
        if (exact_overlap(ref.data(), ref_len, read.data(), query_len, &a, &b, &len) &&
            (long long)len * match + match + mismatch < 32000) {
            g_fast_hits.fetch_add(1, std::memory_order_relaxed);
            exact_run_result(a, b, len);
            return 0;
        }
    }
    // 8-bit pass first; the 16-bit pass replaces it when the score saturates
    const int bias = mismatch;  // |most negative matrix entry|
    int lanes = 16;
    Best fwd;
    // (a pass whose result is known to be "saturated" is not run: an eighth of a typical join's time)
    if (surely_saturates(ref.data(), ref_len, read.data(), query_len, match, bias))
        fwd.score = 255;
    else
        fwd = striped(ref.data(), false, ref_len, read.data(), query_len, gap_open, gap_extend, mat, 16, bias, -1);
    if (fwd.score == 255) {
        lanes = 8;
        fwd = striped(ref.data(), false, ref_len, read.data(), query_len, gap_open, gap_extend, mat, 8, 0, -1);
    }
    const int score = fwd.score, ref_end = fwd.ref, read_end = fwd.read;
    // The second shortcut, after the library's own forward pass: the best cell is the end of an exact run of
    // score / match real bases.  An alignment that scores `score` has at least that many match columns, hence spans at
    // least that many reference columns; the backward pass, walking down from the end column, therefore meets the score
    // first in the run's own first column, in the cell of an alignment that spans exactly those columns and pays for
    // nothing -- the run -- and the banded pass over the box aligns two equal strings.  Begin cell and CIGAR follow.
    if (shortcuts && score > 0 && score % match == 0) {
        const int len = score / match;
        if (len <= ref_end + 1 && len <= read_end + 1) {
            const int8_t* rp = ref.data() + ref_end - len + 1;
            const int8_t* qp = read.data() + read_end - len + 1;
            bool same = memcmp(rp, qp, (size_t)len) == 0;
            for (int k = 0; same && k < len; ++k) same = rp[k] <= 3;
            if (same) {
                g_fast_hits2.fetch_add(1, std::memory_order_relaxed);
                exact_run_result(ref_end - len + 1, read_end - len + 1, len);
                return 0;
            }
        }
    }
    if (shortcuts) g_fast_misses.fetch_add(1, std::memory_order_relaxed);
    // begin cell: reversed query prefix against the reference prefix, walked backwards
    std::vector<int8_t> rq(read.begin(), read.begin() + read_end + 1);
    std::reverse(rq.begin(), rq.end());
    const Best bwd = striped(ref.data(), true, ref_end + 1, rq.data(), read_end + 1, gap_open, gap_extend, mat, lanes,
                             lanes == 16 ? bias : 0, score);
    const int ref_begin = bwd.ref, read_begin = read_end - bwd.read;
    out[0] = score;
    out[1] = ref_begin;
    out[2] = ref_end;
    out[3] = read_begin;
    out[4] = read_end;
    if (ref_begin < 0 || read_begin < 0 || ref_end < ref_begin || read_end < read_begin) return 0;

    const int n = ref_end - ref_begin + 1, m = read_end - read_begin + 1;
    std::vector<Op> ops;
    if (!banded_cigar(ref.data() + ref_begin, read.data() + read_begin, n, m, score, gap_open, gap_extend,
                      std::abs(n - m) + 1, mat, &ops))
        return -1;

    // M runs -> '=' / 'X' runs, soft clips at both ends, mismatch count (ssw_cpp.cpp:97-180)
    std::string s;
    char buf[32];
    auto emit = [&](int len, char op) {
        snprintf(buf, sizeof(buf), "%d%c", len, op);
        s += buf;
    };
    if (read_begin > 0) emit(read_begin, 'S');
    const int8_t* rp = ref.data() + ref_begin;
    const int8_t* qp = read.data() + read_begin;
    int mism = 0, run_eq = 0, run_x = 0;
    auto flush = [&]() {
        if (run_eq) emit(run_eq, '=');
        else if (run_x) emit(run_x, 'X');
        run_eq = run_x = 0;
    };
    for (const Op& o : ops) {
        if (o.op == 'M') {
            for (int k = 0; k < o.len; ++k, ++rp, ++qp) {
                if (*rp != *qp) {
                    ++mism;
                    if (run_eq) emit(run_eq, '=');
                    run_eq = 0;
                    ++run_x;
                } else {
                    if (run_x) emit(run_x, 'X');
                    run_x = 0;
                    ++run_eq;
                }
            }
        } else if (o.op == 'I') {
            qp += o.len;
            mism += o.len;
            flush();
            emit(o.len, 'I');
        } else {
            rp += o.len;
            mism += o.len;
            flush();
            emit(o.len, 'D');
        }
    }
    flush();
    const int tail = query_len - read_end - 1;
    if (tail > 0) emit(tail, 'S');
    out[5] = mism;
    if (cigar_cap > 0) snprintf(cigar, cigar_cap, "%s", s.c_str());
    return 0;
}

/* The exact-overlap shortcut of helen_ssw_align (on by default): enable = 0 / 1 switches it, anything else only asks;
 * returns the previous setting.  helen_ssw_fast_path_counts: joins answered by it / handed on to the three passes since
 * the library was loaded. */
int helen_ssw_fast_path(int enable) {
    const int before = g_fast_path.load();
    if (enable == 0 || enable == 1) g_fast_path.store(enable);
    return before;
}

void helen_ssw_fast_path_counts(long long* hits, long long* misses) {
    if (hits) *hits = g_fast_hits.load() + g_fast_hits2.load();
    if (misses) *misses = g_fast_misses.load();
}

long long helen_ssw_fast_path_after_forward(void) { return g_fast_hits2.load(); }

}  // extern "C"
