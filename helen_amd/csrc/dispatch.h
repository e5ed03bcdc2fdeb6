// dispatch.h -- which kernels a call takes: ONE table, derived from the device's CU count.
//
// Every stage of the fp32 path has several kernels that give the same bits (same MFMA order per accumulator, same gate
// cell: tests/test_gpu_scale.py), so the choice is pure scheduling: how many workgroups a launch has against how many
// CUs the device has.  This header is the whole of that decision -- plain C++, no HIP, no globals: plan_chunk /
// plan_exact_encoder / plan_call map (tiles, CUs) to kernels, describe_dispatch prints the table for a device, and the
// environment overrides (A/B probes and tests) are read ONCE, by read_overrides at helen_model_create (or again through
// helen_reload_overrides), never per call.  helen_describe_dispatch (C ABI) runs it dry for any CU count.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace helen {

enum RecurrenceKernel {       // fp32 GRU recurrence of one chunk (encoder and decoder launches alike)
    kRecPlain = 0,            // gru_kernel: one tile per 4-wave workgroup, two workgroups per CU
    kRecSingle8,              // gru_single8_kernel: one tile per 8-wave workgroup, at most one (tile, direction) per CU
    kRecHalf8,                // gru_half8_kernel: half tiles (8 windows) on v_mfma_f32_4x4x1, 4 workgroups per tile pair
    kRecQuarter4,             // gru_quarter4_kernel: quarter tiles (4 windows), 8 workgroups per tile pair
    kRecPair,                 // gru_pair_kernel: two tiles per 8-wave workgroup, one workgroup per CU
};
enum DecoderProjection {
    kDecStreaming = 0,        // gemm_gi_kernel<16, true>: thousands of single-wave workgroups
    kDecStationary,           // gemm_dec_ws_kernel: one workgroup per (tile, direction), weights in registers
    kDecStationaryRuns,       // gemm_dec_wsp_kernel: the same with a (tile, direction)'s positions cut into runs
};
inline const char* name_of(RecurrenceKernel k) {
    static const char* n[] = {"gru_kernel", "gru_single8_kernel", "gru_half8_kernel", "gru_quarter4_kernel", "gru_pair_kernel"};
    return n[k];
}
inline const char* name_of(DecoderProjection k) {
    static const char* n[] = {"gemm_gi_kernel<16>", "gemm_dec_ws_kernel", "gemm_dec_wsp_kernel"};
    return n[k];
}
// -1 = not forced.  HELEN_<NAME>=0 / 1 forces a choice off / on whatever the size (A/B probes; every forced choice still
// gives the same bits).
struct Overrides {
    int gru_pair = -1, gru_single8 = -1, gru_half8 = -1, gru_quarter4 = -1;
    int dec_ws = -1, dec_wsp = -1, dec_wsp_parts = 0;
    int split = -1, split_at = 0;
    int bf16_pair = -1;
    int x3_pair = -1;                           // fp32x3: 1 / 0 = two tiles per workgroup (gru_x3_il_kernel) / one (gru_x3_kernel)
    int host_lock = -1;                         // helen_polish_host: 0 never page-lock caller memory, 1 ranges that own their pages, 2 all
    bool verbose = false;
};
inline int flag_of(const char* name) {
    const char* v = getenv(name);
    return (v && *v) ? (*v == '1' ? 1 : 0) : -1;
}
inline Overrides read_overrides() {
    Overrides o;
    o.gru_pair = flag_of("HELEN_GRU_PAIR");
    o.gru_single8 = flag_of("HELEN_GRU_SINGLE8");
    o.gru_half8 = flag_of("HELEN_GRU_HALF8");
    o.gru_quarter4 = flag_of("HELEN_GRU_QUARTER4");
    o.dec_ws = flag_of("HELEN_DEC_WS");
    o.dec_wsp = flag_of("HELEN_DEC_WSP");
    if (const char* n = getenv("HELEN_DEC_WSP_PARTS")) o.dec_wsp_parts = atoi(n);
    o.split = flag_of("HELEN_SPLIT");
    if (const char* n = getenv("HELEN_SPLIT_AT")) o.split_at = atoi(n);
    o.bf16_pair = flag_of("HELEN_BF16_PAIR");
    o.x3_pair = flag_of("HELEN_X3_PAIR");
    // exactly none | own | all ("0" = none); anything else is ignored: a typo must not re-open the in-place page-locking
    if (const char* hl = getenv("HELEN_HOST_LOCK")) {
        if (!strcmp(hl, "none") || !strcmp(hl, "0")) o.host_lock = 0;
        else if (!strcmp(hl, "own")) o.host_lock = 1;
        else if (!strcmp(hl, "all")) o.host_lock = 2;
        else fprintf(stderr, "helen: HELEN_HOST_LOCK=%s ignored (none | own | all)\n", hl);
    }
    o.verbose = flag_of("HELEN_VERBOSE") == 1;
    return o;
}

constexpr int kDecStagePositions = 4;    // HELEN_DWS_PB: positions per stage of gemm_dec_ws(p)_kernel
constexpr int kEncStagePositions = 8;    // positions per stage of gemm_enc_x3_kernel

struct ChunkPlan {
    RecurrenceKernel recurrence;
    DecoderProjection decoder;
    int dec_parts = 0, dec_run = 0;      // kDecStationaryRuns: runs per (tile, direction), positions per run (whole stages)
};
struct CallPlan {
    bool split = false;                  // two tile groups on two internal streams
    int first_group = 0;                 // tiles of the first group
};

// ---- the rules (measured on 256 CUs, stated in CUs so that they follow the device) ----
// gru_pair_kernel against one tile per workgroup: a launch lasts as long as its longest CU queue.  Per resident set at
// 100 steps: one workgroup per CU alone 0.31 ms, two per CU 0.635 ms (gru_kernel), a pair workgroup 0.617 ms.
inline bool pair_pays(int tiles, int cus) {
    const int wg_single = 2 * tiles, wg_pair = 2 * ((tiles + 1) / 2);
    const int full = wg_single / (2 * cus), rest = wg_single % (2 * cus);
    const double t_single = full * 0.635 + (rest == 0 ? 0.0 : rest <= cus ? 0.31 : 0.635);
    const double t_pair = ((wg_pair + cus - 1) / cus) * 0.617;
    return t_pair < t_single;
}
inline bool dec_stationary_pays(int tiles, int cus) {   // whole rounds of one workgroup per CU, at most an eighth of the last idle
    const int wgs = 2 * tiles, rounds = (wgs + cus - 1) / cus;
    return wgs >= cus && rounds * cus - wgs <= cus / 8;
}

inline ChunkPlan plan_chunk(int tiles, int T, int cus, const Overrides& o) {
    ChunkPlan p;
    if (o.gru_pair >= 0 ? o.gru_pair == 1 : pair_pays(tiles, cus)) p.recurrence = kRecPair;
    else if (o.gru_quarter4 >= 0 ? o.gru_quarter4 == 1 : 8 * tiles <= cus) p.recurrence = kRecQuarter4;   // an eighth of the CUs in tiles
    else if (o.gru_half8 >= 0 ? o.gru_half8 == 1 : 4 * tiles <= cus) p.recurrence = kRecHalf8;            // a quarter
    else if (o.gru_single8 >= 0 ? o.gru_single8 == 1 : 2 * tiles <= cus) p.recurrence = kRecSingle8;      // one (tile, direction) per CU
    else p.recurrence = kRecPlain;
    if (o.dec_ws >= 0 ? o.dec_ws == 1 : dec_stationary_pays(tiles, cus)) {
        p.decoder = kDecStationary;
    } else if (o.dec_wsp != 0 && (o.dec_wsp == 1 || 4 * tiles <= cus)) {
        // about one workgroup per CU: cus / (2 x tiles rounded up to the XCD count) runs per (tile, direction)
        int want = o.dec_wsp_parts > 0 ? o.dec_wsp_parts : cus / (2 * ((tiles + 7) / 8 * 8));
        want = want < 1 ? 1 : want;
        const int per = (T + want - 1) / want;
        p.dec_run = (per + kDecStagePositions - 1) / kDecStagePositions * kDecStagePositions;
        p.dec_parts = (T + p.dec_run - 1) / p.dec_run;
        p.decoder = kDecStationaryRuns;
    } else {
        p.decoder = kDecStreaming;
    }
    return p;
}

// The exact-product encoder projection (gemm_enc_x3_kernel, the polish entry points of the fp32 and fp32x3 modes): three
// (tile, column set) workgroups per tile; a call with fewer of them than CUs cuts the positions into runs of whole stages.
struct ExactEncoderPlan {
    int parts = 1, run = 0;
};
inline ExactEncoderPlan plan_exact_encoder(int tiles, int npos, int cus) {
    ExactEncoderPlan p;
    const int wgs = 3 * ((tiles + 7) / 8 * 8);
    int want = cus / wgs;
    want = want < 1 ? 1 : want;
    const int per = (npos + want - 1) / want;
    p.run = (per + kEncStagePositions - 1) / kEncStagePositions * kEncStagePositions;
    p.parts = (npos + p.run - 1) / p.run;
    return p;
}

// fp32 calls that neither fill the chip with tile pairs nor fit one (tile, direction) per CU run as two tile groups on two
// streams (more than half and fewer than 15/16 of the CUs in tiles), and so do calls of a little more than a quarter
// (up to a third) of the CUs in tiles: the quarter that fills the chip with half-tile recurrences plus the rest.
inline CallPlan plan_call(int tiles, int cus, bool fp32, const Overrides& o) {
    CallPlan p;
    if (!fp32 || tiles < 2) return p;
    const bool upper = 2 * tiles > cus && 16 * tiles < 15 * cus, lower = 4 * tiles > cus && 3 * tiles <= cus;
    p.split = o.split >= 0 ? o.split == 1 : (upper || lower);
    if (!p.split) return p;
    p.first_group = lower ? cus / 4 : 8 * tiles >= 5 * cus ? cus / 2 : (tiles + 1) / 2;
    if (p.first_group >= tiles || p.first_group < 1) p.first_group = (tiles + 1) / 2;
    if (o.split_at > 0 && o.split_at < tiles) p.first_group = o.split_at;
    return p;
}

inline bool bf16_pair_pays(int tiles, int cus, const Overrides& o) { return o.bf16_pair >= 0 ? o.bf16_pair == 1 : 2 * tiles > cus; }
// fp32x3 likewise: above half the CUs in tiles the two-tile kernel (gate math inside the other tile's MFMA stream)
inline bool x3_pair_pays(int tiles, int cus, const Overrides& o) { return o.x3_pair >= 0 ? o.x3_pair == 1 : 2 * tiles > cus; }

// The table for a device of `cus` compute units: one row per run of tile counts with the same plan.
inline std::string describe_dispatch(int cus, const Overrides& o, int max_tiles = 0) {
    if (max_tiles <= 0) max_tiles = 2 * cus;
    std::string out;
    char line[512];
    snprintf(line, sizeof(line), "dispatch for %d CUs (tiles of 16 windows; every row gives the same bits)\n"
             "%-12s %-22s %-26s %-36s %s\n", cus, "tiles", "recurrences", "decoder projection", "encoder projection (exact products)", "call as");
    out += line;
    auto row = [&](int tiles) {
        const CallPlan c = plan_call(tiles, cus, true, o);
        char buf[400];
        if (c.split) {
            snprintf(buf, sizeof(buf), "two tile groups on two streams: the first %s tiles, each planned as a call of its own",
                     c.first_group == cus / 4 ? "CUs/4" : c.first_group == cus / 2 ? "CUs/2" : "half of the");
            return std::string("-|-|-|") + buf;
        }
        const ChunkPlan k = plan_chunk(tiles, 100, cus, o);
        const ExactEncoderPlan e = plan_exact_encoder(tiles, 1000, cus);
        char enc[64];
        snprintf(enc, sizeof(enc), e.parts > 1 ? "gemm_enc_x3_kernel (%d position runs)" : "gemm_enc_x3_kernel", e.parts);
        snprintf(buf, sizeof(buf), "%s|%s%s|%s|one sequence", name_of(k.recurrence), name_of(k.decoder),
                 k.decoder == kDecStationaryRuns ? " (position runs)" : "", enc);
        return std::string(buf);
    };
    int start = 1;
    std::string cur = row(1);
    for (int t = 2; t <= max_tiles + 1; ++t) {
        const std::string r = t <= max_tiles ? row(t) : std::string();
        if (r != cur) {
            std::string cols[4];
            size_t b = 0;
            for (int i = 0; i < 4; ++i) {
                const size_t e = i < 3 ? cur.find('|', b) : std::string::npos;
                cols[i] = cur.substr(b, e == std::string::npos ? std::string::npos : e - b);
                b = e == std::string::npos ? cur.size() : e + 1;
            }
            char range[32];
            snprintf(range, sizeof(range), start == t - 1 ? "%d" : "%d-%d", start, t - 1);
            snprintf(line, sizeof(line), "%-12s %-22s %-26s %-36s %s\n", range, cols[0].c_str(), cols[1].c_str(), cols[2].c_str(),
                     cols[3].c_str());
            out += line;
            start = t;
            cur = r;
        }
    }
    return out;
}

}  // namespace helen
