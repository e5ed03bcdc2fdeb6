// ssw_ref_shim.cpp -- TEST INFRASTRUCTURE.  A C entry point around the REFERENCE's own striped
// Smith-Waterman (helen/modules/src/local_reassembly/ssw.c, ssw_cpp.cpp), compiled from the sources
// where they lie under /root/reference into oracle/_ref/libssw_ref.so (oracle/Makefile target `ref`).
// It is what helen_amd's own aligner (helen_amd/csrc/ssw.cpp) is fuzz-checked against; nothing in
// helen_amd/ links or loads it.  The call mirrors the reference's use in Stitch.py:111-134:
//   aligner = Aligner(match, mismatch, gap_open, gap_extend); aligner.SetReferenceSequence(ref, len);
//   aligner.Align_cpp(query, Filter(), alignment, 0)
#include <cstring>
#include <string>

#include "local_reassembly/ssw_cpp.h"

extern "C" int ssw_ref_align(const char* ref, int ref_len, const char* query, int match, int mismatch,
                             int gap_open, int gap_extend, int* out /* score, ref_begin, ref_end,
                             query_begin, query_end, mismatches */, char* cigar, int cigar_cap) {
    StripedSmithWaterman::Aligner aligner((uint8_t)match, (uint8_t)mismatch, (uint8_t)gap_open,
                                          (uint8_t)gap_extend);
    StripedSmithWaterman::Filter filter;
    StripedSmithWaterman::Alignment al;
    al.Clear();
    aligner.SetReferenceSequence(ref, ref_len);
    const bool ok = aligner.Align_cpp(query, filter, &al, 0);
    out[0] = al.sw_score;
    out[1] = al.ref_begin;
    out[2] = al.ref_end;
    out[3] = al.query_begin;
    out[4] = al.query_end;
    out[5] = al.mismatches;
    snprintf(cigar, cigar_cap, "%s", al.cigar_string.c_str());
    return ok ? 0 : 1;
}
