// Bit-exact parallel evaluation of numpy's sequential float32 column mean (pch_mean.hip).
#pragma once
#include "pch_common.h"

namespace pch {

struct MsHdr;
struct MsRec;
struct MsWs {
    int*       stats;            // [3][4]: level-2 batches, -, exactly added blocks, descents
    MsRec*     rec;              // [3][nb] level-1 records (three 128-byte lines per column and block)
    MsHdr*     hdr2;
    long long* rows2;
};
void ms_plan(Arena& a, int64_t n, MsWs& w);
// centroid[3] = np.mean(xyz, axis=0) (float32).  zcol (optional, n floats) receives a copy of
// the z column, written by the same pass that reads the tile.
// ev_zcol (optional) is recorded on `s` right after the pass that writes zcol.
int mean_seq_launch(const float* xyz, int64_t n, float* out, MsWs& w, float* zcol, hipStream_t s,
                    hipEvent_t ev_zcol = nullptr);
int mean_seq_serial_launch(const float* xyz, int64_t n, float* out, hipStream_t s);

}  // namespace pch
