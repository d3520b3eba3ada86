// Bit-exact parallel evaluation of numpy's sequential float32 column mean (pch_mean.hip).
#pragma once
#include "pch_common.h"

namespace pch {

struct MsHdr;
struct MsRec;
struct MsPred;
struct MsWs {
    MsPred*    pred;             // [3][nb2] estimated running sum in front of every level-2 row (candidate windows)
    int*       stats;            // [3][4]: level-2 batches, -, exactly added blocks, descents
    MsRec*     rec;              // [3][nb] level-1 records (three 128-byte lines per column and block)
    MsHdr*     hdr2;
    long long* rows2;
};
void ms_plan(Arena& a, int64_t n, MsWs& w);
// centroid[3] = np.mean(xyz, axis=0) (float32).  zcol (optional, n floats) receives a copy of
// the z column, written by the same pass that reads the tile.
// ev_zcol (optional) is recorded on `s` right after the pass that writes zcol.
// sum_in (optional, device float[3]): running sum to continue instead of +0.0; divide_n: MS_DIVIDE_BY_N (the
// mean of these n rows), MS_NO_DIVIDE (out = the running sum after the rows) or the row count to divide by.
// phase: both kernels groups (default), only the tables (summary + level 2: they do not depend on sum_in), or only
// the walk over tables an earlier call left in `w`.
constexpr int64_t MS_DIVIDE_BY_N = -1, MS_NO_DIVIDE = -2;
constexpr int MS_PHASE_BOTH = 0, MS_PHASE_TABLES = 1, MS_PHASE_WALK = 2;
int mean_seq_launch(const float* xyz, int64_t n, float* out, MsWs& w, float* zcol, hipStream_t s,
                    hipEvent_t ev_zcol = nullptr, const float* sum_in = nullptr, int64_t divide_n = MS_DIVIDE_BY_N,
                    int phase = MS_PHASE_BOTH);
int mean_seq_serial_launch(const float* xyz, int64_t n, float* out, hipStream_t s);

}  // namespace pch
