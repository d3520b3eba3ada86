// Bit-exact parallel evaluation of numpy's sequential float32 column mean (pch_mean.hip).
#pragma once
#include "pch_common.h"

namespace pch {

struct MsHdr;
struct MsRec;
struct MsPred;
// Candidate rows for the height filter, emitted by the summary pass while it has the rows in LDS anyway (pch_filter.hip,
// gf_cand_k): rows whose RAW z exceeds `tcand`, a deliberately low estimate of the filter's threshold.
//   slots  [nb][4][MS_CAND_SLOT] words: four PLANES per block - x, y, z and the bits of the row's index inside its
//          1024-row block - file order inside a block.  Planes, not float4 rows: the sweep first counts survivors
//          from the z plane alone (a quarter of the bytes) and then reads the rows once; as rows it pulled every line
//          twice (0.55 GB for 0.35 GB of necessary traffic).  A slot holds half of its block: typical data keeps ~10 % of the rows (a block next to a tower in
//          flight-line order more), and slots that are 8 KB apart instead of 16 KB are what makes their reads and writes stream; a block with more candidates
//          raises the overflow word and the sweep then reads the tile as before
//   counts [nb]       rows used in every slot
//   tcand  [2]        [0] the threshold that was used (device; written before the summary runs), [1] overflow word
constexpr int MS_CAND_SLOT = 512;
struct MsCand {
    float*    slots;
    uint32_t* counts;
    float*    tcand;
    float*    zsample;           // [nb] one sampled z per block (scratch of the estimate)
    double    pct;               // the percentile the filter will ask for
    float     add;               // tcand = (estimated percentile of z) + add
};
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
// walk_stream (optional, needs ev_zcol): level 2 and the walk run there, behind that event, so that whatever the caller
// enqueues on `s` next (the percentile passes over zcol) starts without a cross-stream hop; the caller joins
// walk_stream before it reads `out`.
// sum_in (optional, device float[3]): running sum to continue instead of +0.0; divide_n: MS_DIVIDE_BY_N (the
// mean of these n rows), MS_NO_DIVIDE (out = the running sum after the rows) or the row count to divide by.
// phase: both kernels groups (default), only the tables (summary + level 2: they do not depend on sum_in), or only
// the walk over tables an earlier call left in `w`.
constexpr int64_t MS_DIVIDE_BY_N = -1, MS_NO_DIVIDE = -2;
constexpr int MS_PHASE_BOTH = 0, MS_PHASE_TABLES = 1, MS_PHASE_WALK = 2;
// cand (optional): emit the candidate rows described above; *cand_made says whether they were (needs the sampled
// estimate, i.e. an array of at least two 65 536-row groups and prediction enabled)
int mean_seq_launch(const float* xyz, int64_t n, float* out, MsWs& w, float* zcol, hipStream_t s,
                    hipEvent_t ev_zcol = nullptr, const float* sum_in = nullptr, int64_t divide_n = MS_DIVIDE_BY_N,
                    int phase = MS_PHASE_BOTH, const MsCand* cand = nullptr, bool* cand_made = nullptr,
                    hipStream_t walk_stream = nullptr);
int mean_seq_serial_launch(const float* xyz, int64_t n, float* out, hipStream_t s);

}  // namespace pch
