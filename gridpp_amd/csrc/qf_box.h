// quantile_fast, 3-D input: the count planes between the member pass (neighbourhood.hip) and the box pass (qf_box.hip).
#pragma once
#include "common.h"

// Byte planes in HBM, padded so that the box pass never tests a bound: plane t (t < T) = #(valid members <= thr[t]),
// plane T = #valid members, each [Yp][Xp] bytes with the cell (y, x) at row y + QF_PADY, column x + QF_PADX.
// Padding bytes: 255 in the threshold planes (the table of the box pass maps 255 to 0.0: a padding cell adds nothing
// to a window sum), E in the plane of the valid counts.  rowflag[y] != 0: some cell of row y has fewer than E valid
// members (the box pass then forms count / valid per cell in that row instead of looking count / E up).
#define QF_PADX 16        // >= the largest halfwidth of the fused path; a multiple of 16 keeps every 16-column segment 16-byte aligned
#define QF_PADY 17        // the outgoing row of output row 0 is row -halfwidth - 1
#define QF_MAXHW 16
struct QfGeom {
    int Y, X, Xp, Yp;
    long Pp;              // bytes per plane (Yp * Xp, a multiple of 16)
    int* rowflag;         // [Y + 2]; rowflag[Y] != 0: some row is flagged; rowflag[Y + 1] != 0: the count pass did not run (k_qf_count, Uexp)
};
inline QfGeom qf_geom(int Y, int X) {
    QfGeom g;
    g.Y = Y; g.X = X;
    g.Xp = (X + QF_PADX + 47) / 16 * 16;     // a segment that starts inside the field reads 32 columns beyond its start
    g.Yp = Y + QF_PADY + QF_MAXHW;
    g.Pp = (long)g.Yp * g.Xp;
    g.rowflag = nullptr;
    return g;
}
__device__ __forceinline__ long qf_cell_offset(const QfGeom& g, const long cell) {
    const long y = cell / g.X;
    return (y + QF_PADY) * g.Xp + (cell - y * g.X) + QF_PADX;
}

// box pass: cnt8 = the T + 1 planes above -> out [Y][X]   (neighbourhood.cpp:473-522, util.cpp:339-414)
void qf_box_launch(const unsigned char* cnt8, const QfGeom& g, int reps, int hw, int T, const float* d_thr, const float* d_q, int qfield, float* d_out, hipStream_t st = nullptr);
